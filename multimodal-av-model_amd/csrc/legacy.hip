// Kernels of the legacy mel + GRU model (reference: 이전 버전/multimodal_ctc_korean.py:8-55, train loop
// 이전 버전/train_ctc_korea.py:82-109): a two-conv lip front-end (Conv2d 3x3 + ReLU + MaxPool2d(2), twice), two 2-layer
// bidirectional GRUs and a Linear.  The convolutions and every input / weight-gradient product run on av_gemm (implicit im2col /
// k-major operands); this file holds what av_gemm does not: layout packing, the fused ReLU + 2x2 max-pool (forward / backward),
// an explicit im2col for the convolution weight gradients, and the GRU recurrence (forward / backward time-step kernels on MFMA
// 16x16 tiles, both directions per launch - same scheme as lstm.hip).
#include "av_common.h"

namespace {

inline int lg_grid(long long n) {
    long long b = (n + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

// ---------------------------------------------------------------------------------------------------- layout packing
// frames [N][C][H][W] fp32 (the reference's (B,T,C,H,W) view, :23) -> channel-last [N][H][W][Cp] (Cp >= C, zero padded), compute dtype
template <typename T>
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ in, T* __restrict__ out, long long npix, int C, int HW, int Cp) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < npix; i += (long long)gridDim.x * 256) {
        const long long n = i / HW;
        const int hw = (int)(i - n * HW);
        const float* src = in + n * C * HW + hw;
        T* dst = out + i * Cp;
        for (int c = 0; c < Cp; ++c) dst[c] = from_f32<T>(c < C ? src[(long long)c * HW] : 0.f);
    }
}

// ---------------------------------------------------------------------------------------------------- ReLU + MaxPool2d(2)
// x [N][H][W][C] = convolution output incl. bias; y = maxpool2(relu(x)) as [N][H/2][W/2][C] (nchw_out = 0) or [N][C][H/2][W/2]
// (nchw_out = 1: the reference flattens (C, H', W') per frame for the GRU, :25, so the second pool writes that order directly).
template <typename T>
__global__ __launch_bounds__(256) void relu_pool_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, long long nout, int H, int W, int C, int nchw_out) {
    const int Ho = H / 2, Wo = W / 2;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nout; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % C);                              // thread order = channel-last (coalesced reads)
        long long t = i / C;
        const int wo = (int)(t % Wo); t /= Wo;
        const int ho = (int)(t % Ho);
        const long long n = t / Ho;
        const T* p = x + ((n * H + 2 * ho) * W + 2 * wo) * C + c;
        float m = fmaxf(fmaxf(to_f32<T>(p[0]), to_f32<T>(p[C])), fmaxf(to_f32<T>(p[(long long)W * C]), to_f32<T>(p[(long long)W * C + C])));
        m = fmaxf(m, 0.f);
        const long long o = nchw_out ? ((n * C + c) * Ho + ho) * Wo + wo : i;
        y[o] = from_f32<T>(m);
    }
}
// dx = dy routed to the FIRST maximum of each window (row-major scan order, as torch's max_pool2d) where relu passed it (x > 0)
template <typename T>
__global__ __launch_bounds__(256) void relu_pool_bwd_kernel(const T* __restrict__ x, const T* __restrict__ dy, T* __restrict__ dx, long long nout, int H, int W,
                                                            int C, int nchw_dy) {
    const int Ho = H / 2, Wo = W / 2;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nout; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % C);
        long long t = i / C;
        const int wo = (int)(t % Wo); t /= Wo;
        const int ho = (int)(t % Ho);
        const long long n = t / Ho;
        const long long base = ((n * H + 2 * ho) * W + 2 * wo) * C + c;
        const long long o4[4] = {base, base + C, base + (long long)W * C, base + (long long)W * C + C};
        float v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = to_f32<T>(x[o4[k]]);
        int arg = 0;
        float m = v[0];
#pragma unroll
        for (int k = 1; k < 4; ++k) if (v[k] > m) { m = v[k]; arg = k; }
        const float g = to_f32<T>(dy[nchw_dy ? ((n * C + c) * Ho + ho) * Wo + wo : i]);
#pragma unroll
        for (int k = 0; k < 4; ++k) dx[o4[k]] = from_f32<T>((k == arg && m > 0.f) ? g : 0.f);
    }
}

// ---------------------------------------------------------------------------------------------------- im2col (3x3, stride 1, pad 1)
// cols[(n, h, w)][(ky, kx, c)] = x[n][h + ky - 1][w + kx - 1][c] (zero outside): the k-major operand of dW = dY^T cols
template <typename T>
__global__ __launch_bounds__(256) void im2col3_kernel(const T* __restrict__ x, T* __restrict__ cols, long long n, int H, int W, int C) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % C);
        long long t = i / C;
        const int tap = (int)(t % 9); t /= 9;
        const int w = (int)(t % W); t /= W;
        const int h = (int)(t % H);
        const long long img = t / H;
        const int ih = h + tap / 3 - 1, iw = w + tap % 3 - 1;
        T v = from_f32<T>(0.f);
        if (ih >= 0 && ih < H && iw >= 0 && iw < W) v = x[((img * H + ih) * W + iw) * C + c];
        cols[i] = v;
    }
}

// ---------------------------------------------------------------------------------------------------- GRU time steps
// nn.GRU gate order r, z, n (torch:modules/rnn.py): r = s(gi_r + gh_r), z = s(gi_z + gh_z), n = tanh(gi_n + r o gh_n),
// h = (1 - z) o n + z o h_prev, gi = W_ih x + b_ih (one big GEMM over all steps: gx), gh = W_hh h_prev + b_hh (here).
struct GruFwdP {
    const float* gx;     // [T][B][2][3H]  input projections incl. b_ih
    const void* whh;     // [2][3H][H]     compute dtype
    const float* bhh;    // [2][3H]
    void* hseq;          // [T][B][2H]     compute dtype (MFMA operand of the next step, input of the next layer)
    float* hf;           // [T][B][2][H]   fp32 state (the recurrence itself is carried in fp32)
    float* gates;        // [T][B][2][4H]  r, z, n, gh_n (optional: saved for the backward)
    void* out_bt;        // [B][T][2H]     optional batch-major copy, compute dtype
    int T, B, H, s;
};

template <typename T> struct GFrag;
template <> struct GFrag<bf16_t> { typedef bf16x8 type; static constexpr int KS = 32, PER = 8; };
template <> struct GFrag<float> { typedef f32x4 type; static constexpr int KS = 16, PER = 4; };

template <typename T>
__device__ __forceinline__ typename GFrag<T>::type g_ld(const T* p, bool ok) {
    typename GFrag<T>::type z;
#pragma unroll
    for (int i = 0; i < GFrag<T>::PER; ++i) z[i] = from_f32<T>(0.f);
    return ok ? *(const typename GFrag<T>::type*)p : z;
}
__device__ __forceinline__ void g_mma(f32x4& acc, const bf16x8& a, const bf16x8& b) { acc = AV_MFMA_F32_16X16X32_LP(a, b, acc, 0, 0, 0); }
__device__ __forceinline__ void g_mma(f32x4& acc, const f32x4& a, const f32x4& b) {
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[jj], b[jj], acc, 0, 0, 0);
}

constexpr int GMT = 4;    // 64 batch rows per workgroup (blockIdx.z selects the 64-row group)

// grid (H / 16, 2 directions, ceil(B / 64)); 4 wavefronts split K = H, partial tiles meet in LDS, wavefront w finishes row tile w
template <typename T>
__global__ __launch_bounds__(256) void gru_fwd_step(const GruFwdP p) {
    constexpr int KS = GFrag<T>::KS, PER = GFrag<T>::PER;
    __shared__ float red[4][GMT][3][4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, r = lane & 15, g = lane >> 4;
    const int d = blockIdx.y, j0 = blockIdx.x * 16, mbase = blockIdx.z * 64;
    const int H = p.H, B = p.B;
    const int td = d == 0 ? p.s : p.T - 1 - p.s;
    const int tp = d == 0 ? td - 1 : td + 1;
    int nmt = (B - mbase + 15) / 16;
    if (nmt > GMT) nmt = GMT;
    if (p.s > 0) {
        f32x4 acc[GMT][3];
#pragma unroll
        for (int mt = 0; mt < GMT; ++mt)
#pragma unroll
            for (int q = 0; q < 3; ++q) acc[mt][q] = f32x4{0.f, 0.f, 0.f, 0.f};
        const T* hprev = (const T*)p.hseq + ((long long)tp * B) * 2 * H + d * H;            // row stride 2H
        const T* W = (const T*)p.whh + (long long)d * 3 * H * H;
        const int kq = H / 4;
        for (int k0 = w * kq; k0 < (w + 1) * kq; k0 += KS) {
            typename GFrag<T>::type b[3];
#pragma unroll
            for (int q = 0; q < 3; ++q) b[q] = g_ld<T>(W + (long long)(q * H + j0 + r) * H + k0 + PER * g, true);
#pragma unroll
            for (int mt = 0; mt < GMT; ++mt) {
                if (mt < nmt) {
                    const int row = mbase + mt * 16 + r;
                    const auto a = g_ld<T>(hprev + (long long)row * 2 * H + k0 + PER * g, row < B);
#pragma unroll
                    for (int q = 0; q < 3; ++q) g_mma(acc[mt][q], a, b[q]);
                }
            }
        }
#pragma unroll
        for (int mt = 0; mt < GMT; ++mt)
#pragma unroll
            for (int q = 0; q < 3; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e) red[w][mt][q][e][lane] = acc[mt][q][e];
    }
    __syncthreads();
    const int mt = w;
    if (mt >= nmt) return;
    const int j = j0 + r;
    const float* bh = p.bhh + (long long)d * 3 * H;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int row = mbase + mt * 16 + 4 * g + e;
        if (row < B) {
            float pre[3];
#pragma unroll
            for (int q = 0; q < 3; ++q)
                pre[q] = (p.s > 0 ? red[0][mt][q][e][lane] + red[1][mt][q][e][lane] + red[2][mt][q][e][lane] + red[3][mt][q][e][lane] : 0.f) + bh[q * H + j];
            const float* gxr = p.gx + (((long long)td * B + row) * 2 + d) * 3 * H;
            const float rg = sigmoid_f(gxr[j] + pre[0]);
            const float zg = sigmoid_f(gxr[H + j] + pre[1]);
            const float ng = tanhf(gxr[2 * H + j] + rg * pre[2]);
            const float hp = p.s > 0 ? p.hf[(((long long)tp * B + row) * 2 + d) * H + j] : 0.f;
            const float h = (1.f - zg) * ng + zg * hp;
            p.hf[(((long long)td * B + row) * 2 + d) * H + j] = h;
            ((T*)p.hseq)[((long long)td * B + row) * 2 * H + d * H + j] = from_f32<T>(h);
            if (p.out_bt) ((T*)p.out_bt)[((long long)row * p.T + td) * 2 * H + d * H + j] = from_f32<T>(h);
            if (p.gates) {
                float* go = p.gates + (((long long)td * B + row) * 2 + d) * 4 * H;
                go[j] = rg; go[H + j] = zg; go[2 * H + j] = ng; go[3 * H + j] = pre[2];
            }
        }
    }
}

struct GruBwdP {
    const void* dout;    // grad wrt the layer output: element (b, t, d H + j) at dout + b do_bs + t do_ts + d H + j
    int dout_dtype;
    long long do_bs, do_ts;
    void* dgi;           // [T][B][2][3H] gradient wrt gi = (dr, dz, dn) pre-activation            compute dtype
    void* dgh;           // [T][B][2][3H] gradient wrt gh = (dr, dz, dn o r)                        compute dtype
    const void* whhT;    // [2][H][3H]   W_hh^T
    const float* gates;  // [T][B][2][4H]
    const float* hf;     // [T][B][2][H]
    float* dhc;          // [2][B][H] running direct-path gradient dh o z (in place)
    int T, B, H, s;
};

template <typename T>
__global__ __launch_bounds__(256) void gru_bwd_step(const GruBwdP p) {
    constexpr int KS = GFrag<T>::KS, PER = GFrag<T>::PER;
    __shared__ float red[4][GMT][4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, r = lane & 15, g = lane >> 4;
    const int d = blockIdx.y, j0 = blockIdx.x * 16, mbase = blockIdx.z * 64;
    const int H = p.H, B = p.B;
    const int td = d == 0 ? p.T - 1 - p.s : p.s;            // the backward visits the chain in reverse
    const int tn = d == 0 ? td + 1 : td - 1;                // time handled at the previous backward step
    const int tp = d == 0 ? td - 1 : td + 1;                // forward-previous time (h_prev)
    int nmt = (B - mbase + 15) / 16;
    if (nmt > GMT) nmt = GMT;
    if (p.s > 0) {                                           // rec = dgh[tn] x W_hh  (K = 3H, a quarter per wavefront)
        f32x4 acc[GMT];
#pragma unroll
        for (int mt = 0; mt < GMT; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
        const T* A = (const T*)p.dgh + (((long long)tn * B) * 2 + d) * 3 * H;                 // row stride 6H
        const T* bp = (const T*)p.whhT + (long long)d * H * 3 * H + (long long)(j0 + r) * 3 * H + PER * g;
        const int kq = 3 * H / 4;
        for (int k0 = w * kq; k0 < (w + 1) * kq; k0 += KS) {
            const auto b = g_ld<T>(bp + k0, true);
#pragma unroll
            for (int mt = 0; mt < GMT; ++mt) {
                if (mt < nmt) {
                    const int row = mbase + mt * 16 + r;
                    const auto a = g_ld<T>(A + (long long)row * 6 * H + k0 + PER * g, row < B);
                    g_mma(acc[mt], a, b);
                }
            }
        }
#pragma unroll
        for (int mt = 0; mt < GMT; ++mt)
#pragma unroll
            for (int e = 0; e < 4; ++e) red[w][mt][e][lane] = acc[mt][e];
    }
    __syncthreads();
    const int mt = w;
    if (mt >= nmt) return;
    const int j = j0 + r;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int row = mbase + mt * 16 + 4 * g + e;
        if (row < B) {
            const float rec = p.s > 0 ? red[0][mt][e][lane] + red[1][mt][e][lane] + red[2][mt][e][lane] + red[3][mt][e][lane] : 0.f;
            float* dcp = p.dhc + ((long long)d * B + row) * H + j;
            const float dh = ld_any(p.dout, (long long)row * p.do_bs + (long long)td * p.do_ts + d * H + j, p.dout_dtype) + rec + (p.s > 0 ? *dcp : 0.f);
            const float* gs = p.gates + (((long long)td * B + row) * 2 + d) * 4 * H;
            const float rg = gs[j], zg = gs[H + j], ng = gs[2 * H + j], hn = gs[3 * H + j];
            const bool has_prev = d == 0 ? td > 0 : td < p.T - 1;
            const float hp = has_prev ? p.hf[(((long long)tp * B + row) * 2 + d) * H + j] : 0.f;
            const float dnp = dh * (1.f - zg) * (1.f - ng * ng);
            const float dzp = dh * (hp - ng) * zg * (1.f - zg);
            const float drp = dnp * hn * rg * (1.f - rg);
            *dcp = dh * zg;
            T* gi = (T*)p.dgi + (((long long)td * B + row) * 2 + d) * 3 * H;
            T* gh = (T*)p.dgh + (((long long)td * B + row) * 2 + d) * 3 * H;
            gi[j] = from_f32<T>(drp); gi[H + j] = from_f32<T>(dzp); gi[2 * H + j] = from_f32<T>(dnp);
            gh[j] = from_f32<T>(drp); gh[H + j] = from_f32<T>(dzp); gh[2 * H + j] = from_f32<T>(dnp * rg);
        }
    }
}

}  // namespace

extern "C" int av_nchw_to_nhwc(const float* in, void* out, int out_dtype, long long N, int C, int H, int W, int Cp, void* stream) {
    AV_CHECK(in && out && N > 0 && C > 0 && H > 0 && W > 0 && Cp >= C, "av_nchw_to_nhwc: bad args (N=%lld C=%d H=%d W=%d Cp=%d)", N, C, H, W, Cp);
    AV_CHECK(out_dtype == AV_F32 || out_dtype == AV_BF16, "av_nchw_to_nhwc: bad dtype %d", out_dtype);
    const long long npix = N * H * W;
    if (out_dtype == AV_F32) hipLaunchKernelGGL(nchw_to_nhwc_kernel<float>, dim3(lg_grid(npix)), dim3(256), 0, (hipStream_t)stream, in, (float*)out, npix, C, H * W, Cp);
    else hipLaunchKernelGGL(nchw_to_nhwc_kernel<bf16_t>, dim3(lg_grid(npix)), dim3(256), 0, (hipStream_t)stream, in, (bf16_t*)out, npix, C, H * W, Cp);
    AV_LAUNCH_CHECK();
    return AV_OK;
}

extern "C" int av_relu_maxpool2_fwd(const void* x, void* y, int dtype, long long N, int H, int W, int C, int nchw_out, void* stream) {
    AV_CHECK(x && y && N > 0 && H > 0 && W > 0 && C > 0 && H % 2 == 0 && W % 2 == 0, "av_relu_maxpool2_fwd: bad args (N=%lld H=%d W=%d C=%d; even H, W)", N, H, W, C);
    AV_CHECK(dtype == AV_F32 || dtype == AV_BF16, "av_relu_maxpool2_fwd: bad dtype %d", dtype);
    const long long nout = N * (H / 2) * (W / 2) * C;
    if (dtype == AV_F32) hipLaunchKernelGGL(relu_pool_fwd_kernel<float>, dim3(lg_grid(nout)), dim3(256), 0, (hipStream_t)stream, (const float*)x, (float*)y, nout, H, W, C, nchw_out);
    else hipLaunchKernelGGL(relu_pool_fwd_kernel<bf16_t>, dim3(lg_grid(nout)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (bf16_t*)y, nout, H, W, C, nchw_out);
    AV_LAUNCH_CHECK();
    return AV_OK;
}

extern "C" int av_relu_maxpool2_bwd(const void* x, const void* dy, void* dx, int dtype, long long N, int H, int W, int C, int nchw_dy, void* stream) {
    AV_CHECK(x && dy && dx && N > 0 && H > 0 && W > 0 && C > 0 && H % 2 == 0 && W % 2 == 0, "av_relu_maxpool2_bwd: bad args (N=%lld H=%d W=%d C=%d; even H, W)", N, H, W, C);
    AV_CHECK(dtype == AV_F32 || dtype == AV_BF16, "av_relu_maxpool2_bwd: bad dtype %d", dtype);
    const long long nout = N * (H / 2) * (W / 2) * C;
    if (dtype == AV_F32) hipLaunchKernelGGL(relu_pool_bwd_kernel<float>, dim3(lg_grid(nout)), dim3(256), 0, (hipStream_t)stream, (const float*)x, (const float*)dy, (float*)dx, nout, H, W, C, nchw_dy);
    else hipLaunchKernelGGL(relu_pool_bwd_kernel<bf16_t>, dim3(lg_grid(nout)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (const bf16_t*)dy, (bf16_t*)dx, nout, H, W, C, nchw_dy);
    AV_LAUNCH_CHECK();
    return AV_OK;
}

extern "C" int av_im2col3(const void* x, void* cols, int dtype, long long N, int H, int W, int C, void* stream) {
    AV_CHECK(x && cols && N > 0 && H > 0 && W > 0 && C > 0, "av_im2col3: bad args (N=%lld H=%d W=%d C=%d)", N, H, W, C);
    AV_CHECK(dtype == AV_F32 || dtype == AV_BF16, "av_im2col3: bad dtype %d", dtype);
    const long long n = N * H * W * 9 * C;
    if (dtype == AV_F32) hipLaunchKernelGGL(im2col3_kernel<float>, dim3(lg_grid(n)), dim3(256), 0, (hipStream_t)stream, (const float*)x, (float*)cols, n, H, W, C);
    else hipLaunchKernelGGL(im2col3_kernel<bf16_t>, dim3(lg_grid(n)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (bf16_t*)cols, n, H, W, C);
    AV_LAUNCH_CHECK();
    return AV_OK;
}

extern "C" int av_gru_fwd_step(const float* gx, const void* whh, const float* bhh, void* hseq, float* hf, float* gates, void* out_bt, int dtype, int T,
                               int B, int H, int s, void* stream) {
    AV_CHECK(gx && whh && bhh && hseq && hf, "av_gru_fwd_step: null pointer");
    AV_CHECK(H > 0 && H % 128 == 0 && B > 0 && T > 0 && s >= 0 && s < T, "av_gru_fwd_step: bad shape T=%d B=%d H=%d s=%d (H %% 128 == 0)", T, B, H, s);
    AV_CHECK(dtype == AV_F32 || dtype == AV_BF16, "av_gru_fwd_step: bad dtype %d", dtype);
    GruFwdP p{gx, whh, bhh, hseq, hf, gates, out_bt, T, B, H, s};
    dim3 grid((unsigned)(H / 16), 2, (unsigned)((B + 63) / 64));
    if (dtype == AV_F32) hipLaunchKernelGGL(gru_fwd_step<float>, grid, dim3(256), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(gru_fwd_step<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, p);
    AV_LAUNCH_CHECK();
    return AV_OK;
}

extern "C" int av_gru_bwd_step(const void* dout, int dout_dtype, long long do_bs, long long do_ts, void* dgi, void* dgh, const void* whhT,
                               const float* gates, const float* hf, float* dhc, int dtype, int T, int B, int H, int s, void* stream) {
    AV_CHECK(dout && dgi && dgh && whhT && gates && hf && dhc, "av_gru_bwd_step: null pointer");
    AV_CHECK(H > 0 && H % 128 == 0 && B > 0 && T > 0 && s >= 0 && s < T, "av_gru_bwd_step: bad shape T=%d B=%d H=%d s=%d (H %% 128 == 0)", T, B, H, s);
    AV_CHECK(dtype == AV_F32 || dtype == AV_BF16, "av_gru_bwd_step: bad dtype %d", dtype);
    AV_CHECK(dout_dtype == AV_F32 || dout_dtype == AV_BF16, "av_gru_bwd_step: bad dout dtype %d", dout_dtype);
    GruBwdP p{dout, dout_dtype, do_bs, do_ts, dgi, dgh, whhT, gates, hf, dhc, T, B, H, s};
    dim3 grid((unsigned)(H / 16), 2, (unsigned)((B + 63) / 64));
    if (dtype == AV_F32) hipLaunchKernelGGL(gru_bwd_step<float>, grid, dim3(256), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(gru_bwd_step<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, p);
    AV_LAUNCH_CHECK();
    return AV_OK;
}
