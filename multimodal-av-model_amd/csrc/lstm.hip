// Bidirectional LSTM time-step kernels (nn.LSTM(512,512,2,bidirectional), model/fusion_module.py:21-27,64).
// gates i,f,g,o; c = f*c + i*g; h = o*tanh(c); zero initial state; both directions advance in ONE launch
// (blockIdx.y = direction): at step s the forward chain is at time s, the reverse chain at time T-1-s.
// The input projections x W_ih^T + b_ih + b_hh of all time steps are one big MFMA GEMM done beforehand (gx);
// here each wavefront computes h_prev[16 rows] x W_hh^T for 16 hidden units x 4 gates on MFMA 16x16 tiles with
// operands streamed straight from L2 (W_hh slice = 64 KB bf16 per workgroup, shared by its 4 waves) and applies
// the cell update in the accumulator layout (all four gates of a (row, unit) pair sit in one lane).
// Internal sequence buffers are time-major [T][B][...] so that the weight-gradient GEMMs over all steps see the
// one-step shift between dgates and h as a plain row offset.
#include "av_common.h"

namespace {

struct LstmFwdP {
    const float* gx;     // [T][B][2][4H]
    const void* whh;     // [2][4H][H]
    void* hseq;          // [T][B][2H]
    float* cseq;         // [T][B][2][H]
    void* gates;         // [T][B][2][4H] post-activation (optional)
    void* out_bt;        // [B][T][2H] optional batch-major copy
    int T, B, H, s;
};

template <typename T> struct Frag;
template <> struct Frag<bf16_t> { typedef bf16x8 type; static constexpr int KS = 32, PER = 8; };
template <> struct Frag<float> { typedef f32x4 type; static constexpr int KS = 16, PER = 4; };

template <typename T>
__device__ __forceinline__ typename Frag<T>::type ld_frag(const T* p, bool ok) {
    typename Frag<T>::type z;
#pragma unroll
    for (int i = 0; i < Frag<T>::PER; ++i) z[i] = from_f32<T>(0.f);
    return ok ? *(const typename Frag<T>::type*)p : z;
}
__device__ __forceinline__ void mma(f32x4& acc, const bf16x8& a, const bf16x8& b) {
    acc = AV_MFMA_F32_16X16X32_LP(a, b, acc, 0, 0, 0);
}
__device__ __forceinline__ void mma(f32x4& acc, const f32x4& a, const f32x4& b) {
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[jj], b[jj], acc, 0, 0, 0);
}

// K (= H) is split over the 4 wavefronts (the step is an L2-latency chain, not a throughput problem): every wave
// accumulates a quarter of K for all row tiles, partials meet in LDS, wave w finishes row tile w.
constexpr int MAXMT = 4;   // 64 rows per workgroup (blockIdx.z selects the 64-row group)

template <typename T, int HT>
__global__ __launch_bounds__(256) void lstm_fwd_step(const LstmFwdP p) {
    constexpr int KS = Frag<T>::KS, PER = Frag<T>::PER;
    __shared__ float red[4][MAXMT][4][4][64];        // [wave][row tile][gate][acc reg][lane]  (64 KiB)
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int r = lane & 15, g = lane >> 4;
    const int d = blockIdx.y, j0 = blockIdx.x * 16, mbase = blockIdx.z * 64;
    const int H = HT > 0 ? HT : p.H, B = p.B;
    const int td = d == 0 ? p.s : p.T - 1 - p.s;           // time handled by this direction
    const int tp = d == 0 ? td - 1 : td + 1;               // previous time of the chain
    int nmt = (B - mbase + 15) / 16;
    if (nmt > MAXMT) nmt = MAXMT;
    if (p.s > 0) {
        f32x4 acc[MAXMT][4];
#pragma unroll
        for (int mt = 0; mt < MAXMT; ++mt)
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[mt][q] = f32x4{0.f, 0.f, 0.f, 0.f};
        const T* hprev = (const T*)p.hseq + ((long long)tp * B) * 2 * H + d * H;          // row stride 2H
        const T* W = (const T*)p.whh + (long long)d * 4 * H * H;
        const int kq = H / 4;
#pragma unroll
        for (int kk = 0; kk < (HT > 0 ? HT / 4 / KS : 1 << 20); ++kk) {
            const int k0 = w * kq + kk * KS;
            if (HT == 0 && k0 >= (w + 1) * kq) break;
            typename Frag<T>::type b[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) b[q] = ld_frag<T>(W + (long long)(q * H + j0 + r) * H + k0 + PER * g, true);
#pragma unroll
            for (int mt = 0; mt < MAXMT; ++mt) {
                if (mt < nmt) {
                    const int row = mbase + mt * 16 + r;
                    const auto a = ld_frag<T>(hprev + (long long)row * 2 * H + k0 + PER * g, row < B);
#pragma unroll
                    for (int q = 0; q < 4; ++q) mma(acc[mt][q], a, b[q]);
                }
            }
        }
#pragma unroll
        for (int mt = 0; mt < MAXMT; ++mt)
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e) red[w][mt][q][e][lane] = acc[mt][q][e];
    }
    __syncthreads();
    const int mt = w;                                       // wave w finishes row tile w
    if (mt >= nmt) return;
    const int j = j0 + r;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int row = mbase + mt * 16 + 4 * g + e;
        if (row < B) {
            float pre[4];
#pragma unroll
            for (int q = 0; q < 4; ++q)
                pre[q] = p.s > 0 ? red[0][mt][q][e][lane] + red[1][mt][q][e][lane] + red[2][mt][q][e][lane] + red[3][mt][q][e][lane] : 0.f;
            const float* gxr = p.gx + (((long long)td * B + row) * 2 + d) * 4 * H;
            const float ig = sigmoid_f(pre[0] + gxr[j]);
            const float fg = sigmoid_f(pre[1] + gxr[H + j]);
            const float gg = tanhf(pre[2] + gxr[2 * H + j]);
            const float og = sigmoid_f(pre[3] + gxr[3 * H + j]);
            const float cprev = p.s > 0 ? p.cseq[(((long long)tp * B + row) * 2 + d) * H + j] : 0.f;
            const float c = fg * cprev + ig * gg;
            const float h = og * tanhf(c);
            p.cseq[(((long long)td * B + row) * 2 + d) * H + j] = c;
            ((T*)p.hseq)[((long long)td * B + row) * 2 * H + d * H + j] = from_f32<T>(h);
            if (p.out_bt) ((T*)p.out_bt)[((long long)row * p.T + td) * 2 * H + d * H + j] = from_f32<T>(h);
            if (p.gates) {
                T* go = (T*)p.gates + (((long long)td * B + row) * 2 + d) * 4 * H;
                go[j] = from_f32<T>(ig); go[H + j] = from_f32<T>(fg); go[2 * H + j] = from_f32<T>(gg); go[3 * H + j] = from_f32<T>(og);
            }
        }
    }
}

struct LstmBwdP {
    const void* dout;    // grad wrt h: element (b,t,d*H+j) at dout + b*do_bs + t*do_ts + d*H + j
    int dout_dtype;
    long long do_bs, do_ts;
    void* dgates;        // [T][B][2][4H] pre-activation gate gradients (written at this step's time)
    const void* whhT;    // [2][H][4H]
    const void* gates;   // [T][B][2][4H]
    const float* cseq;   // [T][B][2][H]
    float* dc;           // [2][B][H] running cell gradient (in place)
    int T, B, H, s;
};

template <typename T, int HT>
__global__ __launch_bounds__(256) void lstm_bwd_step(const LstmBwdP p) {
    constexpr int KS = Frag<T>::KS, PER = Frag<T>::PER;
    __shared__ float red[4][MAXMT][4][64];           // [wave][row tile][acc reg][lane]
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int r = lane & 15, g = lane >> 4;
    const int d = blockIdx.y, j0 = blockIdx.x * 16, mbase = blockIdx.z * 64;
    const int H = HT > 0 ? HT : p.H, B = p.B;
    const int td = d == 0 ? p.T - 1 - p.s : p.s;           // backward visits the chain in reverse
    const int tn = d == 0 ? td + 1 : td - 1;               // time handled at the previous backward step
    const int tp = d == 0 ? td - 1 : td + 1;               // forward-previous time (c_prev)
    int nmt = (B - mbase + 15) / 16;
    if (nmt > MAXMT) nmt = MAXMT;
    if (p.s > 0) {                                          // dh_rec = dgates[tn] x W_hh   (K = 4H, a quarter per wave)
        f32x4 acc[MAXMT];
#pragma unroll
        for (int mt = 0; mt < MAXMT; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
        const T* A = (const T*)p.dgates + (((long long)tn * B) * 2 + d) * 4 * H;             // row stride 8H
        const T* Wt = (const T*)p.whhT + (long long)d * H * 4 * H;
        const T* bp = Wt + (long long)(j0 + r) * 4 * H + PER * g;
        const int kq = H;                                   // 4H / 4
#pragma unroll
        for (int kk = 0; kk < (HT > 0 ? HT / KS : 1 << 20); ++kk) {
            const int k0 = w * kq + kk * KS;
            if (HT == 0 && k0 >= (w + 1) * kq) break;
            const auto b = ld_frag<T>(bp + k0, true);
#pragma unroll
            for (int mt = 0; mt < MAXMT; ++mt) {
                if (mt < nmt) {
                    const int row = mbase + mt * 16 + r;
                    const auto a = ld_frag<T>(A + (long long)row * 8 * H + k0 + PER * g, row < B);
                    mma(acc[mt], a, b);
                }
            }
        }
#pragma unroll
        for (int mt = 0; mt < MAXMT; ++mt)
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
            const float dh = ld_any(p.dout, (long long)row * p.do_bs + (long long)td * p.do_ts + d * H + j, p.dout_dtype) + rec;
            const T* gs = (const T*)p.gates + (((long long)td * B + row) * 2 + d) * 4 * H;
            const float ig = to_f32<T>(gs[j]), fg = to_f32<T>(gs[H + j]), gg = to_f32<T>(gs[2 * H + j]), og = to_f32<T>(gs[3 * H + j]);
            const float c = p.cseq[(((long long)td * B + row) * 2 + d) * H + j];
            const bool has_prev = d == 0 ? td > 0 : td < p.T - 1;
            const float cprev = has_prev ? p.cseq[(((long long)tp * B + row) * 2 + d) * H + j] : 0.f;
            const float tc = tanhf(c);
            float* dcp = p.dc + ((long long)d * B + row) * H + j;
            const float dcs = (p.s > 0 ? *dcp : 0.f) + dh * og * (1.f - tc * tc);
            *dcp = dcs * fg;
            T* dg = (T*)p.dgates + (((long long)td * B + row) * 2 + d) * 4 * H;
            dg[j] = from_f32<T>(dcs * gg * ig * (1.f - ig));
            dg[H + j] = from_f32<T>(dcs * cprev * fg * (1.f - fg));
            dg[2 * H + j] = from_f32<T>(dcs * ig * (1.f - gg * gg));
            dg[3 * H + j] = from_f32<T>(dh * tc * og * (1.f - og));
        }
    }
}

}  // namespace

extern "C" int av_lstm_fwd_step(const float* gx, const void* whh, void* hseq, float* cseq, void* gates, void* out_bt, int dtype, int T,
                                int B, int H, int s, void* stream) {
    AV_CHECK(gx && whh && hseq && cseq, "av_lstm_fwd_step: null pointer");
    AV_CHECK(H > 0 && H % 128 == 0 && B > 0 && T > 0 && s >= 0 && s < T, "av_lstm_fwd_step: bad shape T=%d B=%d H=%d s=%d (H %% 128 == 0)", T, B, H, s);
    AV_CHECK(dtype == AV_F32 || dtype == AV_BF16, "av_lstm_fwd_step: bad dtype %d", dtype);
    LstmFwdP p{gx, whh, hseq, cseq, gates, out_bt, T, B, H, s};
    dim3 grid((unsigned)(H / 16), 2, (unsigned)((B + 63) / 64));
    if (dtype == AV_F32) hipLaunchKernelGGL((lstm_fwd_step<float, 0>), grid, dim3(256), 0, (hipStream_t)stream, p);
    else if (H == 512) hipLaunchKernelGGL((lstm_fwd_step<bf16_t, 512>), grid, dim3(256), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL((lstm_fwd_step<bf16_t, 0>), grid, dim3(256), 0, (hipStream_t)stream, p);
    AV_LAUNCH_CHECK();
    return AV_OK;
}

extern "C" int av_lstm_bwd_step(const void* dout, int dout_dtype, long long do_bs, long long do_ts, void* dgates, const void* whhT,
                                const void* gates, const float* cseq, float* dc, int dtype, int T, int B, int H, int s, void* stream) {
    AV_CHECK(dout && dgates && whhT && gates && cseq && dc, "av_lstm_bwd_step: null pointer");
    AV_CHECK(H > 0 && H % 128 == 0 && B > 0 && T > 0 && s >= 0 && s < T, "av_lstm_bwd_step: bad shape T=%d B=%d H=%d s=%d", T, B, H, s);
    AV_CHECK(dtype == AV_F32 || dtype == AV_BF16, "av_lstm_bwd_step: bad dtype %d", dtype);
    LstmBwdP p{dout, dout_dtype, do_bs, do_ts, dgates, whhT, gates, cseq, dc, T, B, H, s};
    dim3 grid((unsigned)(H / 16), 2, (unsigned)((B + 63) / 64));
    if (dtype == AV_F32) hipLaunchKernelGGL((lstm_bwd_step<float, 0>), grid, dim3(256), 0, (hipStream_t)stream, p);
    else if (H == 512) hipLaunchKernelGGL((lstm_bwd_step<bf16_t, 512>), grid, dim3(256), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL((lstm_bwd_step<bf16_t, 0>), grid, dim3(256), 0, (hipStream_t)stream, p);
    AV_LAUNCH_CHECK();
    return AV_OK;
}
