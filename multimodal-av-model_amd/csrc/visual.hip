// Lip-frame encoder glue around the implicit-GEMM convolutions (model/encoder.py:6-75), channel-last (NHWC):
//   * train-mode BatchNorm statistics: finalize the per-block column partials the conv GEMM epilogue wrote
//     (batch mean / biased variance -> scale, shift; running-stat update with momentum 0.1 and the unbiased
//     variance — the side effect of calling .train() on the frozen encoder, model/trainer.py:54),
//   * fused BN-apply + residual (+ its own BN) + PReLU,
//   * fused BN-apply + PReLU + MaxPool3d((1,3,3),(1,2,2),(0,1,1)) of the 3-D front-end,
//   * AdaptiveAvgPool2d(1).
// All of them are HBM-bound single-pass kernels with 8/16-byte accesses along the channel dimension.
#include "av_common.h"

namespace {

constexpr int BNR_ROWS = 64;     // rows of partials per workgroup: nblk / 64 workgroups keep the short reduction off a single-wave tail
// stage 1: column sums of the [nblk][2C] partial matrix in double (coalesced over channels, f64 atomics to ws[2C])
__global__ __launch_bounds__(256) void bn_reduce_kernel(const float* __restrict__ part, int nblk, int C2, double* __restrict__ ws) {
    __shared__ double red[4][64];
    const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cx;
    const int r0 = blockIdx.y * BNR_ROWS;
    int r1 = r0 + BNR_ROWS;
    if (r1 > nblk) r1 = nblk;
    double s = 0.0;
    if (c < C2)
        for (int r = r0 + ry; r < r1; r += 4) s += (double)part[(long long)r * C2 + c];
    red[ry][cx] = s;
    __syncthreads();
    if (ry == 0 && c < C2) atomicAdd(ws + c, red[0][cx] + red[1][cx] + red[2][cx] + red[3][cx]);
}

__global__ void bn_finalize_kernel(double* __restrict__ ws, double count, const float* __restrict__ gamma,
                                   const float* __restrict__ beta, float* __restrict__ rmean, float* __restrict__ rvar, float momentum,
                                   float eps, int training, float* __restrict__ scale, float* __restrict__ shift, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float sc, sh;
    if (training) {
        const double mean = ws[c] / count;
        double var = ws[C + c] / count - mean * mean;
        ws[c] = 0.0; ws[C + c] = 0.0;                          // leave the accumulators zeroed for the next layer (ws_zeroed protocol)
        if (var < 0.0) var = 0.0;
        sc = gamma[c] * (float)(1.0 / sqrt(var + (double)eps));
        sh = beta[c] - (float)mean * sc;
        if (rmean) {
            const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
            rmean[c] = (1.f - momentum) * rmean[c] + momentum * (float)mean;
            rvar[c] = (1.f - momentum) * rvar[c] + momentum * (float)unb;
        }
    } else {
        sc = gamma[c] * rsqrtf(rvar[c] + eps);
        sh = beta[c] - rmean[c] * sc;
    }
    scale[c] = sc;
    shift[c] = sh;
}

// train mode, one launch: stage 1 as above, then the LAST workgroup to arrive (ticket counter stored behind the 2C accumulators)
// finalizes every channel - the accumulators are read back with agent-scope atomic loads (the f64 atomics live in L2) and left
// zeroed, the counter returns to 0 - so a BatchNorm costs one short launch instead of two.
__global__ __launch_bounds__(256) void bn_reduce_finalize_kernel(const float* __restrict__ part, int nblk, double* __restrict__ ws, double count,
                                                                 const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                 float* __restrict__ rmean, float* __restrict__ rvar, float momentum, float eps,
                                                                 float* __restrict__ scale, float* __restrict__ shift, int C) {
    __shared__ double red[4][64];
    __shared__ int s_last;
    const int C2 = 2 * C;
    const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cx;
    const int r0 = blockIdx.y * BNR_ROWS;
    int r1 = r0 + BNR_ROWS;
    if (r1 > nblk) r1 = nblk;
    double s = 0.0;
    if (c < C2)
        for (int r = r0 + ry; r < r1; r += 4) s += (double)part[(long long)r * C2 + c];
    red[ry][cx] = s;
    __syncthreads();
    unsigned* ticket = (unsigned*)(ws + C2);
    if (ry == 0) {                                             // wavefront 0 issues the accumulator atomics AND the ticket: in order, and
        if (c < C2) atomicAdd(ws + c, red[0][cx] + red[1][cx] + red[2][cx] + red[3][cx]);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // ... acknowledged by L2 before the ticket leaves - no fence (a release
        if (threadIdx.x == 0) s_last = atomicAdd(ticket, 1u) == gridDim.x * gridDim.y - 1;      // fence writes the whole dirty L2 back)
    }
    __syncthreads();
    if (!s_last) return;
    for (int ch = threadIdx.x; ch < C; ch += 256) {
        const double sum = __hip_atomic_load(ws + ch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const double sq = __hip_atomic_load(ws + C + ch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ws[ch] = 0.0; ws[C + ch] = 0.0;                        // zeroed for the next layer (ws_zeroed protocol)
        const double mean = sum / count;
        double var = sq / count - mean * mean;
        if (var < 0.0) var = 0.0;
        const float sc = gamma[ch] * (float)(1.0 / sqrt(var + (double)eps));
        scale[ch] = sc;
        shift[ch] = beta[ch] - (float)mean * sc;
        if (rmean) {
            const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
            rmean[ch] = (1.f - momentum) * rmean[ch] + momentum * (float)mean;
            rvar[ch] = (1.f - momentum) * rvar[ch] + momentum * (float)unb;
        }
    }
    if (threadIdx.x == 0) *ticket = 0u;
}

template <typename T>
__global__ void bn_act_kernel(const T* __restrict__ x, const float* __restrict__ scale, const float* __restrict__ shift,
                              const T* __restrict__ res, const float* __restrict__ rscale, const float* __restrict__ rshift,
                              const float* __restrict__ slope, T* __restrict__ out, long long n, int C) {
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(e % C);
        float v = to_f32<T>(x[e]) * scale[c] + shift[c];
        if (res) {
            const float rv = to_f32<T>(res[e]);
            v += rscale ? rv * rscale[c] + rshift[c] : rv;
        }
        if (slope) v = v >= 0.f ? v : v * slope[c];
        out[e] = from_f32<T>(v);
    }
}

template <typename T>
__global__ void bn_prelu_maxpool_kernel(const T* __restrict__ x, const float* __restrict__ scale, const float* __restrict__ shift,
                                        const float* __restrict__ slope, T* __restrict__ out, long long N, int H, int W, int C) {
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    const long long tot = N * Ho * Wo * C;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < tot; e += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(e % C);
        long long q = e / C;
        const int ox = (int)(q % Wo); q /= Wo;
        const int oy = (int)(q % Ho);
        const long long img = q / Ho;
        const float sc = scale[c], sh = shift[c], sl = slope[c];
        float m = -INFINITY;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            const int iy = oy * 2 - 1 + dy;
            if (iy < 0 || iy >= H) continue;
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int ix = ox * 2 - 1 + dx;
                if (ix < 0 || ix >= W) continue;
                float v = to_f32<T>(x[((img * H + iy) * W + ix) * C + c]) * sc + sh;
                v = v >= 0.f ? v : v * sl;
                m = fmaxf(m, v);
            }
        }
        out[e] = from_f32<T>(m);
    }
}

template <typename T>
__global__ void avgpool_kernel(const T* __restrict__ x, float* __restrict__ out, long long N, int HW, int C, int pos_major) {
    const long long tot = N * C;
    const int FB = pos_major;                                                  // images per position-major block (0: frame-major)
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < tot; e += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(e % C);
        const long long img = e / C;
        float s = 0.f;
        if (FB > 0) {
            const long long blk = img / FB, fi = img - blk * FB;
            for (int p = 0; p < HW; ++p) s += to_f32<T>(x[((blk * HW + p) * FB + fi) * C + c]);
        } else {
            for (int p = 0; p < HW; ++p) s += to_f32<T>(x[(img * HW + p) * C + c]);
        }
        out[e] = s / (float)HW;
    }
}

// ---- bf16 vector forms: one thread = 8 consecutive channels of one pixel (16-B accesses, C % 8 == 0) -------------
template <bool RES>
__global__ __launch_bounds__(256) void bn_act_vec_kernel(const bf16x8* __restrict__ x, const float* __restrict__ scale,
                                                         const float* __restrict__ shift, const bf16x8* __restrict__ res,
                                                         const float* __restrict__ rscale, const float* __restrict__ rshift,
                                                         const float* __restrict__ slope, bf16x8* __restrict__ out, long long n8, int C8) {
    const long long stride = (long long)gridDim.x * blockDim.x;      // multiple of C8 (host guarantees) => fixed channel group
    long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int cg = (int)(e % C8) * 8;
    float sc[8], sh[8], rs[8], rb[8], sl[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        sc[i] = scale[cg + i]; sh[i] = shift[cg + i];
        rs[i] = (RES && rscale) ? rscale[cg + i] : 1.f; rb[i] = (RES && rscale) ? rshift[cg + i] : 0.f;
        sl[i] = slope ? slope[cg + i] : 1.f;
    }
    auto apply = [&](const bf16x8& xv, const bf16x8& rv) {
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = (float)xv[i] * sc[i] + sh[i];
        if (RES) {
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] += (float)rv[i] * rs[i] + rb[i];
        }
        bf16x8 o;
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = (bf16_t)(v[i] >= 0.f ? v[i] : v[i] * sl[i]);
        return o;
    };
    // four independent 16-B loads per operand in flight per thread: one load per iteration leaves a CU with 32 KiB in flight, which
    // is latency-bound (3.3 TB/s) on the short per-thread loops of the small layers
    constexpr int U = 4;
    for (; e < n8; e += U * stride) {
        bf16x8 xv[U], rv[U];
        long long idx[U];
#pragma unroll
        for (int u = 0; u < U; ++u) idx[u] = e + u * stride < n8 ? e + u * stride : e;        // clamped: unconditional loads, predicated stores
#pragma unroll
        for (int u = 0; u < U; ++u) xv[u] = x[idx[u]];
#pragma unroll
        for (int u = 0; u < U; ++u) rv[u] = RES ? res[idx[u]] : xv[u];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const bf16x8 o = apply(xv[u], rv[u]);
            if (u == 0 || e + u * stride < n8) out[e + u * stride] = o;
        }
    }
}

__global__ __launch_bounds__(256) void bn_prelu_maxpool_vec_kernel(const bf16x8* __restrict__ x, const float* __restrict__ scale,
                                                                   const float* __restrict__ shift, const float* __restrict__ slope,
                                                                   bf16x8* __restrict__ out, long long N, int H, int W, int C8) {
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const long long tot = N * Ho * Wo * C8;
    const long long stride = (long long)gridDim.x * blockDim.x;
    long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int cgi = (int)(e % C8), cg = cgi * 8;
    float sc[8], sh[8], sl[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { sc[i] = scale[cg + i]; sh[i] = shift[cg + i]; sl[i] = slope[cg + i]; }
    for (; e < tot; e += stride) {
        long long q = e / C8;
        const int ox = (int)(q % Wo); q /= Wo;
        const int oy = (int)(q % Ho);
        const long long img = q / Ho;
        // the 9 taps are loaded unconditionally from clamped (valid) coordinates and masked afterwards: a predicated load compiles
        // to a branch + vmcnt(0), i.e. nine serial memory round trips per output
        bf16x8 tap[9];
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            const int iy = oy * 2 - 1 + dy, iyc = iy < 0 ? 0 : (iy > H - 1 ? H - 1 : iy);
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int ix = ox * 2 - 1 + dx, ixc = ix < 0 ? 0 : (ix > W - 1 ? W - 1 : ix);
                tap[dy * 3 + dx] = x[((img * H + iyc) * W + ixc) * C8 + cgi];
            }
        }
        float m[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) m[i] = -INFINITY;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            const int iy = oy * 2 - 1 + dy;
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int ix = ox * 2 - 1 + dx;
                const bool ok = iy >= 0 && iy < H && ix >= 0 && ix < W;     // a clamped duplicate never changes the max; masked anyway
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    float v = (float)tap[dy * 3 + dx][i] * sc[i] + sh[i];
                    v = v >= 0.f ? v : v * sl[i];
                    m[i] = ok ? fmaxf(m[i], v) : m[i];
                }
            }
        }
        bf16x8 o;
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = (bf16_t)m[i];
        out[e] = o;
    }
}

inline int vec_grid(long long n8, int C8, int cap = 2048) {          // blocks of 256 threads; total threads a multiple of C8 (C8 | 256 here)
    long long b = (n8 + 255) / 256;
    return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}
// BN-apply: few, long-lived threads (2 blocks per CU, 4 x 16 B per operand in flight each).  The per-thread start-up (channel-group
// index, 40 parameter loads) costs ~12 ns per block: 24 us of a 27 us launch at 2048 blocks on the 59 MB tensors of layer4.  The
// 9-tap pooling kernel does more work per output and keeps 2048.
inline int bn_act_cap() {
    static const int cap = [] { const char* e = getenv("AVAMD_EW_BLOCKS"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 512; }();
    return cap;
}

inline int ew_grid(long long n) {
    long long b = (n + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 8192 ? 8192 : b));
}

}  // namespace

extern "C" int av_bn_finalize(const float* partial, int nblk, long long count, const float* gamma, const float* beta, float* running_mean,
                              float* running_var, float momentum, float eps, int training, float* scale, float* shift, int C, double* ws,
                              int ws_zeroed, void* stream) {
    AV_CHECK(gamma && beta && scale && shift && C > 0, "av_bn_finalize: null pointer");
    AV_CHECK(training ? (partial != nullptr && nblk > 0 && count > 0 && ws != nullptr) : (running_mean && running_var), "av_bn_finalize: missing statistics input");
    hipStream_t st = (hipStream_t)stream;
    if (training) {
        if (!ws_zeroed && hipMemsetAsync(ws, 0, sizeof(double) * (2 * C + 1), st) != hipSuccess) { av_set_error("av_bn_finalize: memset failed"); return AV_ERR_LAUNCH; }
        hipLaunchKernelGGL(bn_reduce_finalize_kernel, dim3((2 * C + 63) / 64, (nblk + BNR_ROWS - 1) / BNR_ROWS), dim3(256), 0, st, partial, nblk, ws,
                           (double)count, gamma, beta, running_mean, running_var, momentum, eps, scale, shift, C);
    } else {
        hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 63) / 64), dim3(64), 0, st, ws, (double)count, gamma, beta,
                           running_mean, running_var, momentum, eps, training, scale, shift, C);
    }
    AV_LAUNCH_CHECK();
    return AV_OK;
}

extern "C" int av_bn_act(const void* x, const float* scale, const float* shift, const void* res, const float* rscale, const float* rshift,
                         const float* slope, void* out, int dtype, long long n, int C, void* stream) {
    AV_CHECK(x && scale && shift && out && C > 0, "av_bn_act: null pointer");
    if (n == 0) return AV_OK;
    const bool vec = dtype == AV_BF16 && C % 8 == 0 && 256 % (C / 8) == 0 && ((uintptr_t)x % 16 == 0) && ((uintptr_t)out % 16 == 0) &&
                     (!res || (uintptr_t)res % 16 == 0);
    if (vec)
        if (res) hipLaunchKernelGGL(bn_act_vec_kernel<true>, dim3(vec_grid(n / 8, C / 8, bn_act_cap())), dim3(256), 0, (hipStream_t)stream, (const bf16x8*)x, scale, shift,
                                    (const bf16x8*)res, rscale, rshift, slope, (bf16x8*)out, n / 8, C / 8);
        else hipLaunchKernelGGL(bn_act_vec_kernel<false>, dim3(vec_grid(n / 8, C / 8, bn_act_cap())), dim3(256), 0, (hipStream_t)stream, (const bf16x8*)x, scale, shift,
                                (const bf16x8*)res, rscale, rshift, slope, (bf16x8*)out, n / 8, C / 8);
    else if (dtype == AV_F32)
        hipLaunchKernelGGL(bn_act_kernel<float>, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, (const float*)x, scale, shift, (const float*)res, rscale, rshift, slope, (float*)out, n, C);
    else
        hipLaunchKernelGGL(bn_act_kernel<bf16_t>, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, scale, shift, (const bf16_t*)res, rscale, rshift, slope, (bf16_t*)out, n, C);
    AV_LAUNCH_CHECK();
    return AV_OK;
}

extern "C" int av_bn_prelu_maxpool(const void* x, const float* scale, const float* shift, const float* slope, void* out, int dtype,
                                   long long N, int H, int W, int C, void* stream) {
    AV_CHECK(x && scale && shift && slope && out, "av_bn_prelu_maxpool: null pointer");
    if (N == 0) return AV_OK;
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const long long tot = N * Ho * Wo * C;
    const bool vec = dtype == AV_BF16 && C % 8 == 0 && 256 % (C / 8) == 0 && ((uintptr_t)x % 16 == 0) && ((uintptr_t)out % 16 == 0);
    if (vec)
        hipLaunchKernelGGL(bn_prelu_maxpool_vec_kernel, dim3(vec_grid(tot / 8, C / 8)), dim3(256), 0, (hipStream_t)stream, (const bf16x8*)x, scale,
                           shift, slope, (bf16x8*)out, N, H, W, C / 8);
    else if (dtype == AV_F32)
        hipLaunchKernelGGL(bn_prelu_maxpool_kernel<float>, dim3(ew_grid(tot)), dim3(256), 0, (hipStream_t)stream, (const float*)x, scale, shift, slope, (float*)out, N, H, W, C);
    else
        hipLaunchKernelGGL(bn_prelu_maxpool_kernel<bf16_t>, dim3(ew_grid(tot)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, scale, shift, slope, (bf16_t*)out, N, H, W, C);
    AV_LAUNCH_CHECK();
    return AV_OK;
}

extern "C" int av_avgpool(const void* x, int dtype, float* out, long long N, int HW, int C, int pos_major, void* stream) {
    AV_CHECK(x && out && HW > 0 && C > 0 && pos_major >= 0 && (pos_major == 0 || N % pos_major == 0), "av_avgpool: bad args");
    if (N == 0) return AV_OK;
    if (dtype == AV_F32) hipLaunchKernelGGL(avgpool_kernel<float>, dim3(ew_grid(N * C)), dim3(256), 0, (hipStream_t)stream, (const float*)x, out, N, HW, C, pos_major);
    else hipLaunchKernelGGL(avgpool_kernel<bf16_t>, dim3(ew_grid(N * C)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, out, N, HW, C, pos_major);
    AV_LAUNCH_CHECK();
    return AV_OK;
}
