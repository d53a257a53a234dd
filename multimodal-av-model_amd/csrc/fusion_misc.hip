// Data-dependent glue of CrossAttentionFusion.forward (model/fusion_module.py:40-55,66) without host syncs:
//   * nearest down-sampling of the speaker mask to the encoder frame rate (model/trainer.py:98,102),
//   * per-item compaction of the speech frames (mask in {1,2}), zero-padded to the batch maximum,
//   * linear (align_corners=True) resampling of the compacted frames to the T_v lip frames + nearest resampling
//     of the mask, CTC input_lengths = count(mask != 0),
//   * the exact backward of the gather + lerp (deterministic, no atomics),
//   * row permute / gather / scatter helpers.
#include "av_common.h"

namespace {

__global__ void mask_down_kernel(const long long* __restrict__ m, long long* __restrict__ out, int B, int Tin, int Tout) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * Tout) return;
    const int b = i / Tout, t = i - b * Tout;
    const float scale = (float)Tin / (float)Tout;            // F.interpolate(mode='nearest') legacy law
    int src = (int)floorf((float)t * scale);
    if (src > Tin - 1) src = Tin - 1;
    out[i] = m[(long long)b * Tin + src];
}

// idx[b][0..cnt) = positions with mask in {1,2}, in order; cnt[b]; tmax = max_b cnt[b]
__global__ __launch_bounds__(256) void compact_kernel(const long long* __restrict__ mask, int* __restrict__ idx, int* __restrict__ cnt,
                                                      int* __restrict__ tmax, int Ta, int gsize) {
    __shared__ int wsum[4];
    __shared__ int base_s;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (tid == 0) base_s = 0;
    __syncthreads();
    for (int t0 = 0; t0 < Ta; t0 += 256) {
        const int t = t0 + tid;
        const long long mv = t < Ta ? mask[(long long)b * Ta + t] : 0;
        const bool f = (mv == 1 || mv == 2);
        const unsigned long long bal = __ballot(f);
        const int before = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) wsum[w] = __popcll(bal);
        __syncthreads();
        int off = base_s;
        for (int k = 0; k < w; ++k) off += wsum[k];
        if (f) idx[(long long)b * Ta + off + before] = t;
        __syncthreads();
        if (tid == 0) base_s += wsum[0] + wsum[1] + wsum[2] + wsum[3];
        __syncthreads();
    }
    if (tid == 0) {
        cnt[b] = base_s;
        atomicMax(tmax + b / gsize, base_s);
    }
}

struct LerpCoef { int i0, i1; float lam; };
__device__ __forceinline__ LerpCoef lerp_coef(int i, int Tm, int Tv) {
    LerpCoef c;
    if (Tm == Tv) { c.i0 = i; c.i1 = i; c.lam = 0.f; return c; }          // no interpolation (fusion_module.py:50)
    const float scale = Tv > 1 ? (float)(Tm - 1) / (float)(Tv - 1) : 0.f;  // align_corners=True
    const float src = scale * (float)i;
    c.i0 = (int)src;
    if (c.i0 > Tm - 1) c.i0 = Tm - 1;
    c.lam = src - (float)c.i0;
    c.i1 = c.i0 + (c.i0 < Tm - 1 ? 1 : 0);
    return c;
}

__global__ __launch_bounds__(256) void gather_lerp_fwd_kernel(const float* __restrict__ feat, const long long* __restrict__ mask,
                                                              const int* __restrict__ idx, const int* __restrict__ cnt,
                                                              const int* __restrict__ tmax, float* __restrict__ out,
                                                              long long* __restrict__ mout, unsigned long long* __restrict__ lens,
                                                              int Ta, int Tv, int D, int gsize) {
    const int i = blockIdx.x, b = blockIdx.y;
    const int Tm = tmax[b / gsize], n = cnt[b];
    const LerpCoef c = lerp_coef(i, Tm, Tv);
    const bool ok0 = c.i0 < n, ok1 = c.i1 < n;
    const float* r0 = ok0 ? feat + ((long long)b * Ta + idx[(long long)b * Ta + c.i0]) * D : nullptr;
    const float* r1 = ok1 ? feat + ((long long)b * Ta + idx[(long long)b * Ta + c.i1]) * D : nullptr;
    float* o = out + ((long long)b * Tv + i) * D;
    const float w0 = 1.f - c.lam, w1 = c.lam;
    for (int k = threadIdx.x; k < D; k += 256) o[k] = w0 * (ok0 ? r0[k] : 0.f) + w1 * (ok1 ? r1[k] : 0.f);
    if (threadIdx.x == 0) {
        int j;
        if (Tm == Tv) j = i;
        else {
            j = (int)floorf((float)i * ((float)Tm / (float)Tv));
            if (j > Tm - 1) j = Tm - 1;
        }
        const long long mv = j < n ? mask[(long long)b * Ta + idx[(long long)b * Ta + j]] : 0;
        mout[(long long)b * Tv + i] = mv;
        if (mv != 0) atomicAdd(lens + b, 1ull);
    }
}

// dfeat[b][idx[j]][:] = sum_i w(i,j) dout[b][i][:]   (dfeat pre-zeroed; every source frame is written by one block)
__global__ __launch_bounds__(256) void gather_lerp_bwd_kernel(const float* __restrict__ dout, const int* __restrict__ idx,
                                                              const int* __restrict__ cnt, const int* __restrict__ tmax,
                                                              float* __restrict__ dfeat, int Ta, int Tv, int D, int gsize) {
    const int j = blockIdx.x, b = blockIdx.y;
    const int Tm = tmax[b / gsize], n = cnt[b];
    if (j >= n) return;
    int lo, hi;
    if (Tm == Tv) { lo = j; hi = j; }
    else {
        const float scale = Tv > 1 ? (float)(Tm - 1) / (float)(Tv - 1) : 0.f;
        if (scale <= 0.f) { lo = 0; hi = Tv - 1; }
        else {
            lo = (int)floorf((float)(j - 1) / scale) - 1;
            hi = (int)ceilf((float)(j + 1) / scale) + 1;
            if (lo < 0) lo = 0;
            if (hi > Tv - 1) hi = Tv - 1;
        }
    }
    float* o = dfeat + ((long long)b * Ta + idx[(long long)b * Ta + j]) * D;
    for (int k = threadIdx.x; k < D; k += 256) {
        float acc = 0.f;
        for (int i = lo; i <= hi; ++i) {
            const LerpCoef c = lerp_coef(i, Tm, Tv);
            float wgt = 0.f;
            if (c.i0 == j) wgt += 1.f - c.lam;
            if (c.i1 == j) wgt += c.lam;
            if (wgt != 0.f) acc += wgt * dout[((long long)b * Tv + i) * D + k];
        }
        o[k] = acc;
    }
}

// out[(t*B + b)][:] = in[(b*T + t)][:]  ([B,T,D] <-> [T,B,D] with dims swapped by the caller)
__global__ void permute_kernel(const void* __restrict__ in, int idt, void* __restrict__ out, int odt, int B, int T, int D) {
    const long long n = (long long)B * T * D;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long long)gridDim.x * blockDim.x) {
        const int k = (int)(e % D);
        const long long row = e / D;
        const int b = (int)(row % B), t = (int)(row / B);
        st_any(out, e, odt, ld_any(in, ((long long)b * T + t) * D + k, idt));
    }
}

__global__ void gather_rows_kernel(const void* __restrict__ src, int sdt, const long long* __restrict__ idx, void* __restrict__ out,
                                   int odt, long long n, int D) {
    const long long tot = n * D;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < tot; e += (long long)gridDim.x * blockDim.x) {
        const long long rrow = e / D;
        const int k = (int)(e - rrow * D);
        st_any(out, e, odt, ld_any(src, idx[rrow] * D + k, sdt));
    }
}
// Stable class order of the contrastive loss (contrastive.py:24-26 of the reference: anchors = mask 1, positives = 2, negatives = 0, the
// rest dropped): order[] = the indices of mask == 1 in ascending order, then those of 2, of 0, and of everything else - what a stable
// argsort of the class rank returns, in ONE launch of one workgroup (the sort was ~10 launches of radix / merge passes for 12 736 keys).
// Thread t owns the contiguous chunk [t * per, (t + 1) * per): per-class counts -> exclusive scan over the threads in LDS -> ordered writes.
__global__ __launch_bounds__(1024) void class_order_kernel(const long long* __restrict__ mask, long long n, long long* __restrict__ order) {
    __shared__ int cnt[4][1024];
    __shared__ int tot[4];
    const int t = threadIdx.x;
    const long long per = (n + 1023) / 1024, lo = (long long)t * per, hi = lo + per < n ? lo + per : n;
    int c[4] = {0, 0, 0, 0};
    for (long long i = lo; i < hi; ++i) {
        const long long m = mask[i];
        const int k = m == 1 ? 0 : m == 2 ? 1 : m <= 0 ? 2 : 3;       // rank: 1, 2, 0 (values below 0 count as 0, above 3 as 3: the clamp of the counts)
        ++c[k];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) cnt[k][t] = c[k];
    __syncthreads();
    if (t < 4) {                                              // 4 serial scans of 1024 counters (one thread each): ~1 us, off any critical path
        int run = 0;
        for (int j = 0; j < 1024; ++j) { const int v = cnt[t][j]; cnt[t][j] = run; run += v; }
        tot[t] = run;
    }
    __syncthreads();
    long long base[4];
    base[0] = cnt[0][t];
    base[1] = (long long)tot[0] + cnt[1][t];
    base[2] = (long long)tot[0] + tot[1] + cnt[2][t];
    base[3] = (long long)tot[0] + tot[1] + tot[2] + cnt[3][t];
    for (long long i = lo; i < hi; ++i) {
        const long long m = mask[i];
        const int k = m == 1 ? 0 : m == 2 ? 1 : m <= 0 ? 2 : 3;
        order[k == 0 ? base[0]++ : k == 1 ? base[1]++ : k == 2 ? base[2]++ : base[3]++] = i;
    }
}

__global__ void scatter_rows_kernel(const float* __restrict__ src, const long long* __restrict__ idx, float* __restrict__ out, long long n,
                                    int D, float alpha, int accumulate) {
    const long long tot = n * D;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < tot; e += (long long)gridDim.x * blockDim.x) {
        const long long rrow = e / D;
        const int k = (int)(e - rrow * D);
        float* o = out + idx[rrow] * D + k;
        *o = alpha * src[e] + (accumulate ? *o : 0.f);
    }
}

inline int ew_grid(long long n) {
    long long b = (n + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

}  // namespace

extern "C" int av_mask_downsample(const long long* mask, long long* out, int B, int Tin, int Tout, void* stream) {
    AV_CHECK(mask && out && B > 0 && Tin > 0 && Tout > 0, "av_mask_downsample: bad args");
    hipLaunchKernelGGL(mask_down_kernel, dim3((B * Tout + 255) / 256), dim3(256), 0, (hipStream_t)stream, mask, out, B, Tin, Tout);
    AV_LAUNCH_CHECK();
    return AV_OK;
}

// ws_i32: [B*Ta idx][B cnt][groups tmax] ints.  lens: int64 [B] (zeroed here).  The "batch maximum" the reference pads
// to is taken per group of B/groups consecutive items (groups = 1: the reference's per-call semantics; groups = 2 lets
// the trainer push both speakers through one call without changing the result).
extern "C" int av_fusion_gather_lerp_fwd(const float* feat, const long long* mask, int* ws_i32, float* out, long long* mask_out,
                                         long long* lens, int B, int Ta, int Tv, int D, int groups, void* stream) {
    AV_CHECK(feat && mask && ws_i32 && out && mask_out && lens, "av_fusion_gather_lerp_fwd: null pointer");
    AV_CHECK(B > 0 && Ta > 0 && Tv > 0 && D > 0 && groups >= 1 && B % groups == 0, "av_fusion_gather_lerp_fwd: bad shape (B=%d groups=%d)", B, groups);
    hipStream_t st = (hipStream_t)stream;
    const int gsize = B / groups;
    int* idx = ws_i32; int* cnt = ws_i32 + (long long)B * Ta; int* tmax = cnt + B;
    if (hipMemsetAsync(tmax, 0, sizeof(int) * groups, st) != hipSuccess || hipMemsetAsync(lens, 0, sizeof(long long) * B, st) != hipSuccess) {
        av_set_error("av_fusion_gather_lerp_fwd: memset failed"); return AV_ERR_LAUNCH;
    }
    hipLaunchKernelGGL(compact_kernel, dim3(B), dim3(256), 0, st, mask, idx, cnt, tmax, Ta, gsize);
    AV_LAUNCH_CHECK();
    hipLaunchKernelGGL(gather_lerp_fwd_kernel, dim3(Tv, B), dim3(256), 0, st, feat, mask, idx, cnt, tmax, out, mask_out,
                       (unsigned long long*)lens, Ta, Tv, D, gsize);
    AV_LAUNCH_CHECK();
    return AV_OK;
}

extern "C" int av_fusion_gather_lerp_bwd(const float* dout, const int* ws_i32, float* dfeat, int B, int Ta, int Tv, int D, int groups,
                                         void* stream) {
    AV_CHECK(dout && ws_i32 && dfeat && groups >= 1 && B % groups == 0, "av_fusion_gather_lerp_bwd: bad args");
    const int gsize = B / groups;
    hipStream_t st = (hipStream_t)stream;
    const int* idx = ws_i32; const int* cnt = ws_i32 + (long long)B * Ta; const int* tmax = cnt + B;
    if (hipMemsetAsync(dfeat, 0, sizeof(float) * (size_t)B * Ta * D, st) != hipSuccess) { av_set_error("av_fusion_gather_lerp_bwd: memset failed"); return AV_ERR_LAUNCH; }
    hipLaunchKernelGGL(gather_lerp_bwd_kernel, dim3(Ta, B), dim3(256), 0, st, dout, idx, cnt, tmax, dfeat, Ta, Tv, D, gsize);
    AV_LAUNCH_CHECK();
    return AV_OK;
}

extern "C" int av_permute_bt(const void* in, int idt, void* out, int odt, int B, int T, int D, void* stream) {
    AV_CHECK(in && out && B > 0 && T > 0 && D > 0, "av_permute_bt: bad args");
    hipLaunchKernelGGL(permute_kernel, dim3(ew_grid((long long)B * T * D)), dim3(256), 0, (hipStream_t)stream, in, idt, out, odt, B, T, D);
    AV_LAUNCH_CHECK();
    return AV_OK;
}

extern "C" int av_gather_rows(const void* src, int sdt, const long long* idx, void* out, int odt, long long n, int D, void* stream) {
    AV_CHECK(src && idx && out && D > 0, "av_gather_rows: bad args");
    if (n == 0) return AV_OK;
    hipLaunchKernelGGL(gather_rows_kernel, dim3(ew_grid(n * D)), dim3(256), 0, (hipStream_t)stream, src, sdt, idx, out, odt, n, D);
    AV_LAUNCH_CHECK();
    return AV_OK;
}

extern "C" int av_scatter_rows(const float* src, const long long* idx, float* out, long long n, int D, float alpha, int accumulate,
                               void* stream) {
    AV_CHECK(src && idx && out && D > 0, "av_scatter_rows: bad args");
    if (n == 0) return AV_OK;
    hipLaunchKernelGGL(scatter_rows_kernel, dim3(ew_grid(n * D)), dim3(256), 0, (hipStream_t)stream, src, idx, out, n, D, alpha, accumulate);
    AV_LAUNCH_CHECK();
    return AV_OK;
}

extern "C" int av_class_order(const long long* mask, long long n, long long* order, void* stream) {
    AV_CHECK(mask && order && n >= 0 && n < (1ll << 31), "av_class_order: bad args");
    if (n == 0) return AV_OK;
    hipLaunchKernelGGL(class_order_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, mask, n, order);
    AV_LAUNCH_CHECK();
    return AV_OK;
}
