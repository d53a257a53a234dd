// Small kernels of the loss / optimizer side:
//   * row L2-normalise fwd/bwd (F.normalize, contrastive.py:22),
//   * row log-sum-exp + row sum of the similarity matrix and the softmax-minus-uniform gradient of
//     -log_softmax(sim).mean() (contrastive.py:33-34,41-42),
//   * scalar reductions,
//   * fused Adam step (torch.optim.Adam defaults, model/trainer.py:34-39) with optional gradient scaling.
#include "av_common.h"

namespace {

constexpr int MAXIT = 32;

__global__ __launch_bounds__(256) void l2norm_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, float* __restrict__ nrm,
                                                         long long rows, int cols, float eps) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const long long base = row * cols;
    float q = 0.f;
    for (int c = lane; c < cols; c += 64) { const float v = x[base + c]; q += v * v; }
    const float n = sqrtf(wave_sum(q));
    const float inv = 1.f / fmaxf(n, eps);
    for (int c = lane; c < cols; c += 64) y[base + c] = x[base + c] * inv;
    if (lane == 0) nrm[row] = n;
}
// dx = (dy - y (y . dy)) / max(n, eps)
__global__ __launch_bounds__(256) void l2norm_bwd_kernel(const float* __restrict__ y, const float* __restrict__ dy, const float* __restrict__ nrm,
                                                         float* __restrict__ dx, long long rows, int cols, float eps) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const long long base = row * cols;
    float d = 0.f;
    for (int c = lane; c < cols; c += 64) d += y[base + c] * dy[base + c];
    d = wave_sum(d);
    const float inv = 1.f / fmaxf(nrm[row], eps);
    for (int c = lane; c < cols; c += 64) dx[base + c] = (dy[base + c] - y[base + c] * d) * inv;
}

__global__ __launch_bounds__(256) void lse_rows_kernel(const float* __restrict__ s, float* __restrict__ lse, float* __restrict__ rsum,
                                                       long long rows, int cols, int ld, int accumulate) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* p = s + row * ld;
    float mx = -INFINITY, sm = 0.f;
    for (int c = lane; c < cols; c += 64) { const float v = p[c]; mx = fmaxf(mx, v); sm += v; }
    mx = wave_max(mx);
    float e = 0.f;
    for (int c = lane; c < cols; c += 64) e += expf(p[c] - mx);
    e = wave_sum(e);
    sm = wave_sum(sm);
    if (lane == 0) {
        float l = mx + logf(e);
        if (accumulate) {                                        // merge with the row's running values (earlier column chunks)
            const float lo = lse[row], hi = fmaxf(lo, l);
            l = hi + logf(expf(lo - hi) + expf(l - hi));
            sm += rsum[row];
        }
        lse[row] = l; rsum[row] = sm;
    }
}

// dsim = coef * (exp(sim - lse_row) - 1/cols)
__global__ __launch_bounds__(256) void contrastive_dsim_kernel(const float* __restrict__ s, const float* __restrict__ lse, void* __restrict__ out,
                                                               int odt, long long rows, int cols, int ld, float coef, float u) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float l = lse[row];
    for (int c = lane; c < ld; c += 64) st_any(out, row * ld + c, odt, c < cols ? coef * (expf(s[row * ld + c] - l) - u) : 0.f);
}

__global__ __launch_bounds__(256) void reduce_sum_kernel(const float* __restrict__ x, long long n, float* __restrict__ out, float scale, int accumulate) {
    __shared__ float red[4];
    float s = 0.f;
    for (long long i = threadIdx.x; i < n; i += 256) s += x[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float v = scale * (red[0] + red[1] + red[2] + red[3]);
        out[0] = accumulate ? out[0] + v : v;
    }
}

// total = sum_i nll[i] * w[i] + half_lambda * (c1 + c2);  l1 / l2 = 2 x the two halves of the weighted sum (the per-speaker CTC means of
// model/trainer.py:111-118 with w[i] = 1 / (2 B clamp(target_len_i, 1))); one launch instead of a dozen scalar torch ops
__global__ __launch_bounds__(256) void loss_combine_kernel(const float* __restrict__ nll, const float* __restrict__ w, const float* __restrict__ c1,
                                                           const float* __restrict__ c2, float half_lambda, int n, float* __restrict__ out) {
    __shared__ float red[2][4];
    const int hn = n / 2;
    float s0 = 0.f, s1 = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) {
        const float v = nll[i] * w[i];
        if (i < hn) s0 += v; else s1 += v;
    }
    s0 = wave_sum(s0); s1 = wave_sum(s1);
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s0; red[1][threadIdx.x >> 6] = s1; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const float a = red[0][0] + red[0][1] + red[0][2] + red[0][3], b = red[1][0] + red[1][1] + red[1][2] + red[1][3];
        out[0] = a + b + half_lambda * ((c1 ? c1[0] : 0.f) + (c2 ? c2[0] : 0.f));
        out[1] = 2.f * a;
        out[2] = 2.f * b;
    }
}

__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, long long n,
                            float lr, float b1, float b2, float eps, float bc1, float bc2_sqrt, float gscale) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float gi = g[i] * gscale;
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi; v[i] = vi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        p[i] -= (lr / bc1) * (mi / denom);
    }
}

// multi-tensor Adam: one launch for every trainable tensor.  ptrs[t] = {param, grad, exp_avg, exp_avg_sq, shadow}; chunk c
// covers elements [chunk_start[c], +chunk_elems) of tensor chunk_tensor[c].  shadow (optional, 0 = none) is a bf16 copy of
// the parameter in the layout the compute kernels read (the perf path's cached weight): writing it here saves the separate
// cast pass over every trainable weight after each step.
__device__ __forceinline__ float adam_one(float& p, float g, float& m, float& v, float b1, float b2, float eps, float step_size, float bc2_sqrt,
                                          float gscale) {
    const float gi = g * gscale;
    m = b1 * m + (1.f - b1) * gi;
    v = b2 * v + (1.f - b2) * gi * gi;
    p -= step_size * (m / (sqrtf(v) / bc2_sqrt + eps));
    return p;
}
// Loss-scaling state on the device (torch.amp.GradScaler semantics without the host synchronisation of its found_inf.item()):
// gs[0] scale, gs[1] 1/scale, gs[2] found_inf (0/1), gs[3] growth tracker, gs[4] optimizer steps actually taken.
enum { GS_SCALE = 0, GS_INV = 1, GS_INF = 2, GS_TRACK = 3, GS_STEP = 4 };

// any non-finite gradient element -> gs[GS_INF] = 1   (torch/amp/grad_scaler.py:_unscale_grads_: _amp_foreach_non_finite_check_and_unscale_;
// the unscale itself is folded into the Adam kernel)
__global__ __launch_bounds__(256) void grad_check_multi_kernel(const unsigned long long* __restrict__ ptrs, const long long* __restrict__ sizes,
                                                               const int* __restrict__ chunk_tensor, const long long* __restrict__ chunk_start,
                                                               int chunk_elems, float* __restrict__ gs) {
    const int c = blockIdx.x;
    const int t = chunk_tensor[c];
    const float* g = (const float*)ptrs[5 * t + 1];
    const long long n = sizes[t];
    const long long s0 = chunk_start[c];
    long long s1 = s0 + chunk_elems;
    if (s1 > n) s1 = n;
    bool bad = false;
    long long i = s0;
    if (((uintptr_t)g % 16 == 0) && (s0 % 4 == 0)) {
        const long long nv = (s1 - s0) / 4;
        for (long long q = threadIdx.x; q < nv; q += 256) {
            const f32x4 gg = *(const f32x4*)(g + s0 + 4 * q);
            // x - x is 0 for finite x and NaN for +-inf / NaN
            const float z = (gg[0] - gg[0]) + (gg[1] - gg[1]) + (gg[2] - gg[2]) + (gg[3] - gg[3]);
            bad |= !(z == 0.f);
        }
        i = s0 + 4 * nv;
    }
    for (i += threadIdx.x; i < s1; i += 256) { const float x = g[i]; bad |= !((x - x) == 0.f); }
    if (__any(bad) && (threadIdx.x & 63) == 0) gs[GS_INF] = 1.0f;          // every writer stores the same value
}

// torch/amp/grad_scaler.py:update (_amp_update_scale_) is applied by adam_finish_kernel below: found_inf -> scale *= backoff, tracker = 0;
// otherwise tracker += 1 and scale *= growth when it reaches the interval; it also counts the optimizer steps that were not skipped.

// Per-tensor step counts live on the device (int32 table; tensor t of a launch owns slot step_slot[t]): a parameter that received no
// gradient in some step (LayerDrop skipped its layer in both passes: torch.optim.Adam leaves its state['step'] behind) keeps its own bias
// corrections, every step goes through THIS kernel, and under loss scaling a skipped (overflowing) step advances no counter - all without
// the host knowing.  hyper[t] = {lr, beta1, beta2, eps}.  Bias corrections in double (1 - beta2^step loses half its digits in fp32).
__global__ __launch_bounds__(256) void adam_multi_kernel(const unsigned long long* __restrict__ ptrs, const long long* __restrict__ sizes,
                                                         const f32x4* __restrict__ hyper, const int* __restrict__ chunk_tensor,
                                                         const long long* __restrict__ chunk_start, int chunk_elems,
                                                         const int* __restrict__ steps, const int* __restrict__ step_slot, float gscale,
                                                         const float* __restrict__ gs) {
    if (gs) {                                                    // loss scaling: skip on overflow, unscale
        if (gs[GS_INF] != 0.f) return;
        gscale *= gs[GS_INV];
    }
    const int c = blockIdx.x;
    const int t = chunk_tensor[c];
    const f32x4 hp = hyper[t];
    const float b1 = hp[1], b2 = hp[2], eps = hp[3];
    const double step = (double)(steps[step_slot[t]] + 1);
    const float bc1 = (float)(1.0 - pow((double)b1, step));
    const float bc2_sqrt = (float)sqrt(1.0 - pow((double)b2, step));
    float* p = (float*)ptrs[5 * t + 0];
    const float* g = (const float*)ptrs[5 * t + 1];
    float* m = (float*)ptrs[5 * t + 2];
    float* v = (float*)ptrs[5 * t + 3];
    bf16_t* sh = (bf16_t*)ptrs[5 * t + 4];
    const long long n = sizes[t];
    const long long s0 = chunk_start[c];
    long long s1 = s0 + chunk_elems;
    if (s1 > n) s1 = n;
    const float step_size = hp[0] / bc1;
    const bool vec = (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) % 16 == 0) && ((uintptr_t)sh % 8 == 0) && (s0 % 4 == 0);
    long long i = s0;
    if (vec) {
        const long long nv = (s1 - s0) / 4;
        for (long long q = threadIdx.x; q < nv; q += 256) {
            const long long e = s0 + 4 * q;
            f32x4 pp = *(const f32x4*)(p + e), mm = *(const f32x4*)(m + e), vv = *(const f32x4*)(v + e);
            const f32x4 gg = *(const f32x4*)(g + e);
            bf16x4 o;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float pk = pp[k], mk = mm[k], vk = vv[k];
                o[k] = (bf16_t)adam_one(pk, gg[k], mk, vk, b1, b2, eps, step_size, bc2_sqrt, gscale);
                pp[k] = pk; mm[k] = mk; vv[k] = vk;
            }
            *(f32x4*)(p + e) = pp; *(f32x4*)(m + e) = mm; *(f32x4*)(v + e) = vv;
            if (sh) *(bf16x4*)(sh + e) = o;
        }
        i = s0 + 4 * nv;
    }
    for (i += threadIdx.x; i < s1; i += 256) {
        float pp = p[i], mm = m[i], vv = v[i];
        const float r = adam_one(pp, g[i], mm, vv, b1, b2, eps, step_size, bc2_sqrt, gscale);
        p[i] = pp; m[i] = mm; v[i] = vv;
        if (sh) sh[i] = (bf16_t)r;
    }
}

// after the Adam launch (stream order): the tensors of that launch have taken one more step - unless the step was skipped for overflow.
// With a scaler this launch also applies the GradScaler update law (one thread).
__global__ void adam_finish_kernel(int* __restrict__ steps, const int* __restrict__ step_slot, int n_tensors, float* __restrict__ gs,
                                   float growth, float backoff, int interval) {
    const bool skipped = gs && gs[GS_INF] != 0.f;
    __syncthreads();                                              // everyone has read found_inf before thread 0 clears it
    if (!skipped)
        for (int t = threadIdx.x; t < n_tensors; t += blockDim.x) steps[step_slot[t]] += 1;
    if (gs && threadIdx.x == 0) {
        if (skipped) {
            gs[GS_SCALE] *= backoff;
            gs[GS_TRACK] = 0.f;
        } else {
            gs[GS_STEP] += 1.f;
            const float tr = gs[GS_TRACK] + 1.f;
            if (tr >= (float)interval) { gs[GS_SCALE] *= growth; gs[GS_TRACK] = 0.f; }
            else gs[GS_TRACK] = tr;
        }
        gs[GS_INV] = 1.0f / gs[GS_SCALE];
        gs[GS_INF] = 0.f;
    }
}

inline int ew_grid(long long n) {
    long long b = (n + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

}  // namespace

extern "C" int av_l2norm_fwd(const float* x, float* y, float* nrm, long long rows, int cols, float eps, void* stream) {
    AV_CHECK(x && y && nrm && cols > 0, "av_l2norm_fwd: bad args");
    if (rows == 0) return AV_OK;
    hipLaunchKernelGGL(l2norm_fwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, y, nrm, rows, cols, eps);
    AV_LAUNCH_CHECK();
    return AV_OK;
}
extern "C" int av_l2norm_bwd(const float* y, const float* dy, const float* nrm, float* dx, long long rows, int cols, float eps, void* stream) {
    AV_CHECK(y && dy && nrm && dx && cols > 0, "av_l2norm_bwd: bad args");
    if (rows == 0) return AV_OK;
    hipLaunchKernelGGL(l2norm_bwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, y, dy, nrm, dx, rows, cols, eps);
    AV_LAUNCH_CHECK();
    return AV_OK;
}
extern "C" int av_lse_rows(const float* s, float* lse, float* rowsum, long long rows, int cols, int ld, void* stream) {
    AV_CHECK(s && lse && rowsum && cols > 0 && ld >= cols, "av_lse_rows: bad args");
    if (rows == 0) return AV_OK;
    hipLaunchKernelGGL(lse_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, s, lse, rowsum, rows, cols, ld, 0);
    AV_LAUNCH_CHECK();
    return AV_OK;
}
extern "C" int av_lse_rows_chunk(const float* s, float* lse, float* rowsum, long long rows, int cols, int ld, int accumulate, void* stream) {
    AV_CHECK(s && lse && rowsum && cols > 0 && ld >= cols, "av_lse_rows_chunk: bad args");
    if (rows == 0) return AV_OK;
    hipLaunchKernelGGL(lse_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, s, lse, rowsum, rows, cols, ld, accumulate);
    AV_LAUNCH_CHECK();
    return AV_OK;
}
extern "C" int av_contrastive_dsim(const float* s, const float* lse, void* out, int odt, long long rows, int cols, int ld, float coef, void* stream) {
    AV_CHECK(s && lse && out && cols > 0 && ld >= cols, "av_contrastive_dsim: bad args");
    if (rows == 0) return AV_OK;
    hipLaunchKernelGGL(contrastive_dsim_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, s, lse, out, odt, rows, cols, ld, coef,
                       1.f / (float)cols);
    AV_LAUNCH_CHECK();
    return AV_OK;
}
extern "C" int av_contrastive_dsim_chunk(const float* s, const float* lse, void* out, int odt, long long rows, int cols, int ld, float coef,
                                         int total_cols, void* stream) {
    AV_CHECK(s && lse && out && cols > 0 && ld >= cols && total_cols >= cols, "av_contrastive_dsim_chunk: bad args");
    if (rows == 0) return AV_OK;
    hipLaunchKernelGGL(contrastive_dsim_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, s, lse, out, odt, rows, cols, ld, coef,
                       1.f / (float)total_cols);
    AV_LAUNCH_CHECK();
    return AV_OK;
}
extern "C" int av_loss_combine(const float* nll, const float* w, const float* c1, const float* c2, float half_lambda, int n, float* out3,
                               void* stream) {
    AV_CHECK(nll && w && out3 && n >= 2 && n % 2 == 0, "av_loss_combine: bad args (n=%d)", n);
    hipLaunchKernelGGL(loss_combine_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, nll, w, c1, c2, half_lambda, n, out3);
    AV_LAUNCH_CHECK();
    return AV_OK;
}
extern "C" int av_reduce_sum(const float* x, long long n, float* out, float scale, int accumulate, void* stream) {
    AV_CHECK(x && out, "av_reduce_sum: bad args");
    hipLaunchKernelGGL(reduce_sum_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, x, n, out, scale, accumulate);
    AV_LAUNCH_CHECK();
    return AV_OK;
}
extern "C" int av_adam_step(float* p, const float* g, float* m, float* v, long long n, float lr, float beta1, float beta2, float eps,
                            int step, float grad_scale, void* stream) {
    AV_CHECK(p && g && m && v && step >= 1, "av_adam_step: bad args");
    if (n == 0) return AV_OK;
    const float bc1 = (float)(1.0 - pow((double)beta1, (double)step));
    const float bc2s = (float)sqrt(1.0 - pow((double)beta2, (double)step));
    hipLaunchKernelGGL(adam_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, lr, beta1, beta2, eps, bc1, bc2s, grad_scale);
    AV_LAUNCH_CHECK();
    return AV_OK;
}

extern "C" int av_adam_multi(const void* ptrs, const long long* sizes, const float* hyper, const int* chunk_tensor, const long long* chunk_start,
                             int n_chunks, int chunk_elems, int* steps, const int* step_slot, int n_tensors, float grad_scale,
                             float* scaler_state, float growth, float backoff, int growth_interval, void* stream) {
    AV_CHECK(ptrs && sizes && hyper && chunk_tensor && chunk_start && steps && step_slot && n_chunks >= 0 && n_tensors >= 0 && chunk_elems > 0,
             "av_adam_multi: bad args");
    AV_CHECK(((uintptr_t)hyper % 16) == 0, "av_adam_multi: hyper must be 16-byte aligned ([n_tensors][4] floats)");
    AV_CHECK(!scaler_state || growth_interval >= 1, "av_adam_multi: bad growth interval");
    hipStream_t st = (hipStream_t)stream;
    if (n_chunks > 0) {
        if (scaler_state)
            hipLaunchKernelGGL(grad_check_multi_kernel, dim3(n_chunks), dim3(256), 0, st, (const unsigned long long*)ptrs, sizes, chunk_tensor,
                               chunk_start, chunk_elems, scaler_state);
        hipLaunchKernelGGL(adam_multi_kernel, dim3(n_chunks), dim3(256), 0, st, (const unsigned long long*)ptrs, sizes, (const f32x4*)hyper,
                           chunk_tensor, chunk_start, chunk_elems, (const int*)steps, step_slot, grad_scale, (const float*)scaler_state);
    }
    if (n_tensors > 0 || scaler_state)
        hipLaunchKernelGGL(adam_finish_kernel, dim3(1), dim3(256), 0, st, steps, step_slot, n_tensors, scaler_state, growth, backoff,
                           growth_interval);
    AV_LAUNCH_CHECK();
    return AV_OK;
}
