// Fused multi-head attention forward for gfx950 (flash-style, online softmax):
//   O = softmax(scale * Q K^T + key-padding mask) V,   LSE = log sum exp  (saved for backward)
// replaces hf:438-463 / sdpa (wav2vec2 self-attention, 16 heads x 64) and the need_weights path of
// nn.MultiheadAttention (torch:functional.py:6576-6606; fusion cross-attention, 4 heads x 128).
// One workgroup = 64 query rows of one (batch, head); 4 wavefronts x 16 rows.  Q, K tiles and the transposed V
// tile are staged in LDS; QK^T and PV run on MFMA 16x16 tiles (bf16 16x16x32 / exact-fp32 16x16x4); the row max
// / row sum of the softmax are reduced with wavefront shuffles inside each 16-lane group (the S accumulator has
// the key on the lane).  P goes through a per-wave LDS tile to become the A operand of PV.
// Also here: the row kernels of the (unfused) backward: softmax rows and its gradient.
#include "attn_common.h"

namespace {

template <typename T, int D>
__global__ __launch_bounds__(256) void attn_fwd_kernel(const AttnP p) {
    constexpr int DK = (D + 31) / 32 * 32;
    constexpr int DN = D / 16;
    constexpr int PAD = 16 / sizeof(T);
    constexpr int LDK = DK + PAD, LDV = 64 + PAD;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* Qs = (T*)smem;               // [64][LDK]
    T* Ks = Qs + 64 * LDK;          // [64][LDK]
    T* Vt = Ks + 64 * LDK;          // [D][LDV]
    T* Ps = Vt + D * LDV;           // [4][16][LDV]

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int r = lane & 15, g = lane >> 4;
    const int q0 = blockIdx.x * 64, h = blockIdx.y, b = blockIdx.z;
    const bool vec = p.vec_ok != 0;
    const T* Q = (const T*)p.q + (long long)b * p.q_bs + (long long)h * D;
    const T* K = (const T*)p.k + (long long)b * p.k_bs + (long long)h * D;
    const T* V = (const T*)p.v + (long long)b * p.v_bs + (long long)h * D;
    int klen = p.klen ? p.klen[b] : p.Tk;
    if (klen > p.Tk) klen = p.Tk;
    if (klen < 1) klen = 1;

    stage_rows<T, D, DK>(Qs, LDK, Q, p.q_rs, q0, p.Tq, vec, tid);

    f32x4 O[DN];
#pragma unroll
    for (int n = 0; n < DN; ++n) O[n] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m[4], l[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) { m[e] = -INFINITY; l[e] = 0.f; }

    T* Pw = Ps + w * 16 * LDV;
    const T* Qw = Qs + w * 16 * LDK;
    __syncthreads();                                   // Q tile staged
    AFrag<T, DK> qf;
    qf.load(Qw, LDK, lane);                            // my 16 query rows: loaded once, reused for every key tile
    for (int k0 = 0; k0 < klen; k0 += 64) {
        stage_rows<T, D, DK>(Ks, LDK, K, p.k_rs, k0, klen, vec, tid);
        stage_vt<T, D>(Vt, LDV, V, p.v_rs, k0, klen, vec, tid);
        __syncthreads();
        f32x4 S[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            S[j] = f32x4{0.f, 0.f, 0.f, 0.f};
            qf.mma(S[j], Ks + j * 16 * LDK, LDK, lane);
        }
        float mx[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) mx[e] = -INFINITY;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool valid = (k0 + j * 16 + r) < klen;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                S[j][e] = valid ? S[j][e] * p.scale : -INFINITY;
                mx[e] = fmaxf(mx[e], S[j][e]);
            }
        }
        float alpha[4], rs[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float v = mx[e];
            v = fmaxf(v, __shfl_xor(v, 1, 64)); v = fmaxf(v, __shfl_xor(v, 2, 64));
            v = fmaxf(v, __shfl_xor(v, 4, 64)); v = fmaxf(v, __shfl_xor(v, 8, 64));
            const float mn = fmaxf(m[e], v);
            alpha[e] = expf(m[e] - mn);       // first tile: exp(-inf) = 0
            m[e] = mn;
            rs[e] = 0.f;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float pv = expf(S[j][e] - m[e]);
                rs[e] += pv;
                float pd = pv;
                if (p.drop_p > 0.f) {
                    const unsigned long long idx = (((unsigned long long)b * p.H + h) * p.Tq + (q0 + w * 16 + 4 * g + e)) * ((p.Tk + 3) & ~3) + (k0 + j * 16 + r);   // row pitch padded to 4: see attn_common.h
                    pd *= drop_mult_call(p.drop_seed, p.drop_stream, idx, p.drop_p, drop_inv_keep(p.drop_p));
                }
                Pw[(4 * g + e) * LDV + j * 16 + r] = from_f32<T>(pd);
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float v = rs[e];
            v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
            l[e] = l[e] * alpha[e] + v;
        }
#pragma unroll
        for (int n = 0; n < DN; ++n)
#pragma unroll
            for (int e = 0; e < 4; ++e) O[n][e] *= alpha[e];
        __syncthreads();   // P visible to the whole wave (and keeps the 4 waves in step)
        {
            AFrag<T, 64> pf;
            pf.load(Pw, LDV, lane);
#pragma unroll
            for (int n = 0; n < DN; ++n) pf.mma(O[n], Vt + n * 16 * LDV, LDV, lane);
        }
        __syncthreads();   // K / Vt / P tiles are free again
    }

#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int row = q0 + w * 16 + 4 * g + e;
        if (row < p.Tq) {
            const float inv = 1.0f / l[e];
            T* o = (T*)p.o + (long long)b * p.o_bs + (long long)row * p.o_rs + (long long)h * D;
#pragma unroll
            for (int n = 0; n < DN; ++n) o[n * 16 + r] = from_f32<T>(O[n][e] * inv);
            if (r == 0 && p.lse) p.lse[((long long)b * p.H + h) * p.Tq + row] = m[e] + logf(l[e]);
        }
    }
}

template <typename T, int D>
int launch_fwd(const AttnP& p, hipStream_t st) {
    constexpr int DK = (D + 31) / 32 * 32;
    constexpr int PAD = 16 / sizeof(T);
    const size_t lds = sizeof(T) * (size_t)(2 * 64 * (DK + PAD) + D * (64 + PAD) + 4 * 16 * (64 + PAD));
    auto kern = attn_fwd_kernel<T, D>;
    static bool done = false;
    if (!done && lds > 48 * 1024) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
            av_set_error("av_attention_fwd: cannot raise dynamic LDS to %zu", lds);
            return AV_ERR_LAUNCH;
        }
        done = true;
    }
    dim3 grid((unsigned)((p.Tq + 63) / 64), (unsigned)p.H, (unsigned)p.B);
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, p);
    AV_LAUNCH_CHECK();
    return AV_OK;
}

template <typename T>
int dispatch_d(const AttnP& p, int D, hipStream_t st) {
    switch (D) {
        case 16: return launch_fwd<T, 16>(p, st);
        case 32: return launch_fwd<T, 32>(p, st);
        case 64: return launch_fwd<T, 64>(p, st);
        case 128: return launch_fwd<T, 128>(p, st);
        default: av_set_error("av_attention_fwd: head_dim %d not in {16,32,64,128}", D); return AV_ERR_ARG;
    }
}

// ---- rows of the unfused backward -------------------------------------------------------------------------
constexpr int MAXIT = 32;
// P[row][:] = softmax(scale*S[row][:klen]) (0 beyond klen); rows are [b][h][tq]
template <int NIT>
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ s, void* __restrict__ pout, int pdt, long long rows,
                                                           int cols, float scale, const int* __restrict__ klen, int rows_per_batch, int ld) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    int kl = klen ? klen[row / rows_per_batch] : cols;
    if (kl > cols) kl = cols;
    if (kl < 1) kl = 1;
    const long long base = row * ld;
    float v[NIT];
    float mx = -INFINITY;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int c = lane + 64 * it;
        v[it] = c < kl ? s[base + c] * scale : -INFINITY;
        mx = fmaxf(mx, v[it]);
    }
    mx = wave_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        v[it] = expf(v[it] - mx);
        sum += v[it];
    }
    const float inv = 1.0f / wave_sum(sum);
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int c = lane + 64 * it;
        if (c < cols) st_any(pout, base + c, pdt, v[it] * inv);
    }
}

// dS = scale * P o (dP - sum(dP o P))
template <int NIT>
__global__ __launch_bounds__(256) void softmax_bwd_rows_kernel(const void* __restrict__ pm, int pdt, const float* __restrict__ dp,
                                                               void* __restrict__ ds, int dsdt, long long rows, int cols, float scale, int ld) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const long long base = row * ld;
    float pv[NIT], dv[NIT];
    float s = 0.f;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int c = lane + 64 * it;
        pv[it] = c < cols ? ld_any(pm, base + c, pdt) : 0.f;
        dv[it] = c < cols ? dp[base + c] : 0.f;
        s += pv[it] * dv[it];
    }
    s = wave_sum(s);
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int c = lane + 64 * it;
        if (c < cols) st_any(ds, base + c, dsdt, scale * pv[it] * (dv[it] - s));
    }
}

}  // namespace

extern "C" int av_attention_fwd(const void* q, const void* k, const void* v, void* o, float* lse, int dtype, int B, int H, int Tq,
                                int Tk, int D, long long q_bs, long long q_rs, long long k_bs, long long k_rs, long long v_bs,
                                long long v_rs, long long o_bs, long long o_rs, const int* klen, float scale, float drop_p,
                                unsigned long long drop_seed, unsigned int drop_stream, void* stream) {
    return av_attention_fwd_mask(q, k, v, o, lse, dtype, B, H, Tq, Tk, D, q_bs, q_rs, k_bs, k_rs, v_bs, v_rs, o_bs, o_rs, klen, scale, drop_p, drop_seed,
                                 drop_stream, nullptr, stream);
}

extern "C" int av_attention_fwd_mask(const void* q, const void* k, const void* v, void* o, float* lse, int dtype, int B, int H, int Tq,
                                     int Tk, int D, long long q_bs, long long q_rs, long long k_bs, long long k_rs, long long v_bs,
                                     long long v_rs, long long o_bs, long long o_rs, const int* klen, float scale, float drop_p,
                                     unsigned long long drop_seed, unsigned int drop_stream, const void* drop_mask, void* stream) {
    AV_CHECK(q && k && v && o, "av_attention_fwd: null pointer");
    AV_CHECK(B > 0 && H > 0 && Tq > 0 && Tk > 0, "av_attention_fwd: bad shape B=%d H=%d Tq=%d Tk=%d", B, H, Tq, Tk);
    AV_CHECK(dtype == AV_F32 || dtype == AV_BF16, "av_attention_fwd: bad dtype %d", dtype);
    AttnP p;
    p.q = q; p.k = k; p.v = v; p.o = o; p.lse = lse; p.klen = klen;
    p.B = B; p.H = H; p.Tq = Tq; p.Tk = Tk;
    p.q_bs = q_bs; p.q_rs = q_rs; p.k_bs = k_bs; p.k_rs = k_rs; p.v_bs = v_bs; p.v_rs = v_rs; p.o_bs = o_bs; p.o_rs = o_rs;
    p.scale = scale;
    AV_CHECK(drop_p >= 0.f && drop_p < 1.f, "av_attention_fwd: drop_p=%f out of [0,1)", drop_p);
    p.drop_p = drop_p; p.drop_seed = drop_seed; p.drop_stream = drop_stream;
    AV_CHECK(!drop_mask || (dtype == AV_BF16 && D == 64 && Tq <= 256 && Tk <= 256 && (uintptr_t)drop_mask % 32 == 0),
             "av_attention_fwd_mask: the keep-bit mask is a feature of the whole-sequence bf16 kernels (head_dim 64, T <= 256), 32-byte aligned");
    p.dmask = (const unsigned long long*)drop_mask;
    const long long es = dtype == AV_F32 ? 4 : 2;
    auto al = [&](const void* ptr, long long bs, long long rs) {
        return ((uintptr_t)ptr % 16 == 0) && ((bs * es) % 16 == 0) && ((rs * es) % 16 == 0) && ((D * es) % 16 == 0);
    };
    p.vec_ok = al(q, q_bs, q_rs) && al(k, k_bs, k_rs) && al(v, v_bs, v_rs);
    if (dtype == AV_BF16) {
        const int rc = av_attention_short_fwd_try(p, D, (hipStream_t)stream);      // whole-sequence kernel: D = 64, T <= 256
        if (rc != AV_SHORT_NOT_TAKEN) return rc;
    }
    AV_CHECK(!drop_mask, "av_attention_fwd_mask: the whole-sequence kernel did not take this call (alignment / AVAMD_ATTN_SHORT=0): the keep bits cannot be used");
    return dtype == AV_F32 ? dispatch_d<float>(p, D, (hipStream_t)stream) : dispatch_d<bf16_t>(p, D, (hipStream_t)stream);
}

extern "C" int av_softmax_rows(const float* s, void* p, int pdt, long long rows, int cols, float scale, const int* klen,
                               int rows_per_batch, int ld, void* stream) {
    AV_CHECK(s && p, "av_softmax_rows: null pointer");
    AV_CHECK(cols > 0 && cols <= 64 * MAXIT, "av_softmax_rows: cols=%d out of range (1..%d)", cols, 64 * MAXIT);
    AV_CHECK(rows_per_batch > 0, "av_softmax_rows: rows_per_batch=%d", rows_per_batch);
    if (rows == 0) return AV_OK;
#define SMR(N) hipLaunchKernelGGL((softmax_rows_kernel<N>), dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, s, p, pdt, rows, cols, scale, klen, rows_per_batch, ld)
    if (cols <= 64) SMR(1); else if (cols <= 128) SMR(2); else if (cols <= 256) SMR(4); else if (cols <= 512) SMR(8); else if (cols <= 1024) SMR(16); else SMR(32);
#undef SMR
    AV_LAUNCH_CHECK();
    return AV_OK;
}

extern "C" int av_softmax_bwd_rows(const void* p, int pdt, const float* dp, void* ds, int dsdt, long long rows, int cols, float scale,
                                   int ld, void* stream) {
    AV_CHECK(p && dp && ds, "av_softmax_bwd_rows: null pointer");
    AV_CHECK(cols > 0 && cols <= 64 * MAXIT, "av_softmax_bwd_rows: cols=%d out of range", cols);
    if (rows == 0) return AV_OK;
#define SMB(N) hipLaunchKernelGGL((softmax_bwd_rows_kernel<N>), dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, p, pdt, dp, ds, dsdt, rows, cols, scale, ld)
    if (cols <= 64) SMB(1); else if (cols <= 128) SMB(2); else if (cols <= 256) SMB(4); else if (cols <= 512) SMB(8); else if (cols <= 1024) SMB(16); else SMB(32);
#undef SMB
    AV_LAUNCH_CHECK();
    return AV_OK;
}
