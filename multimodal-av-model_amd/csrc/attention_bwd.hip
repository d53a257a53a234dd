// Fused (flash-style) attention backward for the bf16 perf path: P is recomputed from Q, K and the forward's LSE, the
// T x T score matrices never reach HBM.  Backward of hf:438-463 (sdpa) and of nn.MultiheadAttention's core.
//   delta[q]  = sum_d dO[q][d] O[q][d]
//   P         = exp(scale * Q K^T - LSE[q])      (0 for masked keys)
//   dV       += P^T dO ;   dP = dO V^T ;   dS = scale * P o (dP - delta[q]) ;   dQ += dS K ;   dK += dS^T Q
// Two kernels so that no accumulation crosses workgroups (no atomics, bitwise reproducible):
//   attn_bwd_kv : one workgroup = 64 keys of a (batch, head); each wave keeps dK, dV of 16 keys in accumulators and
//                 sweeps the query tiles.  S^T and dP^T are computed with the KEY on the accumulator row, so that P^T
//                 and dS^T only need the per-wave LDS round trip to become A operands of the dV / dK products.
//   attn_bwd_q  : one workgroup = 64 queries; each wave keeps dQ of 16 rows and sweeps the key tiles.
// Operands are staged row-major AND (where they are the B operand of a product over the token axis) transposed in LDS.
#include "attn_common.h"

namespace {

// delta[b][h][t] = sum_d dO o O.  One wavefront per (b, t): the H*D contiguous elements are read with 16-B loads, each lane
// reduces its 8-element chunks and the lanes of one head are combined with shuffles (chunks per head = D/8, power of 2).
__global__ __launch_bounds__(256) void attn_delta_kernel(const bf16_t* __restrict__ o, const bf16_t* __restrict__ dout, float* __restrict__ delta, int B, int H,
                                  int T, int D, long long o_bs, long long o_rs, long long do_bs, long long do_rs) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);          // (b, t)
    if (row >= (long long)B * T) return;
    const int t = (int)(row % T), b = (int)(row / T);
    const bf16_t* po = o + (long long)b * o_bs + (long long)t * o_rs;
    const bf16_t* pd = dout + (long long)b * do_bs + (long long)t * do_rs;
    const int cph = D / 8;                                                         // 16-B chunks per head
    const int nchunk = H * cph;
    for (int c0 = 0; c0 < nchunk; c0 += 64) {
        const int c = c0 + lane;
        float s = 0.f;
        if (c < nchunk) {
            const bf16x8 a = *(const bf16x8*)(po + c * 8), d = *(const bf16x8*)(pd + c * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) s += (float)a[e] * (float)d[e];
        }
        for (int off = 1; off < cph && off < 64; off <<= 1) s += __shfl_xor(s, off, 64);
        if (c < nchunk && (c % cph) == 0) delta[((long long)b * H + c / cph) * T + t] = s;
    }
}

template <int D>
__global__ __launch_bounds__(256) void attn_bwd_kv_kernel(const BwdP p) {
    typedef bf16_t T;
    constexpr int DK = (D + 31) / 32 * 32, DN = D / 16;
    constexpr int LDK = DK + 8, LDT = 64 + 8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* Ks = (T*)smem;                 // [64][LDK]   keys of this workgroup
    T* Vs = Ks + 64 * LDK;            // [64][LDK]
    T* Qs = Vs + 64 * LDK;            // [64][LDK]   current query tile
    T* Os = Qs + 64 * LDK;            // [64][LDK]   dO tile
    T* Qt = Os + 64 * LDK;            // [D][LDT]    Q^T  (B operand of dK)
    T* Ot = Qt + D * LDT;             // [D][LDT]    dO^T (B operand of dV)
    T* Pw = Ot + D * LDT;             // [4][16][LDT]  P^T  per wave
    T* Sw = Pw + 4 * 16 * LDT;        // [4][16][LDT]  dS^T per wave
    float* lse_s = (float*)(Sw + 4 * 16 * LDT);   // [64]
    float* del_s = lse_s + 64;                    // [64]

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int r = lane & 15, g = lane >> 4;
    const int j0 = blockIdx.x * 64, h = blockIdx.y, b = blockIdx.z;
    const bool vec = p.vec_ok != 0;
    int klen = p.klen ? p.klen[b] : p.Tk;
    if (klen > p.Tk) klen = p.Tk;
    if (klen < 1) klen = 1;
    const T* Q = p.q + (long long)b * p.q_bs + (long long)h * D;
    const T* K = p.k + (long long)b * p.k_bs + (long long)h * D;
    const T* V = p.v + (long long)b * p.v_bs + (long long)h * D;
    const T* DO = p.dout + (long long)b * p.do_bs + (long long)h * D;
    const float* lse = p.lse + ((long long)b * p.H + h) * p.Tq;
    const float* del = p.delta + ((long long)b * p.H + h) * p.Tq;

    stage_rows<T, D, DK>(Ks, LDK, K, p.k_rs, j0, p.Tk, vec, tid);
    stage_rows<T, D, DK>(Vs, LDK, V, p.v_rs, j0, p.Tk, vec, tid);

    f32x4 dKa[DN], dVa[DN];
#pragma unroll
    for (int n = 0; n < DN; ++n) { dKa[n] = f32x4{0.f, 0.f, 0.f, 0.f}; dVa[n] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    T* Pme = Pw + w * 16 * LDT;
    T* Sme = Sw + w * 16 * LDT;
    const T* Kme = Ks + w * 16 * LDK;
    const T* Vme = Vs + w * 16 * LDK;

    __syncthreads();                                                // K / V tiles staged
    AFrag<T, DK> kf, vf;
    kf.load(Kme, LDK, lane);                                        // my 16 keys: fragments loaded once for the whole sweep
    vf.load(Vme, LDK, lane);
    for (int i0 = 0; i0 < p.Tq; i0 += 64) {
        __syncthreads();                                            // previous tile's operands are no longer read
        stage_rows<T, D, DK>(Qs, LDK, Q, p.q_rs, i0, p.Tq, vec, tid);
        stage_rows<T, D, DK>(Os, LDK, DO, p.do_rs, i0, p.Tq, vec, tid);
        stage_vt<T, D>(Qt, LDT, Q, p.q_rs, i0, p.Tq, vec, tid);
        stage_vt<T, D>(Ot, LDT, DO, p.do_rs, i0, p.Tq, vec, tid);
        if (tid < 64) {
            const int qq = i0 + tid;
            lse_s[tid] = qq < p.Tq ? lse[qq] : 0.f;
            del_s[tid] = qq < p.Tq ? del[qq] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int n = 0; n < 4; ++n) {                               // S^T / dP^T tile: rows = my 16 keys, cols = 16 queries
            f32x4 st = f32x4{0.f, 0.f, 0.f, 0.f}, dpt = f32x4{0.f, 0.f, 0.f, 0.f};
            kf.mma(st, Qs + n * 16 * LDK, LDK, lane);
            vf.mma(dpt, Os + n * 16 * LDK, LDK, lane);
            const int qc = n * 16 + r;
            const bool qok = (i0 + qc) < p.Tq;
            const float l = lse_s[qc], dl = del_s[qc];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int key = j0 + w * 16 + 4 * g + e;
                const float pv = (qok && key < klen) ? expf(st[e] * p.scale - l) : 0.f;
                float dm = 1.f;                                     // dropout multiplier of P[q][key] (same mask as the forward)
                if (p.drop_p > 0.f) {
                    const unsigned long long idx = (((unsigned long long)b * p.H + h) * p.Tq + (i0 + qc)) * ((p.Tk + 3) & ~3) + key;
                    dm = drop_mult_call(p.drop_seed, p.drop_stream, idx, p.drop_p, drop_inv_keep(p.drop_p));
                }
                const float ds = pv * (dpt[e] * dm - dl) * p.scale;
                Pme[(4 * g + e) * LDT + qc] = (T)(pv * dm);
                Sme[(4 * g + e) * LDT + qc] = (T)ds;
            }
        }
        __syncthreads();                                            // P^T / dS^T tiles visible
        {
            AFrag<T, 64> pf, sf;
            pf.load(Pme, LDT, lane);
            sf.load(Sme, LDT, lane);
#pragma unroll
            for (int n = 0; n < DN; ++n) {
                pf.mma(dVa[n], Ot + n * 16 * LDT, LDT, lane);
                sf.mma(dKa[n], Qt + n * 16 * LDT, LDT, lane);
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int key = j0 + w * 16 + 4 * g + e;
        if (key < p.Tk) {
            T* ok = p.dk + (long long)b * p.dk_bs + (long long)key * p.dk_rs + (long long)h * D;
            T* ov = p.dv + (long long)b * p.dv_bs + (long long)key * p.dv_rs + (long long)h * D;
#pragma unroll
            for (int n = 0; n < DN; ++n) { ok[n * 16 + r] = (T)dKa[n][e]; ov[n * 16 + r] = (T)dVa[n][e]; }
        }
    }
}

template <int D>
__global__ __launch_bounds__(256) void attn_bwd_q_kernel(const BwdP p) {
    typedef bf16_t T;
    constexpr int DK = (D + 31) / 32 * 32, DN = D / 16;
    constexpr int LDK = DK + 8, LDT = 64 + 8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* Qs = (T*)smem;                 // [64][LDK]   queries of this workgroup
    T* Os = Qs + 64 * LDK;            // [64][LDK]   dO
    T* Ks = Os + 64 * LDK;            // [64][LDK]   current key tile
    T* Vs = Ks + 64 * LDK;            // [64][LDK]
    T* Kt = Vs + 64 * LDK;            // [D][LDT]    K^T (B operand of dQ)
    T* Sw = Kt + D * LDT;             // [4][16][LDT] dS per wave

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int r = lane & 15, g = lane >> 4;
    const int i0 = blockIdx.x * 64, h = blockIdx.y, b = blockIdx.z;
    const bool vec = p.vec_ok != 0;
    int klen = p.klen ? p.klen[b] : p.Tk;
    if (klen > p.Tk) klen = p.Tk;
    if (klen < 1) klen = 1;
    const T* Q = p.q + (long long)b * p.q_bs + (long long)h * D;
    const T* K = p.k + (long long)b * p.k_bs + (long long)h * D;
    const T* V = p.v + (long long)b * p.v_bs + (long long)h * D;
    const T* DO = p.dout + (long long)b * p.do_bs + (long long)h * D;
    const float* lse = p.lse + ((long long)b * p.H + h) * p.Tq;
    const float* del = p.delta + ((long long)b * p.H + h) * p.Tq;

    stage_rows<T, D, DK>(Qs, LDK, Q, p.q_rs, i0, p.Tq, vec, tid);
    stage_rows<T, D, DK>(Os, LDK, DO, p.do_rs, i0, p.Tq, vec, tid);
    float lrow[4], drow[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int qq = i0 + w * 16 + 4 * g + e;
        lrow[e] = qq < p.Tq ? lse[qq] : 0.f;
        drow[e] = qq < p.Tq ? del[qq] : 0.f;
    }
    f32x4 dQa[DN];
#pragma unroll
    for (int n = 0; n < DN; ++n) dQa[n] = f32x4{0.f, 0.f, 0.f, 0.f};
    T* Sme = Sw + w * 16 * LDT;
    const T* Qme = Qs + w * 16 * LDK;
    const T* Ome = Os + w * 16 * LDK;

    __syncthreads();                                                // Q / dO tiles staged
    AFrag<T, DK> qf, of;
    qf.load(Qme, LDK, lane);
    of.load(Ome, LDK, lane);
    for (int j0 = 0; j0 < klen; j0 += 64) {
        __syncthreads();
        stage_rows<T, D, DK>(Ks, LDK, K, p.k_rs, j0, klen, vec, tid);
        stage_rows<T, D, DK>(Vs, LDK, V, p.v_rs, j0, klen, vec, tid);
        stage_vt<T, D>(Kt, LDT, K, p.k_rs, j0, klen, vec, tid);
        __syncthreads();
#pragma unroll
        for (int n = 0; n < 4; ++n) {                               // S / dP tile: rows = my 16 queries, cols = 16 keys
            f32x4 st = f32x4{0.f, 0.f, 0.f, 0.f}, dpt = f32x4{0.f, 0.f, 0.f, 0.f};
            qf.mma(st, Ks + n * 16 * LDK, LDK, lane);
            of.mma(dpt, Vs + n * 16 * LDK, LDK, lane);
            const bool kok = (j0 + n * 16 + r) < klen;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float pv = kok ? expf(st[e] * p.scale - lrow[e]) : 0.f;
                float dm = 1.f;
                if (p.drop_p > 0.f) {
                    const unsigned long long idx = (((unsigned long long)b * p.H + h) * p.Tq + (i0 + w * 16 + 4 * g + e)) * ((p.Tk + 3) & ~3) + (j0 + n * 16 + r);
                    dm = drop_mult_call(p.drop_seed, p.drop_stream, idx, p.drop_p, drop_inv_keep(p.drop_p));
                }
                Sme[(4 * g + e) * LDT + n * 16 + r] = (T)(pv * (dpt[e] * dm - drow[e]) * p.scale);
            }
        }
        __syncthreads();
        {
            AFrag<T, 64> sf;
            sf.load(Sme, LDT, lane);
#pragma unroll
            for (int n = 0; n < DN; ++n) sf.mma(dQa[n], Kt + n * 16 * LDT, LDT, lane);
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int qq = i0 + w * 16 + 4 * g + e;
        if (qq < p.Tq) {
            T* o = p.dq + (long long)b * p.dq_bs + (long long)qq * p.dq_rs + (long long)h * D;
#pragma unroll
            for (int n = 0; n < DN; ++n) o[n * 16 + r] = (T)dQa[n][e];
        }
    }
}

template <int D>
int launch_bwd(const BwdP& p, hipStream_t st) {
    constexpr int DK = (D + 31) / 32 * 32;
    constexpr int LDK = DK + 8, LDT = 64 + 8;
    const int lds_kv = 2 * (4 * 64 * LDK + 2 * D * LDT + 2 * 4 * 16 * LDT) + 2 * 64 * 4;
    const int lds_q = 2 * (4 * 64 * LDK + D * LDT + 4 * 16 * LDT);
    static bool done = false;
    if (!done) {
        if (hipFuncSetAttribute((const void*)attn_bwd_kv_kernel<D>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_kv) != hipSuccess ||
            hipFuncSetAttribute((const void*)attn_bwd_q_kernel<D>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_q) != hipSuccess) {
            av_set_error("av_attention_bwd: cannot raise dynamic LDS (%d / %d)", lds_kv, lds_q);
            return AV_ERR_LAUNCH;
        }
        done = true;
    }
    hipLaunchKernelGGL(attn_bwd_kv_kernel<D>, dim3((p.Tk + 63) / 64, p.H, p.B), dim3(256), lds_kv, st, p);
    AV_LAUNCH_CHECK();
    hipLaunchKernelGGL(attn_bwd_q_kernel<D>, dim3((p.Tq + 63) / 64, p.H, p.B), dim3(256), lds_q, st, p);
    AV_LAUNCH_CHECK();
    return AV_OK;
}

}  // namespace

extern "C" int av_attention_bwd(const void* q, const void* k, const void* v, const void* o, const void* dout, const float* lse,
                                float* delta_ws, void* dq, void* dk, void* dv, int B, int H, int Tq, int Tk, int D,
                                const long long* strides /* 16: (bs, rs) of q,k,v,o,dout,dq,dk,dv */, const int* klen, float scale,
                                float drop_p, unsigned long long drop_seed, unsigned int drop_stream, void* stream) {
    return av_attention_bwd_mask(q, k, v, o, dout, lse, delta_ws, dq, dk, dv, B, H, Tq, Tk, D, strides, klen, scale, drop_p, drop_seed, drop_stream,
                                 nullptr, stream);
}

extern "C" int av_attention_bwd_mask(const void* q, const void* k, const void* v, const void* o, const void* dout, const float* lse,
                                     float* delta_ws, void* dq, void* dk, void* dv, int B, int H, int Tq, int Tk, int D,
                                     const long long* strides, const int* klen, float scale, float drop_p, unsigned long long drop_seed,
                                     unsigned int drop_stream, const void* drop_mask, void* stream) {
    AV_CHECK(q && k && v && o && dout && lse && delta_ws && dq && dk && dv && strides, "av_attention_bwd: null pointer");
    AV_CHECK(B > 0 && H > 0 && Tq > 0 && Tk > 0, "av_attention_bwd: bad shape");
    BwdP p;
    p.q = (const bf16_t*)q; p.k = (const bf16_t*)k; p.v = (const bf16_t*)v; p.dout = (const bf16_t*)dout;
    p.lse = lse; p.delta = delta_ws; p.dq = (bf16_t*)dq; p.dk = (bf16_t*)dk; p.dv = (bf16_t*)dv; p.klen = klen;
    p.B = B; p.H = H; p.Tq = Tq; p.Tk = Tk;
    p.q_bs = strides[0]; p.q_rs = strides[1]; p.k_bs = strides[2]; p.k_rs = strides[3]; p.v_bs = strides[4]; p.v_rs = strides[5];
    const long long o_bs = strides[6], o_rs = strides[7];
    p.o = (const bf16_t*)o; p.o_bs = o_bs; p.o_rs = o_rs;
    p.do_bs = strides[8]; p.do_rs = strides[9]; p.dq_bs = strides[10]; p.dq_rs = strides[11]; p.dk_bs = strides[12]; p.dk_rs = strides[13];
    p.dv_bs = strides[14]; p.dv_rs = strides[15];
    p.scale = scale;
    AV_CHECK(drop_p >= 0.f && drop_p < 1.f, "av_attention_bwd: drop_p=%f out of [0,1)", drop_p);
    p.drop_p = drop_p; p.drop_seed = drop_seed; p.drop_stream = drop_stream;
    AV_CHECK(!drop_mask || (D == 64 && Tq <= 256 && Tk <= 256 && (uintptr_t)drop_mask % 32 == 0),
             "av_attention_bwd_mask: the keep-bit mask belongs to the whole-sequence kernels (head_dim 64, T <= 256), 32-byte aligned");
    p.dmask = (const unsigned long long*)drop_mask;
    auto al = [&](const void* ptr, long long bs, long long rs) {
        return ((uintptr_t)ptr % 16 == 0) && ((bs * 2) % 16 == 0) && ((rs * 2) % 16 == 0) && ((D * 2) % 16 == 0);
    };
    p.vec_ok = al(q, p.q_bs, p.q_rs) && al(k, p.k_bs, p.k_rs) && al(v, p.v_bs, p.v_rs) && al(dout, p.do_bs, p.do_rs);
    hipStream_t st = (hipStream_t)stream;
    const long long n = (long long)B * H * Tq;
    AV_CHECK(D % 8 == 0 && ((D / 8) & (D / 8 - 1)) == 0 && (D / 8 <= 64) && (uintptr_t)o % 16 == 0 && (uintptr_t)dout % 16 == 0 &&
             (o_bs * 2) % 16 == 0 && (o_rs * 2) % 16 == 0 && (p.do_bs * 2) % 16 == 0 && (p.do_rs * 2) % 16 == 0,
             "av_attention_bwd: o / dout must be 16-byte aligned views and D/8 a power of two");
    (void)n;
    {
        const int rc = av_attention_short_bwd_try(p, D, st);       // whole-sequence kernel (attention_short.hip): D = 64, T <= 256
        if (rc != AV_SHORT_NOT_TAKEN) return rc;
    }
    AV_CHECK(!drop_mask, "av_attention_bwd_mask: the whole-sequence kernel did not take this call: the stored mask cannot be used");
    hipLaunchKernelGGL(attn_delta_kernel, dim3((unsigned)(((long long)B * Tq + 3) / 4)), dim3(256), 0, st, (const bf16_t*)o, (const bf16_t*)dout,
                       delta_ws, B, H, Tq, D, o_bs, o_rs, p.do_bs, p.do_rs);
    AV_LAUNCH_CHECK();
    switch (D) {
        case 16: return launch_bwd<16>(p, st);
        case 32: return launch_bwd<32>(p, st);
        case 64: return launch_bwd<64>(p, st);
        case 128: return launch_bwd<128>(p, st);
        default: av_set_error("av_attention_bwd: head_dim %d not in {16,32,64,128}", D); return AV_ERR_ARG;
    }
}
