// Whole-sequence multi-head attention for SHORT sequences (bf16, head_dim 64, T <= 256): forward and backward of
// hf:438-463 (wav2vec2 self-attention; T_enc = 49 / 199 for the 1 s / 4 s clips).  At these lengths a flash-style
// sweep over key tiles is latency-bound (a barrier pair and a global round trip per 64 keys), so here ONE workgroup
// (8 wavefronts) owns one (batch, head): every operand of the sequence is staged into LDS once, there is a single
// barrier per phase, and each wavefront walks its 16-row tiles with no further synchronisation.
//
//  * products are oriented so that the probability tile is already the next MFMA's B operand: S^T = K Q^T puts the
//    query on the lane and 4 keys in the accumulator registers, which IS the [k = key][col = query] fragment of
//    O^T = V^T P^T (two 16-key tiles make one K = 32 step; k-slot (g, j) <-> key 16*(2tp + j/4) + 4g + j%4).  No LDS
//    round trip for P, no online-softmax rescale: the whole row of scores lives in registers.
//  * the transposed operand (V^T, and in the backward dO^T, Q^T, K^T) comes from the ROW-major LDS image through
//    ds_read_b64_tr_b16 (gfx950 transpose read: a 16-lane group reads 4 rows x 16 columns and receives them column-major,
//    which matches that k-slot map).  One image serves row reads (ds_read_b128) and transposed reads: 128-B rows with
//    the 16-B chunk XOR-swizzled by (row & 7) are conflict-free for both.
//  * backward: phase A (LDS = Q, dO; a wavefront owns 16 keys) accumulates dK^T, dV^T from S = Q K^T; phase B
//    (LDS = K, V; a wavefront owns 16 queries) accumulates dQ^T from S^T = K Q^T.  delta = rowsum(dO o O) is computed while
//    dO is staged.  No atomics, no cross-workgroup accumulation: bitwise reproducible.
// Dropout on the probabilities uses the same (seed, stream, index) Philox mask as the tiled kernels.
#include <stdlib.h>

#include "attn_common.h"

namespace {

constexpr int NW = 8;                          // wavefronts per workgroup
constexpr int NT = NW * 64;
constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;

typedef __attribute__((address_space(3))) bf16x4* lds_b4_t;

// element offset of 16-B chunk `ch` (0..7) of row `row` in a [rows][64] bf16 image
__device__ __forceinline__ int swz(int row, int ch) { return row * 64 + ((ch ^ (row & 7)) << 3); }

__device__ __forceinline__ bf16x8 row_frag(const bf16_t* img, int row, int ch) { return *(const bf16x8*)(img + swz(row, ch)); }

// transposed read of rows r0 + 4g .. r0 + 4g + 3 (g = lane >> 4), columns 16n .. 16n + 15: the lane receives column
// 16n + (lane & 15), element q = row r0 + 4g + q.  Lane 4q + pp of the group supplies the address of row q, columns 4pp..4pp+3.
__device__ __forceinline__ bf16x4 tr_read(const bf16_t* img, int r0, int n, int lane) {
    const int row = r0 + 4 * (lane >> 4) + ((lane >> 2) & 3), pp = lane & 3;
    const bf16_t* a = img + row * 64 + (((2 * n + (pp >> 1)) ^ (row & 7)) << 3) + 4 * (pp & 1);
    return AV_DS_READ_TR16_B64((lds_b4_t)a);
}
// A fragment [16 columns of the image][k = 32 rows r0 .. r0+31] in the k-slot order of pack8()
__device__ __forceinline__ bf16x8 tr_frag(const bf16_t* img, int r0, int n, int lane) {
    const bf16x4 lo = tr_read(img, r0, n, lane), hi = tr_read(img, r0 + 16, n, lane);
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}
// B fragment from two accumulator tiles (rows 4g + e of 16-row tiles 2tp and 2tp + 1)
__device__ __forceinline__ bf16x8 pack8(const f32x4& a, const f32x4& b) {
    bf16x8 v;
    v[0] = (bf16_t)a[0]; v[1] = (bf16_t)a[1]; v[2] = (bf16_t)a[2]; v[3] = (bf16_t)a[3];
    v[4] = (bf16_t)b[0]; v[5] = (bf16_t)b[1]; v[6] = (bf16_t)b[2]; v[7] = (bf16_t)b[3];
    return v;
}
__device__ __forceinline__ f32x4 mfma(const bf16x8& a, const bf16x8& b, const f32x4& c) {
    return AV_MFMA_F32_16X16X32_LP(a, b, c, 0, 0, 0);
}

// rows [0, nvalid) of src (row stride rs elements, 64 contiguous) -> swizzled image of npad rows, zero-filled beyond nvalid
__device__ __forceinline__ void stage_img(bf16_t* img, const bf16_t* __restrict__ src, long long rs, int nvalid, int npad, int tid) {
    for (int c0 = tid; c0 < npad * 8; c0 += 4 * NT) {
        uint4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int c = c0 + u * NT, row = c >> 3, ch = c & 7;
            v[u] = make_uint4(0, 0, 0, 0);
            if (row < nvalid) v[u] = *(const uint4*)(src + (long long)row * rs + ch * 8);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int c = c0 + u * NT, row = c >> 3, ch = c & 7;
            if (row < npad) *(uint4*)(img + swz(row, ch)) = v[u];
        }
    }
}

// ---------------------------------------------------------------------------------------------------- dropout keep bits
// One Philox evaluation per probability and step: thread = (b, h, query tile, lane (r, g)) writes the 64 keep bits of query 16 qt + r
// for keys 16 t + 4 g + e (bit 4 t + e, t < 16) - exactly the (seed, stream, index) masks the kernels below would generate themselves
// (index = ((b H + h) Tq + q) Tk4 + key).  The forward and both phases of the backward then read 1 bit per probability.
__global__ __launch_bounds__(256) void attn_dropmask_kernel(uint2* __restrict__ mask, int B, int H, int Tq, int Tk, float drop_p,
                                                            unsigned long long seed, unsigned stream_id, long long n) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int lane = (int)(i & 63), r = lane & 15, g = lane >> 4;
    const int nqt = (Tq + 15) >> 4;
    const long long bhq = i >> 6;
    const int qt = (int)(bhq % nqt);
    const long long bh = bhq / nqt;
    const int q = qt * 16 + r;
    unsigned lo = 0u, hi = 0u;
    if (q < Tq) {
        const unsigned long long base = ((unsigned long long)bh * Tq + q) * ((Tk + 3) & ~3);
        const int nt = (Tk + 15) >> 4;
        const unsigned thr = (unsigned)__builtin_ceilf(drop_p * 65536.0f);     // u = k / 65536 >= p  <=>  k >= ceil(65536 p)
        for (int t = 0; t < nt; ++t) {
            const unsigned nib = drop_keep4(seed, stream_id, base + (16 * t + 4 * g), thr);
            if (t < 8) lo |= nib << (4 * t); else hi |= nib << (4 * (t - 8));
        }
    }
    mask[i] = make_uint2(lo, hi);
}

// ---------------------------------------------------------------------------------------------------- forward
template <int NKP, int DROP>   // 32-key pairs: keys padded to 32 * NKP; DROP: 0 none, 1 generated masks, 2 precomputed keep bits
__global__ __launch_bounds__(NT, 4) void attn_fwd_short_kernel(const AttnP p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16_t* Ks = (bf16_t*)smem;
    bf16_t* Vs = Ks + NKP * 32 * 64;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r = lane & 15, g = lane >> 4;
    const int h = blockIdx.x, b = blockIdx.y;
    int klen = p.klen ? p.klen[b] : p.Tk;
    if (klen > p.Tk) klen = p.Tk;
    if (klen < 1) klen = 1;
    const bf16_t* Q = (const bf16_t*)p.q + (long long)b * p.q_bs + (long long)h * 64;
    const bf16_t* K = (const bf16_t*)p.k + (long long)b * p.k_bs + (long long)h * 64;
    const bf16_t* V = (const bf16_t*)p.v + (long long)b * p.v_bs + (long long)h * 64;
    stage_img(Ks, K, p.k_rs, p.Tk, NKP * 32, tid);
    stage_img(Vs, V, p.v_rs, p.Tk, NKP * 32, tid);
    __syncthreads();

    const float c = p.scale * LOG2E;
    const int nqt = (p.Tq + 15) >> 4;
    for (int qt = w; qt < nqt; qt += NW) {                      // wave-uniform: EXEC stays all ones for the transposed reads
        asm volatile("" ::: "memory");                         // keep the (loop-invariant) LDS fragment reads inside the loop: 28 hoisted fragments would spill
        const int qrow = qt * 16 + r;
        const bf16_t* qp = Q + (long long)(qrow < p.Tq ? qrow : p.Tq - 1) * p.q_rs + 8 * g;
        const bf16x8 qf0 = *(const bf16x8*)qp, qf1 = *(const bf16x8*)(qp + 32);
        f32x4 S[2 * NKP];
        float mx = -INFINITY;
#pragma unroll
        for (int t = 0; t < 2 * NKP; ++t) {                    // S^T tile t: rows = keys 16t + 4g + e, column = my query
            f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f};
            a = mfma(row_frag(Ks, 16 * t + r, g), qf0, a);
            a = mfma(row_frag(Ks, 16 * t + r, 4 + g), qf1, a);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                a[e] = (16 * t + 4 * g + e) < klen ? a[e] * c : -INFINITY;
                mx = fmaxf(mx, a[e]);
            }
            S[t] = a;
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float sum = 0.f;
#pragma unroll
        for (int t = 0; t < 2 * NKP; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float pv = __builtin_amdgcn_exp2f(S[t][e] - mx);
                sum += pv;
                S[t][e] = pv;
            }
        if (DROP) {
            const float ik = drop_inv_keep(p.drop_p);
            if constexpr (DROP == 2) {
                // keep bits precomputed by attn_dropmask_kernel (the same Philox masks, evaluated ONCE per step and element instead of in the
                // forward and in both phases of the backward): bit 4 t + e of my word <-> key 16 t + 4 g + e of my query
                const uint2 mb = ((const uint2*)p.dmask)[((((long long)b * p.H + h) * nqt + qt) << 6) + lane];
#pragma unroll
                for (int t = 0; t < 2 * NKP; ++t)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        S[t][e] = (((4 * t + e < 32 ? mb.x : mb.y) >> ((4 * t + e) & 31)) & 1u) ? S[t][e] * ik : 0.f;
            } else {
                const unsigned long long base = (((unsigned long long)b * p.H + h) * p.Tq + qrow) * ((p.Tk + 3) & ~3);      // multiple of 4
#pragma unroll
                for (int t = 0; t < 2 * NKP; ++t) {            // my 4 keys of tile t are consecutive and 4-aligned: ONE Philox call
                    float m4[4];
                    drop_mult4(p.drop_seed, p.drop_stream, base + (16 * t + 4 * g), p.drop_p, ik, m4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) S[t][e] *= m4[e];
                }
            }
        }
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        f32x4 O[4];
#pragma unroll
        for (int n = 0; n < 4; ++n) O[n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int tp = 0; tp < NKP; ++tp) {                     // O^T[d][q] += V^T[d][32 keys] P^T[32 keys][q]
            const bf16x8 pf = pack8(S[2 * tp], S[2 * tp + 1]);
#pragma unroll
            for (int n = 0; n < 4; ++n) O[n] = mfma(tr_frag(Vs, 32 * tp, n, lane), pf, O[n]);
        }
        if (qrow < p.Tq) {
            const float inv = 1.0f / sum;
            bf16_t* o = (bf16_t*)p.o + (long long)b * p.o_bs + (long long)qrow * p.o_rs + (long long)h * 64 + 4 * g;
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                bf16x4 ov;
#pragma unroll
                for (int e = 0; e < 4; ++e) ov[e] = (bf16_t)(O[n][e] * inv);
                *(bf16x4*)(o + 16 * n) = ov;
            }
            if (g == 0 && p.lse) p.lse[((long long)b * p.H + h) * p.Tq + qrow] = (mx + __builtin_amdgcn_logf(sum)) * LN2;
        }
    }
}

// ---------------------------------------------------------------------------------------------------- forward, one exposed round trip
// Same arithmetic as attn_fwd_short_kernel; what changes is WHEN memory is touched.  The kernel above pays four serial global round
// trips per workgroup before its first MFMA (K image, V image - each through registers, the LDS store waits for the loads - then the
// Q fragments of the first and of the second query tile inside the tile loop) and was latency-bound (42 us at 64 x 16 x 199: 2.1 TB/s of
// algorithmic traffic, 8.6 % MFMA).  Here every byte the workgroup needs is requested in the first ~20 instructions: the K and V images
// by LDS-DMA (global_load_lds_dwordx4: one wavefront instruction fills 8 swizzled 128-B rows, source chunk = LDS chunk ^ (row & 7), no
// registers), the Q fragments of BOTH query tiles of the wavefront into registers; one vmcnt(0) + one barrier, then pure LDS / MFMA work.
// Rows beyond Tk are clamped to the last row instead of zero-filled: their scores are masked to -inf (P = 0 exactly) and V is finite.
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// image rows [8 grp, 8 grp + 8) <- src rows clamped to nvalid - 1 (one instruction per wavefront; grp must be wave-uniform)
__device__ __forceinline__ void dma_rows8(bf16_t* img, const bf16_t* __restrict__ src, long long rs, int nvalid, int grp, int lane) {
    const int sub = lane >> 3, pch = lane & 7;
    int row = grp * 8 + sub;
    row = row < nvalid ? row : nvalid - 1;
    const bf16_t* s = src + (long long)row * rs + ((pch ^ sub) << 3);
    __builtin_amdgcn_global_load_lds((gptr_t)s, (lptr_t)((char*)img + grp * 1024), 16, 0, 0);
}

// Diagnostic builds only (-DAV_ATTN_STAMPS, tools/attn_stamps.py; never the product library): s_memrealtime (100 MHz) per workgroup at entry /
// operands landed / end of the arithmetic of wavefront 0's first query tile / exit
#ifdef AV_ATTN_STAMPS
__device__ unsigned long long g_attn_stamps[4096 * 4];
#define AV_ASTAMP(SLOT) do { if (threadIdx.x == 0) { const unsigned bid_ = blockIdx.y * gridDim.x + blockIdx.x; if (bid_ < 4096) g_attn_stamps[bid_ * 4 + (SLOT)] = __builtin_amdgcn_s_memrealtime(); } } while (0)
#else
#define AV_ASTAMP(SLOT) do { } while (0)
#endif

template <int NKP, int DROP>
__global__ __launch_bounds__(NT, 4) void attn_fwd_short2_kernel(const AttnP p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    AV_ASTAMP(0);
    bf16_t* Ks = (bf16_t*)smem;
    bf16_t* Vs = Ks + NKP * 32 * 64;
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), r = lane & 15, g = lane >> 4;
    const int h = blockIdx.x, b = blockIdx.y;
    int klen = p.klen ? p.klen[b] : p.Tk;
    if (klen > p.Tk) klen = p.Tk;
    if (klen < 1) klen = 1;
    klen = __builtin_amdgcn_readfirstlane(klen);                // wave-uniform: whole key tiles below klen skip the padding test
    const bf16_t* Q = (const bf16_t*)p.q + (long long)b * p.q_bs + (long long)h * 64;
    const bf16_t* K = (const bf16_t*)p.k + (long long)b * p.k_bs + (long long)h * 64;
    const bf16_t* V = (const bf16_t*)p.v + (long long)b * p.v_bs + (long long)h * 64;
    constexpr int NG = NKP * 4;                                 // 8-row groups per image
    for (int i = w; i < 2 * NG; i += NW) {                      // wave-uniform
        if (i < NG) dma_rows8(Ks, K, p.k_rs, p.Tk, i, lane);
        else dma_rows8(Vs, V, p.v_rs, p.Tk, i - NG, lane);
    }
    const int nqt = (p.Tq + 15) >> 4;                           // <= 16: at most two query tiles per wavefront
    bf16x8 qf[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int qrow = (w + NW * i) * 16 + r;
        const bf16_t* qp = Q + (long long)(qrow < p.Tq ? qrow : p.Tq - 1) * p.q_rs + 8 * g;
        qf[i][0] = *(const bf16x8*)qp; qf[i][1] = *(const bf16x8*)(qp + 32);
    }
    uint2 mb[2] = {make_uint2(0u, 0u), make_uint2(0u, 0u)};
    if constexpr (DROP == 2) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int qt = (w + NW * i) < nqt ? (w + NW * i) : nqt - 1;
            mb[i] = ((const uint2*)p.dmask)[((((long long)b * p.H + h) * nqt + qt) << 6) + lane];
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    AV_ASTAMP(1);

    const float c = p.scale * LOG2E;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int qt = w + NW * i;
        if (qt >= nqt) break;                                   // wave-uniform
        if (i == 1) AV_ASTAMP(2);
        asm volatile("" ::: "memory");                          // keep the LDS fragment reads of the two tiles apart (no cross-tile hoisting: spills)
        const int qrow = qt * 16 + r;
        f32x4 S[2 * NKP];
        float mx = -INFINITY;
#pragma unroll
        for (int t = 0; t < 2 * NKP; ++t) {
            f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f};
            a = mfma(row_frag(Ks, 16 * t + r, g), qf[i][0], a);
            a = mfma(row_frag(Ks, 16 * t + r, 4 + g), qf[i][1], a);
            S[t] = a;
        }
        // the kernel is bound by vector issue: the maximum is taken over the RAW scores (the scale is positive and folds into the
        // exponent's FMA), and only key tiles that reach past klen pay the padding test (wave-uniform branches, kept out of the MFMA loop
        // above so that its LDS reads and MFMAs stay one schedulable block)
#pragma unroll
        for (int t = 0; t < 2 * NKP; ++t)
            if (16 * t + 16 > klen) {
#pragma unroll
                for (int e = 0; e < 4; ++e) S[t][e] = (16 * t + 4 * g + e) < klen ? S[t][e] : -INFINITY;
            }
#pragma unroll
        for (int t = 0; t < 2 * NKP; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e) mx = fmaxf(mx, S[t][e]);
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float nmxc = -(mx * c);
        float sum = 0.f;
#pragma unroll
        for (int t = 0; t < 2 * NKP; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(S[t][e], c, nmxc));      // exp2(-inf) = 0 for masked keys
                sum += pv;
                S[t][e] = pv;
            }
        float ikf = 1.0f;                                       // DROP 2: 1 / (1 - p) multiplies the normalisation instead of every probability
        if (DROP) {
            const float ik = drop_inv_keep(p.drop_p);
            if constexpr (DROP == 2) {
                ikf = ik;
#pragma unroll
                for (int t = 0; t < 2 * NKP; ++t)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        S[t][e] = (((4 * t + e < 32 ? mb[i].x : mb[i].y) >> ((4 * t + e) & 31)) & 1u) ? S[t][e] : 0.f;
            } else {
                const unsigned long long base = (((unsigned long long)b * p.H + h) * p.Tq + qrow) * ((p.Tk + 3) & ~3);
#pragma unroll
                for (int t = 0; t < 2 * NKP; ++t) {
                    float m4[4];
                    drop_mult4(p.drop_seed, p.drop_stream, base + (16 * t + 4 * g), p.drop_p, ik, m4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) S[t][e] *= m4[e];
                }
            }
        }
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        f32x4 O[4];
#pragma unroll
        for (int n = 0; n < 4; ++n) O[n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int tp = 0; tp < NKP; ++tp) {
            const bf16x8 pf = pack8(S[2 * tp], S[2 * tp + 1]);
#pragma unroll
            for (int n = 0; n < 4; ++n) O[n] = mfma(tr_frag(Vs, 32 * tp, n, lane), pf, O[n]);
        }
        if (qrow < p.Tq) {
            const float inv = ikf / sum;
            bf16_t* o = (bf16_t*)p.o + (long long)b * p.o_bs + (long long)qrow * p.o_rs + (long long)h * 64 + 4 * g;
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                bf16x4 ov;
#pragma unroll
                for (int e = 0; e < 4; ++e) ov[e] = (bf16_t)(O[n][e] * inv);
                *(bf16x4*)(o + 16 * n) = ov;
            }
            if (g == 0 && p.lse) p.lse[((long long)b * p.H + h) * p.Tq + qrow] = (mx * c + __builtin_amdgcn_logf(sum)) * LN2;
        }
    }
#ifdef AV_ATTN_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // diagnostic: the exit stamp includes the completion of this wavefront's stores
#endif
    AV_ASTAMP(3);
}

// ---------------------------------------------------------------------------------------------------- forward, T > 256
// Same register-resident-P scheme, keys streamed through LDS in chunks of 128 with an online softmax: a workgroup owns 128 queries
// (8 wavefronts x 16) of one (batch, head); per chunk a wavefront computes S^T (8 tiles), folds the chunk maximum into its running
// maximum, rescales its running sum and O^T by 2^(m_old - m_new) and adds the chunk's P V.  (config 3: T_enc = 749.)
template <bool DROP>
__global__ __launch_bounds__(NT, 4) void attn_fwd_long_kernel(const AttnP p) {
    constexpr int NKP = 4, KC = 32 * NKP;                      // 128 keys per chunk (256 would need 64 S registers: spills at 4 waves per SIMD)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16_t* Ks = (bf16_t*)smem;
    bf16_t* Vs = Ks + KC * 64;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r = lane & 15, g = lane >> 4;
    const int h = blockIdx.x, b = blockIdx.y;
    int klen = p.klen ? p.klen[b] : p.Tk;
    if (klen > p.Tk) klen = p.Tk;
    if (klen < 1) klen = 1;
    const bf16_t* Q = (const bf16_t*)p.q + (long long)b * p.q_bs + (long long)h * 64;
    const bf16_t* K = (const bf16_t*)p.k + (long long)b * p.k_bs + (long long)h * 64;
    const bf16_t* V = (const bf16_t*)p.v + (long long)b * p.v_bs + (long long)h * 64;
    const float c = p.scale * LOG2E;
    const int qt = blockIdx.z * NW + w;                         // my 16-query tile (wave-uniform)
    const bool active = qt * 16 < p.Tq;
    const int qrow = qt * 16 + r;
    const bf16_t* qp = Q + (long long)(qrow < p.Tq ? qrow : p.Tq - 1) * p.q_rs + 8 * g;
    const bf16x8 qf0 = *(const bf16x8*)qp, qf1 = *(const bf16x8*)(qp + 32);
    float m_run = -INFINITY, l_run = 0.f;
    f32x4 O[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) O[n] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int nchunk = (klen + KC - 1) / KC;                    // chunks past klen hold only masked keys
    for (int kc = 0; kc < nchunk; ++kc) {
        const int k0 = kc * KC;
        __syncthreads();                                        // every wavefront is done with the previous chunk
        stage_img(Ks, K + (long long)k0 * p.k_rs, p.k_rs, p.Tk - k0 < KC ? p.Tk - k0 : KC, KC, tid);
        stage_img(Vs, V + (long long)k0 * p.v_rs, p.v_rs, p.Tk - k0 < KC ? p.Tk - k0 : KC, KC, tid);
        __syncthreads();
        if (!active) continue;                                  // wave-uniform
        asm volatile("" ::: "memory");
        f32x4 S[2 * NKP];
        float mx = -INFINITY;
#pragma unroll
        for (int t = 0; t < 2 * NKP; ++t) {
            f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f};
            a = mfma(row_frag(Ks, 16 * t + r, g), qf0, a);
            a = mfma(row_frag(Ks, 16 * t + r, 4 + g), qf1, a);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                a[e] = (k0 + 16 * t + 4 * g + e) < klen ? a[e] * c : -INFINITY;
                mx = fmaxf(mx, a[e]);
            }
            S[t] = a;
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);                   // finite: the first chunk holds key 0 < klen
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        float sum = 0.f;
#pragma unroll
        for (int t = 0; t < 2 * NKP; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float pv = __builtin_amdgcn_exp2f(S[t][e] - m_new);
                sum += pv;
                S[t][e] = pv;
            }
        if (DROP) {
            const float ik = drop_inv_keep(p.drop_p);
            const unsigned long long base = (((unsigned long long)b * p.H + h) * p.Tq + qrow) * ((p.Tk + 3) & ~3) + k0;
#pragma unroll
            for (int t = 0; t < 2 * NKP; ++t) {
                float m4[4];
                drop_mult4(p.drop_seed, p.drop_stream, base + (16 * t + 4 * g), p.drop_p, ik, m4);
#pragma unroll
                for (int e = 0; e < 4; ++e) S[t][e] *= m4[e];
            }
        }
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        l_run = l_run * alpha + sum;
        m_run = m_new;
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int e = 0; e < 4; ++e) O[n][e] *= alpha;
#pragma unroll
        for (int tp = 0; tp < NKP; ++tp) {
            const bf16x8 pf = pack8(S[2 * tp], S[2 * tp + 1]);
#pragma unroll
            for (int n = 0; n < 4; ++n) O[n] = mfma(tr_frag(Vs, 32 * tp, n, lane), pf, O[n]);
        }
    }
    if (active && qrow < p.Tq) {
        const float inv = 1.0f / l_run;
        bf16_t* o = (bf16_t*)p.o + (long long)b * p.o_bs + (long long)qrow * p.o_rs + (long long)h * 64 + 4 * g;
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            bf16x4 ov;
#pragma unroll
            for (int e = 0; e < 4; ++e) ov[e] = (bf16_t)(O[n][e] * inv);
            *(bf16x4*)(o + 16 * n) = ov;
        }
        if (g == 0 && p.lse) p.lse[((long long)b * p.H + h) * p.Tq + qrow] = (m_run + __builtin_amdgcn_logf(l_run)) * LN2;
    }
}

// Dropout multipliers of the kv-owner phases: a lane holds P[q = qb + e][key = krow], e = 0..3 - four different Philox rows.  The four
// lanes of a quad (keys krow & ~3 .. + 3, same four queries) share the four blocks: lane j of the quad evaluates the block of query
// qb + j (its four keys) and the values travel by quad-permute DPP moves, so a lane pays ONE Philox call for its four elements.
template <int SRC>
__device__ __forceinline__ float quad_from(float v) {          // value of lane SRC of my quad
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), SRC * 0x55, 0xf, 0xf, true));
}
__device__ __forceinline__ void drop_quad4(unsigned long long seed, unsigned stream, unsigned long long row0_idx, unsigned long long pitch,
                                           int krow, int r, float p, float ik, float (&dm)[4]) {
    float m4[4];                                                // my block: query qb + (r & 3), keys (krow & ~3) + 0..3
    drop_mult4(seed, stream, row0_idx + (unsigned long long)(r & 3) * pitch + (unsigned long long)(krow & ~3), p, ik, m4);
    const int pos = r & 3;                                      // my key's position inside every block
    float got[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        got[0][j] = quad_from<0>(m4[j]); got[1][j] = quad_from<1>(m4[j]); got[2][j] = quad_from<2>(m4[j]); got[3][j] = quad_from<3>(m4[j]);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) dm[e] = pos == 0 ? got[e][0] : (pos == 1 ? got[e][1] : (pos == 2 ? got[e][2] : got[e][3]));
}

// ---------------------------------------------------------------------------------------------------- backward
template <int DROP>             // 0 none, 1 generated masks, 2 precomputed keep bits
__global__ __launch_bounds__(NT, 4) void attn_bwd_short_kernel(const BwdP p, int R) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16_t* I0 = (bf16_t*)smem;                 // phase A: Q      phase B: K
    bf16_t* I1 = I0 + R * 64;                   // phase A: dO     phase B: V
    float* lse_s = (float*)(I1 + R * 64);       // [R]  log2-domain LSE of each query
    float* del_s = lse_s + R;                   // [R]  delta = sum_d dO o O
    uint2* mask_s = (uint2*)(del_s + R);        // DROP == 2: the keep bits of this (batch, head): [query tile][lane]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r = lane & 15, g = lane >> 4;
    const int h = blockIdx.x, b = blockIdx.y;
    int klen = p.klen ? p.klen[b] : p.Tk;
    if (klen > p.Tk) klen = p.Tk;
    if (klen < 1) klen = 1;
    klen = __builtin_amdgcn_readfirstlane(klen);
    const bf16_t* Q = p.q + (long long)b * p.q_bs + (long long)h * 64;
    const bf16_t* K = p.k + (long long)b * p.k_bs + (long long)h * 64;
    const bf16_t* V = p.v + (long long)b * p.v_bs + (long long)h * 64;
    const bf16_t* O = p.o + (long long)b * p.o_bs + (long long)h * 64;
    const bf16_t* DO = p.dout + (long long)b * p.do_bs + (long long)h * 64;
    const float* lse = p.lse + ((long long)b * p.H + h) * p.Tq;
    const float c = p.scale * LOG2E;
    const float ik = DROP ? drop_inv_keep(p.drop_p) : 1.0f;
    const unsigned long long dbase = ((unsigned long long)b * p.H + h) * p.Tq;
    const int nqt_m = (p.Tq + 15) >> 4;                         // stored keep bits: [b][h][query tile][lane] x 64 bits (forward kernel)
    const long long dbase_m = ((long long)b * p.H + h) * nqt_m;
    const int RA = ((p.Tq + 31) >> 5) << 5, RB = ((p.Tk + 31) >> 5) << 5;

    // ---- phase A operands: Q and dO images, delta and LSE per query
    stage_img(I0, Q, p.q_rs, p.Tq, RA, tid);
    for (int cb = tid; cb < RA * 8; cb += 4 * NT) {             // 8 consecutive lanes share a row; 4 rows per thread in flight:
        bf16x8 dvv[4], ovv[4];                                  // unconditional loads from clamped rows first, select afterwards
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int c0 = cb + u * NT, row = c0 >> 3, ch = c0 & 7;
            const int rr = row < p.Tq ? row : p.Tq - 1;
            dvv[u] = *(const bf16x8*)(DO + (long long)rr * p.do_rs + ch * 8);
            ovv[u] = *(const bf16x8*)(O + (long long)rr * p.o_rs + ch * 8);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int c0 = cb + u * NT, row = c0 >> 3, ch = c0 & 7;
            if (c0 < RA * 8) {                                  // wave-uniform (RA * 8 and NT are multiples of 64)
                const bool ok = row < p.Tq;
                float s = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) s += (float)dvv[u][e] * (float)ovv[u][e];
                s = ok ? s : 0.f;
                *(bf16x8*)(I1 + swz(row, ch)) = ok ? dvv[u] : bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
                s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64);
                if (ch == 0) del_s[row] = s;
            }
        }
    }
    for (int i = tid; i < R; i += NT) lse_s[i] = i < p.Tq ? lse[i] * LOG2E : INFINITY;      // padded queries: exp2(s c - inf) = 0, no per-element test
    if constexpr (DROP == 2) {
        const uint2* gm = (const uint2*)p.dmask + (dbase_m << 6);
        for (int i = tid; i < nqt_m * 64; i += NT) mask_s[i] = gm[i];
    }
    __syncthreads();

    // ---- phase A: a wavefront owns 16 keys; S = Q K^T has the key on the lane, 4 queries in the registers
    const int nkt = (p.Tk + 15) >> 4, nqp = RA >> 5;
    for (int kt = w; kt < nkt; kt += NW) {
        const int krow = kt * 16 + r;
        const long long kld = krow < p.Tk ? krow : p.Tk - 1;
        const bf16x8 kf0 = *(const bf16x8*)(K + kld * p.k_rs + 8 * g), kf1 = *(const bf16x8*)(K + kld * p.k_rs + 32 + 8 * g);
        const bf16x8 vf0 = *(const bf16x8*)(V + kld * p.v_rs + 8 * g), vf1 = *(const bf16x8*)(V + kld * p.v_rs + 32 + 8 * g);
        const float kneg = krow < klen ? 0.f : -INFINITY;           // padded / masked keys vanish in the exponent too
        const int mask_bit = 4 * kt + (r & 3), mask_sh = mask_bit & 31;
        const unsigned* mask_w = (const unsigned*)mask_s + (mask_bit >> 5);
        f32x4 dVt[4], dKt[4];
#pragma unroll
        for (int n = 0; n < 4; ++n) { dVt[n] = f32x4{0.f, 0.f, 0.f, 0.f}; dKt[n] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        for (int qp = 0; qp < nqp; ++qp) {
            f32x4 Pt[2], St[2];
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                const int q0 = 32 * qp + 16 * hf;
                f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f}, dp = f32x4{0.f, 0.f, 0.f, 0.f};
                s = mfma(row_frag(I0, q0 + r, g), kf0, s);
                s = mfma(row_frag(I0, q0 + r, 4 + g), kf1, s);
                dp = mfma(row_frag(I1, q0 + r, g), vf0, dp);
                dp = mfma(row_frag(I1, q0 + r, 4 + g), vf1, dp);
                const f32x4 l4 = *(const f32x4*)(lse_s + q0 + 4 * g), d4 = *(const f32x4*)(del_s + q0 + 4 * g);
                float dm[4] = {1.f, 1.f, 1.f, 1.f};
                if (DROP) {
                    if constexpr (DROP == 2) {
                        // forward layout: query tile q0 / 16, lane' = (r' = query % 16 = 4 g + e, g' = (key % 16) / 4 = r >> 2), bit 4 kt + (key & 3):
                        // my four queries are four consecutive 64-bit words
                        // my key's bit (4 kt + key % 4: the same 32-bit half and shift for every query of the sweep) of four consecutive
                        // 64-bit words: four 4-byte reads at an 8-byte stride and one bit-field extract each
                        const int qt_ = (q0 >> 4) < nqt_m ? (q0 >> 4) : nqt_m - 1;
                        const unsigned* mp = mask_w + 2 * ((qt_ << 6) + ((r >> 2) << 4) + 4 * g);
#pragma unroll
                        for (int e = 0; e < 4; ++e) dm[e] = __builtin_amdgcn_ubfe(mp[2 * e], (unsigned)mask_sh, 1u) ? ik : 0.f;
                    } else {
                        drop_quad4(p.drop_seed, p.drop_stream, (dbase + q0 + 4 * g) * ((p.Tk + 3) & ~3), (unsigned long long)((p.Tk + 3) & ~3), krow, r, p.drop_p, ik, dm);
                    }
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {                // vector issue bounds this kernel: one FMA into the exponent, the softmax scale
                    const float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(s[e], c, kneg - l4[e]));      // multiplies dK^T once at the end
                    Pt[hf][e] = pv * dm[e];
                    St[hf][e] = pv * __builtin_fmaf(dp[e], dm[e], -d4[e]);
                }
            }
            const bf16x8 pf = pack8(Pt[0], Pt[1]), sf = pack8(St[0], St[1]);
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                dVt[n] = mfma(tr_frag(I1, 32 * qp, n, lane), pf, dVt[n]);      // dV^T[d][key] += dO^T[d][32 q] P[32 q][key]
                dKt[n] = mfma(tr_frag(I0, 32 * qp, n, lane), sf, dKt[n]);      // dK^T[d][key] += Q^T[d][32 q] dS[32 q][key]
            }
        }
        if (krow < p.Tk) {
            bf16_t* ok = p.dk + (long long)b * p.dk_bs + (long long)krow * p.dk_rs + (long long)h * 64 + 4 * g;
            bf16_t* ov = p.dv + (long long)b * p.dv_bs + (long long)krow * p.dv_rs + (long long)h * 64 + 4 * g;
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                bf16x4 a, c4;
#pragma unroll
                for (int e = 0; e < 4; ++e) { a[e] = (bf16_t)(dKt[n][e] * p.scale); c4[e] = (bf16_t)dVt[n][e]; }
                *(bf16x4*)(ok + 16 * n) = a;
                *(bf16x4*)(ov + 16 * n) = c4;
            }
        }
    }

    // ---- phase B operands: K and V images replace Q and dO
    __syncthreads();
    stage_img(I0, K, p.k_rs, p.Tk, RB, tid);
    stage_img(I1, V, p.v_rs, p.Tk, RB, tid);
    __syncthreads();

    // ---- phase B: a wavefront owns 16 queries; S^T = K Q^T has the query on the lane, 4 keys in the registers
    const int nqt = (p.Tq + 15) >> 4, nkp = RB >> 5;
    for (int qt = w; qt < nqt; qt += NW) {
        const int qrow = qt * 16 + r;
        const long long qld = qrow < p.Tq ? qrow : p.Tq - 1;
        const bf16x8 qf0 = *(const bf16x8*)(Q + qld * p.q_rs + 8 * g), qf1 = *(const bf16x8*)(Q + qld * p.q_rs + 32 + 8 * g);
        const bf16x8 of0 = *(const bf16x8*)(DO + qld * p.do_rs + 8 * g), of1 = *(const bf16x8*)(DO + qld * p.do_rs + 32 + 8 * g);
        const bool qok = qrow < p.Tq;
        const float lq = lse_s[qrow], dq_ = del_s[qrow];            // lq = +inf for a padded query
        const uint2 mb2 = DROP == 2 ? mask_s[(qt << 6) + lane] : make_uint2(0u, 0u);                   // my query's keep bits
        const unsigned long long mbits = ((unsigned long long)mb2.y << 32) | mb2.x;
        f32x4 dQt[4];
#pragma unroll
        for (int n = 0; n < 4; ++n) dQt[n] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int tp = 0; tp < nkp; ++tp) {
            f32x4 St[2];
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                const int k0 = 32 * tp + 16 * hf;
                f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f}, dp = f32x4{0.f, 0.f, 0.f, 0.f};
                s = mfma(row_frag(I0, k0 + r, g), qf0, s);
                s = mfma(row_frag(I0, k0 + r, 4 + g), qf1, s);
                dp = mfma(row_frag(I1, k0 + r, g), of0, dp);
                dp = mfma(row_frag(I1, k0 + r, 4 + g), of1, dp);
                float m4[4] = {1.f, 1.f, 1.f, 1.f};
                if (DROP) {
                    if constexpr (DROP == 2) {
                        const unsigned nib = (unsigned)(mbits >> (4 * (2 * tp + hf))) & 15u;
#pragma unroll
                        for (int e = 0; e < 4; ++e) m4[e] = ((nib >> e) & 1u) ? ik : 0.f;
                    } else {
                        drop_mult4(p.drop_seed, p.drop_stream, (dbase + qrow) * ((p.Tk + 3) & ~3) + (k0 + 4 * g), p.drop_p, ik, m4);
                    }
                }
                float pv[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) pv[e] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[e], c, -lq));
                if (k0 + 16 > klen) {                           // wave-uniform: only key tiles that reach past klen pay the test
#pragma unroll
                    for (int e = 0; e < 4; ++e) pv[e] = (k0 + 4 * g + e) < klen ? pv[e] : 0.f;
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) St[hf][e] = pv[e] * __builtin_fmaf(dp[e], m4[e], -dq_);
            }
            const bf16x8 sf = pack8(St[0], St[1]);
#pragma unroll
            for (int n = 0; n < 4; ++n) dQt[n] = mfma(tr_frag(I0, 32 * tp, n, lane), sf, dQt[n]);   // dQ^T[d][q] += K^T[d][32 keys] dS^T[32 keys][q]
        }
        if (qok) {
            bf16_t* oq = p.dq + (long long)b * p.dq_bs + (long long)qrow * p.dq_rs + (long long)h * 64 + 4 * g;
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                bf16x4 a;
#pragma unroll
                for (int e = 0; e < 4; ++e) a[e] = (bf16_t)(dQt[n][e] * p.scale);
                *(bf16x4*)(oq + 16 * n) = a;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------- backward, T > 256
// The two phases of the kernel above as two launches with the OTHER operand streamed through LDS in chunks of 128 rows:
//   kv kernel: a workgroup owns 128 keys (a wavefront 16: K / V fragments and the dK^T / dV^T accumulators stay in registers), the
//              queries (Q, dO images, delta, LSE) arrive chunk by chunk;
//   q kernel:  a workgroup owns 128 queries (a wavefront 16: Q / dO fragments, delta, LSE, dQ^T in registers), K and V arrive in chunks.
constexpr int LC = 128;                          // rows per streamed chunk

template <bool DROP>
__global__ __launch_bounds__(NT, 4) void attn_bwd_long_kv_kernel(const BwdP p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16_t* I0 = (bf16_t*)smem;                 // Q chunk image
    bf16_t* I1 = I0 + LC * 64;                  // dO chunk image
    float* lse_s = (float*)(I1 + LC * 64);
    float* del_s = lse_s + LC;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r = lane & 15, g = lane >> 4;
    const int h = blockIdx.x, b = blockIdx.y;
    int klen = p.klen ? p.klen[b] : p.Tk;
    if (klen > p.Tk) klen = p.Tk;
    if (klen < 1) klen = 1;
    const bf16_t* Q = p.q + (long long)b * p.q_bs + (long long)h * 64;
    const bf16_t* K = p.k + (long long)b * p.k_bs + (long long)h * 64;
    const bf16_t* V = p.v + (long long)b * p.v_bs + (long long)h * 64;
    const bf16_t* O = p.o + (long long)b * p.o_bs + (long long)h * 64;
    const bf16_t* DO = p.dout + (long long)b * p.do_bs + (long long)h * 64;
    const float* lse = p.lse + ((long long)b * p.H + h) * p.Tq;
    const float c = p.scale * LOG2E;
    const float ik = DROP ? drop_inv_keep(p.drop_p) : 1.0f;
    const unsigned long long dbase = ((unsigned long long)b * p.H + h) * p.Tq;
    const int kt = blockIdx.z * NW + w;                         // my 16-key tile (wave-uniform)
    const bool active = kt * 16 < p.Tk;
    const int krow = kt * 16 + r;
    const long long kld = krow < p.Tk ? krow : p.Tk - 1;
    const bf16x8 kf0 = *(const bf16x8*)(K + kld * p.k_rs + 8 * g), kf1 = *(const bf16x8*)(K + kld * p.k_rs + 32 + 8 * g);
    const bf16x8 vf0 = *(const bf16x8*)(V + kld * p.v_rs + 8 * g), vf1 = *(const bf16x8*)(V + kld * p.v_rs + 32 + 8 * g);
    const bool kok = krow < klen;
    f32x4 dVt[4], dKt[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) { dVt[n] = f32x4{0.f, 0.f, 0.f, 0.f}; dKt[n] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    for (int q0 = 0; q0 < p.Tq; q0 += LC) {
        const int nv = p.Tq - q0 < LC ? p.Tq - q0 : LC;
        __syncthreads();                                        // every wavefront is done with the previous chunk
        stage_img(I0, Q + (long long)q0 * p.q_rs, p.q_rs, nv, LC, tid);
        {   // dO image + delta: 8 consecutive lanes share a row; unconditional loads from clamped rows, select afterwards
            bf16x8 dvv[2], ovv[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int c0 = tid + u * NT, row = c0 >> 3, ch = c0 & 7;
                const long long rr = q0 + (row < nv ? row : nv - 1);
                dvv[u] = *(const bf16x8*)(DO + rr * p.do_rs + ch * 8);
                ovv[u] = *(const bf16x8*)(O + rr * p.o_rs + ch * 8);
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int c0 = tid + u * NT, row = c0 >> 3, ch = c0 & 7;
                const bool ok = row < nv;
                float sacc = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) sacc += (float)dvv[u][e] * (float)ovv[u][e];
                sacc = ok ? sacc : 0.f;
                *(bf16x8*)(I1 + swz(row, ch)) = ok ? dvv[u] : bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
                sacc += __shfl_xor(sacc, 1, 64); sacc += __shfl_xor(sacc, 2, 64); sacc += __shfl_xor(sacc, 4, 64);
                if (ch == 0) del_s[row] = sacc;
            }
        }
        if (tid < LC) lse_s[tid] = tid < nv ? lse[q0 + tid] * LOG2E : 0.f;
        __syncthreads();
        if (!active) continue;                                  // wave-uniform
        asm volatile("" ::: "memory");
        for (int qp = 0; qp < LC / 32; ++qp) {
            if (32 * qp >= nv) break;                            // wave-uniform
            f32x4 Pt[2], St[2];
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                const int ql = 32 * qp + 16 * hf;
                f32x4 sv = f32x4{0.f, 0.f, 0.f, 0.f}, dp = f32x4{0.f, 0.f, 0.f, 0.f};
                sv = mfma(row_frag(I0, ql + r, g), kf0, sv);
                sv = mfma(row_frag(I0, ql + r, 4 + g), kf1, sv);
                dp = mfma(row_frag(I1, ql + r, g), vf0, dp);
                dp = mfma(row_frag(I1, ql + r, 4 + g), vf1, dp);
                const f32x4 l4 = *(const f32x4*)(lse_s + ql + 4 * g), d4 = *(const f32x4*)(del_s + ql + 4 * g);
                float dm[4] = {1.f, 1.f, 1.f, 1.f};
                if (DROP) drop_quad4(p.drop_seed, p.drop_stream, (dbase + q0 + ql + 4 * g) * ((p.Tk + 3) & ~3), (unsigned long long)((p.Tk + 3) & ~3), krow, r, p.drop_p, ik, dm);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int q = q0 + ql + 4 * g + e;
                    const float pv = (kok && q < p.Tq) ? __builtin_amdgcn_exp2f(sv[e] * c - l4[e]) : 0.f;
                    Pt[hf][e] = pv * dm[e];
                    St[hf][e] = pv * (dp[e] * dm[e] - d4[e]) * p.scale;
                }
            }
            const bf16x8 pf = pack8(Pt[0], Pt[1]), sf = pack8(St[0], St[1]);
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                dVt[n] = mfma(tr_frag(I1, 32 * qp, n, lane), pf, dVt[n]);
                dKt[n] = mfma(tr_frag(I0, 32 * qp, n, lane), sf, dKt[n]);
            }
        }
    }
    if (active && krow < p.Tk) {
        bf16_t* ok = p.dk + (long long)b * p.dk_bs + (long long)krow * p.dk_rs + (long long)h * 64 + 4 * g;
        bf16_t* ov = p.dv + (long long)b * p.dv_bs + (long long)krow * p.dv_rs + (long long)h * 64 + 4 * g;
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            bf16x4 a, c4;
#pragma unroll
            for (int e = 0; e < 4; ++e) { a[e] = (bf16_t)dKt[n][e]; c4[e] = (bf16_t)dVt[n][e]; }
            *(bf16x4*)(ok + 16 * n) = a;
            *(bf16x4*)(ov + 16 * n) = c4;
        }
    }
}

template <bool DROP>
__global__ __launch_bounds__(NT, 4) void attn_bwd_long_q_kernel(const BwdP p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16_t* I0 = (bf16_t*)smem;                 // K chunk image
    bf16_t* I1 = I0 + LC * 64;                  // V chunk image
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r = lane & 15, g = lane >> 4;
    const int h = blockIdx.x, b = blockIdx.y;
    int klen = p.klen ? p.klen[b] : p.Tk;
    if (klen > p.Tk) klen = p.Tk;
    if (klen < 1) klen = 1;
    const bf16_t* Q = p.q + (long long)b * p.q_bs + (long long)h * 64;
    const bf16_t* K = p.k + (long long)b * p.k_bs + (long long)h * 64;
    const bf16_t* V = p.v + (long long)b * p.v_bs + (long long)h * 64;
    const bf16_t* O = p.o + (long long)b * p.o_bs + (long long)h * 64;
    const bf16_t* DO = p.dout + (long long)b * p.do_bs + (long long)h * 64;
    const float* lse = p.lse + ((long long)b * p.H + h) * p.Tq;
    const float c = p.scale * LOG2E;
    const float ik = DROP ? drop_inv_keep(p.drop_p) : 1.0f;
    const unsigned long long dbase = ((unsigned long long)b * p.H + h) * p.Tq;
    const int qt = blockIdx.z * NW + w;
    const bool active = qt * 16 < p.Tq;
    const int qrow = qt * 16 + r;
    const long long qld = qrow < p.Tq ? qrow : p.Tq - 1;
    const bf16x8 qf0 = *(const bf16x8*)(Q + qld * p.q_rs + 8 * g), qf1 = *(const bf16x8*)(Q + qld * p.q_rs + 32 + 8 * g);
    const bf16x8 of0 = *(const bf16x8*)(DO + qld * p.do_rs + 8 * g), of1 = *(const bf16x8*)(DO + qld * p.do_rs + 32 + 8 * g);
    const bf16x8 oo0 = *(const bf16x8*)(O + qld * p.o_rs + 8 * g), oo1 = *(const bf16x8*)(O + qld * p.o_rs + 32 + 8 * g);
    float dq_ = 0.f;                                            // delta = sum_d dO o O of my query: 16 of the 64 d on this lane
#pragma unroll
    for (int e = 0; e < 8; ++e) dq_ += (float)of0[e] * (float)oo0[e] + (float)of1[e] * (float)oo1[e];
    dq_ += __shfl_xor(dq_, 16, 64);
    dq_ += __shfl_xor(dq_, 32, 64);
    const bool qok = qrow < p.Tq;
    const float lq = lse[qld] * LOG2E;
    f32x4 dQt[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) dQt[n] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int kend = klen;                                      // keys >= klen contribute nothing
    for (int k0 = 0; k0 < kend; k0 += LC) {
        const int nv = p.Tk - k0 < LC ? p.Tk - k0 : LC;
        __syncthreads();
        stage_img(I0, K + (long long)k0 * p.k_rs, p.k_rs, nv, LC, tid);
        stage_img(I1, V + (long long)k0 * p.v_rs, p.v_rs, nv, LC, tid);
        __syncthreads();
        if (!active) continue;
        asm volatile("" ::: "memory");
        for (int tp = 0; tp < LC / 32; ++tp) {
            if (k0 + 32 * tp >= kend) break;                     // wave-uniform
            f32x4 St[2];
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                const int kl = 32 * tp + 16 * hf;
                f32x4 sv = f32x4{0.f, 0.f, 0.f, 0.f}, dp = f32x4{0.f, 0.f, 0.f, 0.f};
                sv = mfma(row_frag(I0, kl + r, g), qf0, sv);
                sv = mfma(row_frag(I0, kl + r, 4 + g), qf1, sv);
                dp = mfma(row_frag(I1, kl + r, g), of0, dp);
                dp = mfma(row_frag(I1, kl + r, 4 + g), of1, dp);
                float m4[4] = {1.f, 1.f, 1.f, 1.f};
                if (DROP) drop_mult4(p.drop_seed, p.drop_stream, (dbase + qrow) * ((p.Tk + 3) & ~3) + (k0 + kl + 4 * g), p.drop_p, ik, m4);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int key = k0 + kl + 4 * g + e;
                    const float pv = (qok && key < klen) ? __builtin_amdgcn_exp2f(sv[e] * c - lq) : 0.f;
                    St[hf][e] = pv * (dp[e] * m4[e] - dq_) * p.scale;
                }
            }
            const bf16x8 sf = pack8(St[0], St[1]);
#pragma unroll
            for (int n = 0; n < 4; ++n) dQt[n] = mfma(tr_frag(I0, 32 * tp, n, lane), sf, dQt[n]);
        }
    }
    if (active && qok) {
        bf16_t* oq = p.dq + (long long)b * p.dq_bs + (long long)qrow * p.dq_rs + (long long)h * 64 + 4 * g;
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            bf16x4 a;
#pragma unroll
            for (int e = 0; e < 4; ++e) a[e] = (bf16_t)dQt[n][e];
            *(bf16x4*)(oq + 16 * n) = a;
        }
    }
}

bool short_enabled() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("AVAMD_ATTN_SHORT");
        v = (e && e[0] == '0') ? 0 : 1;
    }
    return v != 0;
}

bool al8(const void* ptr, long long bs, long long rs) { return ((uintptr_t)ptr % 8 == 0) && (bs % 4 == 0) && (rs % 4 == 0); }

bool v2_enabled() {                                             // AVAMD_ATTN_V2=0: the kernels with register staging (older path, kept for A/B)
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("AVAMD_ATTN_V2");
        v = (e && e[0] == '0') ? 0 : 1;
    }
    return v != 0;
}

template <int NKP, int DROP>
int launch_fwd_short2(const AttnP& p, hipStream_t st) {
    const int lds = 2 * NKP * 32 * 64 * 2;
    static bool done = false;
    if (!done) {
        if (hipFuncSetAttribute((const void*)attn_fwd_short_kernel<NKP, DROP>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess ||
            hipFuncSetAttribute((const void*)attn_fwd_short2_kernel<NKP, DROP>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) {
            av_set_error("av_attention_fwd: cannot raise dynamic LDS to %d", lds);
            return AV_ERR_LAUNCH;
        }
        done = true;
    }
    // in-kernel Philox (DROP 1) spills in the front-loaded form (65 vs 55 us): it keeps the older kernel
    if (v2_enabled() && DROP != 1) hipLaunchKernelGGL((attn_fwd_short2_kernel<NKP, DROP>), dim3((unsigned)p.H, (unsigned)p.B), dim3(NT), lds, st, p);
    else hipLaunchKernelGGL((attn_fwd_short_kernel<NKP, DROP>), dim3((unsigned)p.H, (unsigned)p.B), dim3(NT), lds, st, p);
    AV_LAUNCH_CHECK();
    return AV_OK;
}
template <int NKP>
int launch_fwd_short(const AttnP& p, hipStream_t st) {
    return p.drop_p > 0.f ? (p.dmask ? launch_fwd_short2<NKP, 2>(p, st) : launch_fwd_short2<NKP, 1>(p, st)) : launch_fwd_short2<NKP, 0>(p, st);
}

template <int DROP>
int launch_bwd_short(const BwdP& p, int R, int lds, hipStream_t st) {
    static bool done = false;
    if (!done) {
        const int mx = 2 * 256 * 64 * 2 + 2 * 256 * 4 + 16 * 64 * 8;
        if (hipFuncSetAttribute((const void*)attn_bwd_short_kernel<DROP>, hipFuncAttributeMaxDynamicSharedMemorySize, mx) != hipSuccess) {
            av_set_error("av_attention_bwd: cannot raise dynamic LDS");
            return AV_ERR_LAUNCH;
        }
        done = true;
    }
    hipLaunchKernelGGL(attn_bwd_short_kernel<DROP>, dim3((unsigned)p.H, (unsigned)p.B), dim3(NT), lds, st, p, R);
    AV_LAUNCH_CHECK();
    return AV_OK;
}

}  // namespace

#ifdef AV_ATTN_STAMPS
extern "C" int av_attn_stamps_read(unsigned long long* host_out) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_attn_stamps), sizeof(unsigned long long) * 4096 * 4) == hipSuccess ? 0 : 1;
}
#endif

template <bool DROP>
int launch_fwd_long(const AttnP& p, hipStream_t st) {
    const int lds = 2 * 128 * 64 * 2;
    static bool done = false;
    if (!done) {
        if (hipFuncSetAttribute((const void*)attn_fwd_long_kernel<DROP>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) {
            av_set_error("av_attention_fwd: cannot raise dynamic LDS to %d", lds);
            return AV_ERR_LAUNCH;
        }
        done = true;
    }
    hipLaunchKernelGGL((attn_fwd_long_kernel<DROP>), dim3((unsigned)p.H, (unsigned)p.B, (unsigned)((p.Tq + 16 * NW - 1) / (16 * NW))), dim3(NT), lds, st, p);
    AV_LAUNCH_CHECK();
    return AV_OK;
}

int av_attention_short_fwd_try(const AttnP& p, int D, hipStream_t st) {
    if (D != 64 || !p.vec_ok || !short_enabled() || !al8(p.o, p.o_bs, p.o_rs) || p.B > 65535) return AV_SHORT_NOT_TAKEN;
    if (p.Tk > 256 || p.Tq > 256) {
        static const bool long_on = [] { const char* e = getenv("AVAMD_ATTN_LONG"); return !(e && e[0] == '0'); }();
        if (!long_on || (p.Tq + 16 * NW - 1) / (16 * NW) > 65535) return AV_SHORT_NOT_TAKEN;
        return p.drop_p > 0.f ? launch_fwd_long<true>(p, st) : launch_fwd_long<false>(p, st);
    }
    switch ((p.Tk + 31) / 32) {
        case 1: return launch_fwd_short<1>(p, st);
        case 2: return launch_fwd_short<2>(p, st);
        case 3: return launch_fwd_short<3>(p, st);
        case 4: return launch_fwd_short<4>(p, st);
        case 5: return launch_fwd_short<5>(p, st);
        case 6: return launch_fwd_short<6>(p, st);
        case 7: return launch_fwd_short<7>(p, st);
        default: return launch_fwd_short<8>(p, st);
    }
}

template <bool DROP>
int launch_bwd_long(const BwdP& p, hipStream_t st) {
    const int lds = 2 * LC * 64 * 2 + 2 * LC * 4;
    const unsigned zk = (unsigned)((p.Tk + 16 * NW - 1) / (16 * NW)), zq = (unsigned)((p.Tq + 16 * NW - 1) / (16 * NW));
    hipLaunchKernelGGL(attn_bwd_long_kv_kernel<DROP>, dim3((unsigned)p.H, (unsigned)p.B, zk), dim3(NT), lds, st, p);
    hipLaunchKernelGGL(attn_bwd_long_q_kernel<DROP>, dim3((unsigned)p.H, (unsigned)p.B, zq), dim3(NT), 2 * LC * 64 * 2, st, p);
    AV_LAUNCH_CHECK();
    return AV_OK;
}

int av_attention_short_bwd_try(const BwdP& p, int D, hipStream_t st) {
    if (D != 64 || !p.vec_ok || !short_enabled() || p.B > 65535 || !al8(p.dq, p.dq_bs, p.dq_rs) ||
        !al8(p.dk, p.dk_bs, p.dk_rs) || !al8(p.dv, p.dv_bs, p.dv_rs))
        return AV_SHORT_NOT_TAKEN;
    if (p.Tk > 256 || p.Tq > 256) {
        static const bool long_on = [] { const char* e = getenv("AVAMD_ATTN_LONG"); return !(e && e[0] == '0'); }();
        if (!long_on || !p.o || !p.lse || (p.Tq + 16 * NW - 1) / (16 * NW) > 65535 || (p.Tk + 16 * NW - 1) / (16 * NW) > 65535) return AV_SHORT_NOT_TAKEN;
        return p.drop_p > 0.f ? launch_bwd_long<true>(p, st) : launch_bwd_long<false>(p, st);
    }
    const int tmax = p.Tq > p.Tk ? p.Tq : p.Tk;
    const int R = (tmax + 31) / 32 * 32;
    const int lds = 2 * R * 64 * 2 + 2 * R * 4 + (p.drop_p > 0.f && p.dmask ? ((p.Tq + 15) / 16) * 64 * 8 : 0);      // + the (b, h) keep bits
    return p.drop_p > 0.f ? (p.dmask ? launch_bwd_short<2>(p, R, lds, st) : launch_bwd_short<1>(p, R, lds, st)) : launch_bwd_short<0>(p, R, lds, st);
}

extern "C" int av_attention_dropmask(void* mask, int B, int H, int Tq, int Tk, float drop_p, unsigned long long drop_seed, unsigned int drop_stream,
                                     void* stream) {
    AV_CHECK(mask && B > 0 && H > 0 && Tq > 0 && Tk > 0 && Tq <= 256 && Tk <= 256 && drop_p > 0.f && drop_p < 1.f && (uintptr_t)mask % 32 == 0,
             "av_attention_dropmask: bad args (Tq=%d Tk=%d p=%f; T <= 256, mask 32-byte aligned)", Tq, Tk, drop_p);
    const long long n = (long long)B * H * ((Tq + 15) / 16) * 64;
    hipLaunchKernelGGL(attn_dropmask_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (uint2*)mask, B, H, Tq, Tk, drop_p,
                       drop_seed, drop_stream, n);
    AV_LAUNCH_CHECK();
    return AV_OK;
}
