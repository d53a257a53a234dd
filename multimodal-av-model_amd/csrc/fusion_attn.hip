// Fused cross-attention block of CrossAttentionFusion (model/fusion_module.py:57-61 of the reference: nn.MultiheadAttention with
// audio queries and visual keys / values, need-weights path torch:functional.py:6206,6576-6606): the packed input projection and
// the attention core of one (batch item, head) in ONE workgroup, so that the projected Q / K / V never make an HBM round trip before
// the softmax and the kernel is MFMA- instead of launch- and HBM-bound (unfused, the 100 x 100 x 128 attention core has an
// intensity of 50 flop/B; with the in-projection fused the block reaches ~700 flop/B: SURVEY 8(d)).
//
//   q = a W_q,h^T + b_q,h      k = v W_k,h^T + b_k,h      vv = v W_v,h^T + b_v,h         (K = E = 512, head_dim 128)
//   o = softmax(scale * q k^T) vv                                                         (T <= 112 frames, no masks, no dropout)
//
// Phase 1 (projection, 8 wavefronts as 2 x 4): the three products are computed TRANSPOSED - M = head dimension (weights are the A
// operand), N = frames - so that an accumulator register quad is 4 consecutive d of one frame and leaves as one 8-byte LDS write
// into the row-major [frame][d] images.  K = 512 is streamed in tiles of 64 by LDS-DMA (global_load_lds_dwordx4; 128-B rows, 16-B
// chunk XOR-swizzled with row & 7 on the SOURCE address), first for Q (a + W_q,h: ring of four 32-KiB slots), then for K and V
// (v + W_k,h + W_v,h: ring of three 48-KiB slots) while the Q accumulators wait in registers.  A wavefront owns 2 d-tiles of each
// matrix x 4 frame tiles (96 accumulator registers).
// Phase 2 (attention, as attention_short.hip with head_dim 128): wavefront w owns query tile w; S^T = K Q^T puts the query on the
// lane and 4 keys in the accumulator = the B fragment of O^T = V^T P^T; V^T comes from the row-major image through
// ds_read_b64_tr_b16; the whole score row stays in registers (no online rescale).  Images use 256-B rows with the chunk swizzle
// s(row) = ((row & 3) << 2) | ((row >> 2) & 3), conflict-free for the row reads and the transposed reads alike.
// The projected q / k / vv and the row LSE are written out when the backward needs them.
#include <stdlib.h>

#include "av_common.h"

namespace {

constexpr int E = 512, HD = 128, TQ = 112, TK = 128, NT = 512, BK = 64;
constexpr int ROWS_X = 112, ROWS_W = 128;                   // frames of a / v per stage (T <= 112), rows of a head's weight slice
// Q pass: stage = a (112 rows x 128 B) + W_q,h (128 rows) = 30 KiB in a 32-KiB slot, ring of 4 (three K-tiles in flight);
// K/V pass: stage = v + W_k,h + W_v,h = 46 KiB in a 48-KiB slot, ring of 3 (two in flight)
constexpr int QS_SLOT = 32768, QS_N = 4, QS_W = ROWS_X * 128, QS_GRP = 30;
constexpr int KS_SLOT = 49152, KS_N = 3, KS_WK = ROWS_X * 128, KS_WV = KS_WK + ROWS_W * 128, KS_GRP = 46;
constexpr int IMG_Q = 0, IMG_K = TQ * 256, IMG_V = IMG_K + TK * 256;       // bf16 images, 256-B rows (94 208 B, reuse the rings)
constexpr int LDS_BYTES = KS_N * KS_SLOT;                    // 147 456 B >= QS_N * QS_SLOT = 131 072
constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef __attribute__((address_space(3))) bf16x4* lds_b4_t;

struct FxP {
    const bf16_t *a, *v, *w;          // a, v: [B, T, E]; w: packed in-projection [3E, E]
    const float* bias;                // [3E]
    bf16_t *q, *kv, *o;               // q [B, T, E] (optional), kv [B, T, 2, E] (optional), o [B, T, E]
    float* lse;                       // [B, H, T] (optional)
    int B, T, H;
    float scale;
};

__device__ __forceinline__ int sw256(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }

__global__ __launch_bounds__(NT, 2) void fusion_xattn_fwd_kernel(const FxP p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r = lane & 15, g = lane >> 4;
    const int h = blockIdx.x, b = blockIdx.y;
    const bf16_t* a_b = p.a + (long long)b * p.T * E;
    const bf16_t* v_b = p.v + (long long)b * p.T * E;
    const bf16_t* wq = p.w + (long long)(h * HD) * E;
    const bf16_t* wk = p.w + (long long)(E + h * HD) * E;
    const bf16_t* wv = p.w + (long long)(2 * E + h * HD) * E;

    // ---- phase 1: Q^T = W_q,h a^T, then K^T, V^T = W_k,h / W_v,h v^T -----------------------------------------------------------
    // The block is bound by the L2 -> LDS rate of its operand tiles, not by the MFMAs, so the tiles go through deep LDS-DMA rings with
    // counted waits (one raw s_barrier per K-tile, the queue never drains inside a pass).  Every wavefront issues the same number of
    // DMA instructions per stage (a wavefront whose last row group does not exist repeats its previous one): the vmcnt immediates
    // are compile-time constants.
    const int sub = lane >> 3, pch = lane & 7;
    auto dma = [&](const bf16_t* base, int row, int clampT, int kt, char* dst) {
        if (clampT && row > p.T - 1) row = p.T - 1;
        const bf16_t* src = base + (long long)row * E + kt * BK + ((pch ^ sub) << 3);
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)dst, 16, 0, 0);
    };
    auto issue_q = [&](int kt) {                             // 4 instructions per wavefront
        char* slot = smem + (kt % QS_N) * QS_SLOT;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int idx = __builtin_amdgcn_readfirstlane(w + 8 * i);
            if (idx >= QS_GRP) idx -= 8;
            if (idx < 14) dma(a_b, idx * 8 + sub, 1, kt, slot + idx * 1024);
            else dma(wq, (idx - 14) * 8 + sub, 0, kt, slot + idx * 1024);
        }
    };
    auto issue_kv = [&](int kt) {                            // 6 instructions per wavefront
        char* slot = smem + (kt % KS_N) * KS_SLOT;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            int idx = __builtin_amdgcn_readfirstlane(w + 8 * i);
            if (idx >= KS_GRP) idx -= 8;
            if (idx < 14) dma(v_b, idx * 8 + sub, 1, kt, slot + idx * 1024);
            else if (idx < 30) dma(wk, (idx - 14) * 8 + sub, 0, kt, slot + idx * 1024);
            else dma(wv, (idx - 30) * 8 + sub, 0, kt, slot + idx * 1024);
        }
    };
    const int wr = w >> 2, wn = w & 3;                       // frame tiles 4 wr .. 4 wr + 3; d-tiles 2 wn, 2 wn + 1 of each matrix
    f32x4 acc[3][2][4];                                      // [Q / K / V][d-tile][frame tile]
#pragma unroll
    for (int m = 0; m < 3; ++m)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[m][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int sw = r & 7;
    constexpr int NKT = E / BK;                              // 8 K-tiles
    // frame tile 7 (frames 112..127) does not exist: its fragment rows read the next region of the slot (finite weights), the results
    // are masked (keys) or never stored (queries)
    issue_q(0); issue_q(1); issue_q(2);
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
        if (kt + 2 < NKT) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");          // tile kt landed; kt+1, kt+2 may fly
        else if (kt + 1 < NKT) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                        // everyone's part of tile kt landed; slot of tile kt-1 is free
        asm volatile("" ::: "memory");
        if (kt + 3 < NKT) issue_q(kt + 3);
        const char* st = smem + (kt % QS_N) * QS_SLOT;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int choff = ((ks * 4 + g) ^ sw) << 4;
            bf16x8 fw[2], fa[4];
#pragma unroll
            for (int i = 0; i < 2; ++i) fw[i] = *(const bf16x8*)(st + QS_W + ((2 * wn + i) * 16 + r) * 128 + choff);
#pragma unroll
            for (int j = 0; j < 4; ++j) fa[j] = *(const bf16x8*)(st + ((4 * wr + j) * 16 + r) * 128 + choff);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[0][i][j] = AV_MFMA_F32_16X16X32_LP(fw[i], fa[j], acc[0][i][j], 0, 0, 0);
        }
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_s_barrier();                            // the Q ring is free
    asm volatile("" ::: "memory");
    issue_kv(0); issue_kv(1);
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
        if (kt + 1 < NKT) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");          // tile kt landed; kt+1 may fly
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (kt + 2 < NKT) issue_kv(kt + 2);
        const char* st = smem + (kt % KS_N) * KS_SLOT;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int choff = ((ks * 4 + g) ^ sw) << 4;
            bf16x8 fk[2], fvw[2], fv[4];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                fk[i] = *(const bf16x8*)(st + KS_WK + ((2 * wn + i) * 16 + r) * 128 + choff);
                fvw[i] = *(const bf16x8*)(st + KS_WV + ((2 * wn + i) * 16 + r) * 128 + choff);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) fv[j] = *(const bf16x8*)(st + ((4 * wr + j) * 16 + r) * 128 + choff);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc[1][i][j] = AV_MFMA_F32_16X16X32_LP(fk[i], fv[j], acc[1][i][j], 0, 0, 0);
                    acc[2][i][j] = AV_MFMA_F32_16X16X32_LP(fvw[i], fv[j], acc[2][i][j], 0, 0, 0);
                }
        }
    }
    __syncthreads();                                         // the rings are free for the images

    // ---- accumulators (+ bias) -> bf16 row-major images [frame][d]: lane (r = frame, g) holds d = 16 mt + 4 g .. + 3 ---------------
#pragma unroll
    for (int m = 0; m < 3; ++m)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int d0 = (2 * wn + i) * 16 + 4 * g;
            const f32x4 bv = *(const f32x4*)(p.bias + m * E + h * HD + d0);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int row = (4 * wr + j) * 16 + r;
                if (m == 0 && row >= TQ) continue;           // the Q image has 112 rows
                bf16x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (bf16_t)(acc[m][i][j][e] + bv[e]);
                char* img = smem + (m == 0 ? IMG_Q : m == 1 ? IMG_K : IMG_V);
                *(bf16x4*)(img + row * 256 + (((d0 >> 3) ^ sw256(row)) << 4) + 8 * ((d0 >> 2) & 1)) = o;
            }
        }
    __syncthreads();

    // ---- projected q / k / vv for the backward (16-B chunks, 256 contiguous bytes per frame and matrix) -------------------------------
    if (p.q) {
        for (int c = tid; c < p.T * 16; c += NT) {
            const int row = c >> 4, ch = c & 15;
            *(uint4*)(p.q + ((long long)b * p.T + row) * E + h * HD + ch * 8) = *(const uint4*)(smem + IMG_Q + row * 256 + ((ch ^ sw256(row)) << 4));
        }
    }
    if (p.kv) {
        for (int c = tid; c < p.T * 32; c += NT) {
            const int row = c >> 5, m = (c >> 4) & 1, ch = c & 15;
            *(uint4*)(p.kv + (((long long)b * p.T + row) * 2 + m) * E + h * HD + ch * 8) =
                *(const uint4*)(smem + (m ? IMG_V : IMG_K) + row * 256 + ((ch ^ sw256(row)) << 4));
        }
    }

    // ---- phase 2: attention of query tile w (wave-uniform; EXEC stays all ones for the transposed reads) ----------------------------
    if (w < TQ / 16 && w * 16 < p.T) {
        const int qrow = w * 16 + r;
        bf16x8 qf[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qf[ks] = *(const bf16x8*)(smem + IMG_Q + qrow * 256 + (((4 * ks + g) ^ sw256(qrow)) << 4));
        const float c = p.scale * LOG2E;
        f32x4 S[8];
        float mx = -INFINITY;
#pragma unroll
        for (int t = 0; t < 8; ++t) {                          // S^T tile t: rows = keys 16 t + 4 g + e, column = my query
            const int krow = 16 * t + r;
            f32x4 s4 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
                s4 = AV_MFMA_F32_16X16X32_LP(*(const bf16x8*)(smem + IMG_K + krow * 256 + (((4 * ks + g) ^ sw256(krow)) << 4)), qf[ks], s4, 0, 0, 0);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                s4[e] = (16 * t + 4 * g + e) < p.T ? s4[e] * c : -INFINITY;
                mx = fmaxf(mx, s4[e]);
            }
            S[t] = s4;
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float sum = 0.f;
#pragma unroll
        for (int t = 0; t < 8; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float pv = __builtin_amdgcn_exp2f(S[t][e] - mx);
                sum += pv;
                S[t][e] = pv;
            }
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        f32x4 O[8];
#pragma unroll
        for (int n = 0; n < 8; ++n) O[n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int tp = 0; tp < 4; ++tp) {                       // O^T[d][q] += V^T[d][32 keys] P^T[32 keys][q]
            bf16x8 pf;
#pragma unroll
            for (int e = 0; e < 4; ++e) { pf[e] = (bf16_t)S[2 * tp][e]; pf[4 + e] = (bf16_t)S[2 * tp + 1][e]; }
            // transposed reads of keys 32 tp + 4 g' + (0..3) and + 16: lane 4 q + pp of a 16-lane group addresses key row q, d columns 4 pp..
            const int row_lo = 32 * tp + 4 * g + ((lane >> 2) & 3), row_hi = row_lo + 16, pp = lane & 3;
#pragma unroll
            for (int n = 0; n < 8; ++n) {
                const int ch = 2 * n + (pp >> 1);
                const bf16x4 lo = AV_DS_READ_TR16_B64((lds_b4_t)(smem + IMG_V + row_lo * 256 + ((ch ^ sw256(row_lo)) << 4) + 8 * (pp & 1)));
                const bf16x4 hi = AV_DS_READ_TR16_B64((lds_b4_t)(smem + IMG_V + row_hi * 256 + ((ch ^ sw256(row_hi)) << 4) + 8 * (pp & 1)));
                O[n] = AV_MFMA_F32_16X16X32_LP(__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7), pf, O[n], 0, 0, 0);
            }
        }
        if (qrow < p.T) {
            const float inv = 1.0f / sum;
            bf16_t* o = p.o + ((long long)b * p.T + qrow) * E + h * HD + 4 * g;
#pragma unroll
            for (int n = 0; n < 8; ++n) {
                bf16x4 ov;
#pragma unroll
                for (int e = 0; e < 4; ++e) ov[e] = (bf16_t)(O[n][e] * inv);
                *(bf16x4*)(o + 16 * n) = ov;
            }
            if (g == 0 && p.lse) p.lse[((long long)b * p.H + h) * p.T + qrow] = (mx + __builtin_amdgcn_logf(sum)) * LN2;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Two items per workgroup (the default forward).  The one-item kernel above is bound by the L2 -> LDS rate of its operand tiles
// (608 KB per workgroup, 63 % of it the head's weight slices, identical for every item: 22 % MFMA-busy).  Here a workgroup projects
// TWO batch items against each streamed weight tile: 416 KB per (item, head) instead of 608, no padded eighth frame tile, and one
// round of B / 2 x 4 workgroups on the 256 CUs at the benchmark size.
//   * wavefront w = (item w >> 2, d-tile pair w & 3): 2 d-tiles x 7 frame tiles of each of Q^T, K^T, V^T = 168 accumulator registers;
//   * Q pass: stage = a0 | a1 | W_q,h (44 KiB), ring of 3 (two K-tiles in flight); K/V pass: stage = v0 | v1 | W_k,h | W_v,h (60 KiB),
//     ring of 2 (the next K-tile is requested as soon as the barrier has freed its slot);
//   * images: Q and K of both items (112 KiB) -> every attention wavefront takes the Q fragments of its query tiles into registers ->
//     the V accumulators overwrite the Q images (128 rows, rows >= 112 zeroed) -> S^T = K Q^T, softmax, O^T = V^T P^T as above;
//   * attention: wavefront w owns query tiles (w & 3) and (w & 3) + 4 of item w >> 2, processed together.
// Measured anatomy per workgroup at 2 x 64 items, T = 100 (tools/k17_stamps.py, wall-clock stamps): Q pass 9.4 us (352 KB), K/V pass 15.5 us
// (480 KB), images 2.9 - 6.2 us, attention 5.5 us = 37 us for 12 us of MFMA time.  Both passes run at 31 - 37 GB/s per CU of LDS-DMA: 44 / 60
// wave-instructions of 1 KiB per K-tile at ~55 clocks each - the same L2 -> LDS rate at which the 256 x 256 GEMM's main loop runs (64 KB per
// 1.6 - 1.8 us K-tile) and the vendor library's (46 GB/s per CU at 1.55 PF/s): the block is bound by that path, not by the matrix pipes.
// Tried on top and dropped (measured): touching the pair's a / v lines up front (Q pass + 3.5 us: the inputs are L2 / Infinity-Cache hits already).
// ---------------------------------------------------------------------------------------------------------------
// diagnostic build only (-DAV_K17_STAMPS, tools/k17_stamps.py): wall-clock stamps of every workgroup at entry / after the Q pass / after the
// K-V pass / before the attention / at exit, into a buffer nothing else reads.  No stamp exists in the product build.
#ifdef AV_K17_STAMPS
__device__ unsigned long long g_k17_stamps[1024 * 8];
#define K17_STAMP(SLOT) do { if (threadIdx.x == 0 && blockIdx.x < 1024) g_k17_stamps[blockIdx.x * 8 + (SLOT)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define K17_STAMP(SLOT) do { } while (0)
#endif
constexpr int P2_XROWS = 112 * 128;                          // one item's a / v tile: 112 rows x 128 B = 14 336 B (14 row groups)
constexpr int P2_QSLOT = 2 * P2_XROWS + ROWS_W * 128;        // 45 056 B
constexpr int P2_KSLOT = 2 * P2_XROWS + 2 * ROWS_W * 128;    // 61 440 B
constexpr int P2_QGRP = 44, P2_KGRP = 60;                    // 1-KiB row groups per stage
constexpr int P2_IMG_V = 0, P2_IMG_VSZ = 128 * 256;          // V images: 2 x 32 KiB (first the Q images: 2 x 28 KiB at the same place)
constexpr int P2_IMG_QSZ = TQ * 256;
constexpr int P2_IMG_K = 2 * P2_IMG_VSZ, P2_IMG_KSZ = TQ * 256;
constexpr int P2_LDS = 3 * P2_QSLOT;                         // 135 168 B >= 2 x 61 440 and >= images 65 536 + 57 344

__global__ __launch_bounds__(NT, 2) void fusion_xattn_fwd2_kernel(const FxP p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), r = lane & 15, g = lane >> 4;
    // XCD-aware job map (speed only): workgroup ids that are equal mod 8 share an XCD and its L2, so the four heads of an item pair get ids
    // k, k + 8, k + 16, k + 24: the pair's a / v tiles are fetched into that L2 once and hit three more times (with heads on the fastest grid
    // axis every head of a pair sat on a different XCD and the inputs crossed the fabric four times)
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int h = slot & 3, pair = (slot >> 2) * 8 + xcd;
    if (pair * 2 >= p.B) return;                             // padding of the last group of 8 pairs (workgroup-uniform)
    K17_STAMP(0);
    const int it = w >> 2, wn = w & 3;                       // my item of the pair, my d-tile pair
    const int b0 = 2 * pair, b1 = (b0 + 1 < p.B) ? b0 + 1 : b0;
    const bool has2 = b0 + 1 < p.B;
    const bf16_t* a0 = p.a + (long long)b0 * p.T * E;
    const bf16_t* a1 = p.a + (long long)b1 * p.T * E;
    const bf16_t* v0 = p.v + (long long)b0 * p.T * E;
    const bf16_t* v1 = p.v + (long long)b1 * p.T * E;
    const bf16_t* wq = p.w + (long long)(h * HD) * E;
    const bf16_t* wk = p.w + (long long)(E + h * HD) * E;
    const bf16_t* wv = p.w + (long long)(2 * E + h * HD) * E;

    const int sub = lane >> 3, pch = lane & 7;
    auto dma = [&](const bf16_t* base, int row, int clampT, int kt, char* dst) {
        if (clampT && row > p.T - 1) row = p.T - 1;
        const bf16_t* src = base + (long long)row * E + kt * BK + ((pch ^ sub) << 3);
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)dst, 16, 0, 0);
    };
    auto issue_q = [&](int kt) {                             // 6 instructions per wavefront
        char* slot = smem + (kt % 3) * P2_QSLOT;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            int idx = __builtin_amdgcn_readfirstlane(w + 8 * i);
            if (idx >= P2_QGRP) idx -= 8;
            if (idx < 14) dma(a0, idx * 8 + sub, 1, kt, slot + idx * 1024);
            else if (idx < 28) dma(a1, (idx - 14) * 8 + sub, 1, kt, slot + idx * 1024);
            else dma(wq, (idx - 28) * 8 + sub, 0, kt, slot + idx * 1024);
        }
    };
    auto issue_kv = [&](int kt) {                            // 8 instructions per wavefront
        char* slot = smem + (kt & 1) * P2_KSLOT;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            int idx = __builtin_amdgcn_readfirstlane(w + 8 * i);
            if (idx >= P2_KGRP) idx -= 8;
            if (idx < 14) dma(v0, idx * 8 + sub, 1, kt, slot + idx * 1024);
            else if (idx < 28) dma(v1, (idx - 14) * 8 + sub, 1, kt, slot + idx * 1024);
            else if (idx < 44) dma(wk, (idx - 28) * 8 + sub, 0, kt, slot + idx * 1024);
            else dma(wv, (idx - 44) * 8 + sub, 0, kt, slot + idx * 1024);
        }
    };
    f32x4 acc[3][2][7];                                      // [Q / K / V][d-tile][frame tile]
#pragma unroll
    for (int m = 0; m < 3; ++m)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 7; ++j) acc[m][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int sw = r & 7;
    constexpr int NKT = E / BK;                              // 8 K-tiles
    const int x_off = it * P2_XROWS + r * 128;               // my item's rows inside a stage
    const int w_off = 2 * P2_XROWS + (2 * wn * 16 + r) * 128;

    // ---- Q pass
    issue_q(0); issue_q(1);
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
        if (kt + 1 < NKT) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");          // tile kt landed; kt+1 may fly
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                        // everyone's part of tile kt landed; the slot of tile kt-1 is free
        asm volatile("" ::: "memory");
        if (kt + 2 < NKT) issue_q(kt + 2);
        const char* st = smem + (kt % 3) * P2_QSLOT;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int choff = ((ks * 4 + g) ^ sw) << 4;
            // 168 accumulator registers leave room for few fragments: the frame fragments stream through two registers sets (the next one is
            // requested before the current one's MFMAs; the order is pinned, or the scheduler hoists all seven loads and spills)
            bf16x8 fw[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) fw[i] = *(const bf16x8*)(st + w_off + i * 2048 + choff);
            bf16x8 cur = *(const bf16x8*)(st + x_off + choff);
#pragma unroll
            for (int j = 0; j < 7; ++j) {
                bf16x8 nxt = cur;
                if (j + 1 < 7) nxt = *(const bf16x8*)(st + x_off + (j + 1) * 2048 + choff);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 2; ++i) acc[0][i][j] = AV_MFMA_F32_16X16X32_LP(fw[i], cur, acc[0][i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                cur = nxt;
            }
        }
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_s_barrier();                            // the Q ring is free
    asm volatile("" ::: "memory");
    K17_STAMP(1);
    // ---- K / V pass
    issue_kv(0);
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // tile kt landed (my part)
        __builtin_amdgcn_s_barrier();                        // ... everyone's; the other slot (tile kt-1) is free
        asm volatile("" ::: "memory");
        if (kt + 1 < NKT) issue_kv(kt + 1);
        const char* st = smem + (kt & 1) * P2_KSLOT;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int choff = ((ks * 4 + g) ^ sw) << 4;
            bf16x8 fk[2], fvw[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                fk[i] = *(const bf16x8*)(st + w_off + i * 2048 + choff);
                fvw[i] = *(const bf16x8*)(st + w_off + ROWS_W * 128 + i * 2048 + choff);
            }
            bf16x8 cur = *(const bf16x8*)(st + x_off + choff);
#pragma unroll
            for (int j = 0; j < 7; ++j) {
                bf16x8 nxt = cur;
                if (j + 1 < 7) nxt = *(const bf16x8*)(st + x_off + (j + 1) * 2048 + choff);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    acc[1][i][j] = AV_MFMA_F32_16X16X32_LP(fk[i], cur, acc[1][i][j], 0, 0, 0);
                    acc[2][i][j] = AV_MFMA_F32_16X16X32_LP(fvw[i], cur, acc[2][i][j], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                cur = nxt;
            }
        }
    }
    __syncthreads();                                         // the rings are free for the images
    K17_STAMP(2);

    // ---- Q, K accumulators (+ bias) -> bf16 row-major images [frame][d] of my item; the V accumulators wait in registers
    auto dump = [&](int m, char* img) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int d0 = (2 * wn + i) * 16 + 4 * g;
            const f32x4 bv = *(const f32x4*)(p.bias + m * E + h * HD + d0);
#pragma unroll
            for (int j = 0; j < 7; ++j) {
                const int row = j * 16 + r;
                bf16x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (bf16_t)(acc[m][i][j][e] + bv[e]);
                *(bf16x4*)(img + row * 256 + (((d0 >> 3) ^ sw256(row)) << 4) + 8 * ((d0 >> 2) & 1)) = o;
            }
        }
    };
    char* imgQ = smem + P2_IMG_V + it * P2_IMG_QSZ;          // (the V images take this region over below, at a 32-KiB pitch)
    char* imgK = smem + P2_IMG_K + it * P2_IMG_KSZ;
    char* imgV = smem + P2_IMG_V + it * P2_IMG_VSZ;
    dump(0, imgQ);
    dump(1, imgK);
    __syncthreads();
    // my query tiles' Q fragments -> registers; projected q / k for the backward
    const int qt0 = wn, qt1 = wn + 4;
    const bool t0 = qt0 * 16 < p.T, t1 = qt1 < 7 && qt1 * 16 < p.T;                 // wave-uniform
    bf16x8 qf[2][4];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int qrow = (u ? (t1 ? qt1 : qt0) : qt0) * 16 + r;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qf[u][ks] = *(const bf16x8*)(imgQ + qrow * 256 + (((4 * ks + g) ^ sw256(qrow)) << 4));
    }
    if (p.q) {
        for (int c = tid; c < 2 * p.T * 16; c += NT) {
            const int im = c / (p.T * 16), cc = c - im * p.T * 16, row = cc >> 4, ch = cc & 15;
            if (im == 1 && !has2) continue;
            *(uint4*)(p.q + ((long long)(im ? b1 : b0) * p.T + row) * E + h * HD + ch * 8) =
                *(const uint4*)(smem + P2_IMG_V + im * P2_IMG_QSZ + row * 256 + ((ch ^ sw256(row)) << 4));
        }
    }
    __syncthreads();                                         // every Q fragment / copy has been read: the V images may take the region
    dump(2, imgV);
    for (int c = tid; c < 2 * 16 * 16; c += NT) {            // rows 112..127 of both V images: zero (P is exactly 0 there, the image must be finite)
        const int im = c >> 8, cc = c & 255;
        *(uint4*)(smem + P2_IMG_V + im * P2_IMG_VSZ + TQ * 256 + cc * 16) = make_uint4(0, 0, 0, 0);
    }
    __syncthreads();

    // ---- attention of my query tiles (wave-uniform control flow; EXEC stays all ones for the transposed reads).  The two tiles of a wavefront
    // are processed TOGETHER (two independent dependency chains S -> max -> exp -> sum -> P -> O: with 2 wavefronts per SIMD one chain alone
    // leaves the pipes waiting on it)
    K17_STAMP(3);
    const int bi = it ? b1 : b0;
    const float c = p.scale * LOG2E;
    const bool do0 = t0 && !(it == 1 && !has2), do1 = t1 && !(it == 1 && !has2);
    if (do0) {
        const int nu = do1 ? 2 : 1;                          // wave-uniform
        f32x4 S[2][7];
        float mx[2] = {-INFINITY, -INFINITY};
#pragma unroll
        for (int t = 0; t < 7; ++t) {                          // S^T tile t: rows = keys 16 t + 4 g + e, column = my query
            const int krow = 16 * t + r;
            bf16x8 kf[4];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) kf[ks] = *(const bf16x8*)(imgK + krow * 256 + (((4 * ks + g) ^ sw256(krow)) << 4));
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                f32x4 s4 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) s4 = AV_MFMA_F32_16X16X32_LP(kf[ks], qf[u][ks], s4, 0, 0, 0);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    s4[e] = (16 * t + 4 * g + e) < p.T ? s4[e] * c : -INFINITY;
                    mx[u] = fmaxf(mx[u], s4[e]);
                }
                S[u][t] = s4;
            }
        }
        float sum[2] = {0.f, 0.f};
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            mx[u] = fmaxf(mx[u], __shfl_xor(mx[u], 16, 64));
            mx[u] = fmaxf(mx[u], __shfl_xor(mx[u], 32, 64));
#pragma unroll
            for (int t = 0; t < 7; ++t)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float pv = __builtin_amdgcn_exp2f(S[u][t][e] - mx[u]);
                    sum[u] += pv;
                    S[u][t][e] = pv;
                }
            sum[u] += __shfl_xor(sum[u], 16, 64);
            sum[u] += __shfl_xor(sum[u], 32, 64);
        }
        f32x4 O[2][8];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int n = 0; n < 8; ++n) O[u][n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int tp = 0; tp < 4; ++tp) {                       // O^T[d][q] += V^T[d][32 keys] P^T[32 keys][q]; keys 112..127: P = 0, V = 0
            bf16x8 pf[2];
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    pf[u][e] = (bf16_t)S[u][2 * tp][e];
                    pf[u][4 + e] = (2 * tp + 1 < 7) ? (bf16_t)S[u][2 * tp + 1 < 7 ? 2 * tp + 1 : 6][e] : (bf16_t)0.f;
                }
            const int row_lo = 32 * tp + 4 * g + ((lane >> 2) & 3), row_hi = row_lo + 16, pp = lane & 3;
#pragma unroll
            for (int n = 0; n < 8; ++n) {
                const int ch = 2 * n + (pp >> 1);
                const bf16x4 lo = AV_DS_READ_TR16_B64((lds_b4_t)(imgV + row_lo * 256 + ((ch ^ sw256(row_lo)) << 4) + 8 * (pp & 1)));
                const bf16x4 hi = AV_DS_READ_TR16_B64((lds_b4_t)(imgV + row_hi * 256 + ((ch ^ sw256(row_hi)) << 4) + 8 * (pp & 1)));
                const bf16x8 vt = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                O[0][n] = AV_MFMA_F32_16X16X32_LP(vt, pf[0], O[0][n], 0, 0, 0);
                O[1][n] = AV_MFMA_F32_16X16X32_LP(vt, pf[1], O[1][n], 0, 0, 0);
            }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (u >= nu) continue;
            const int qrow = (u ? qt1 : qt0) * 16 + r;
            if (qrow < p.T) {
                const float inv = 1.0f / sum[u];
                bf16_t* o = p.o + ((long long)bi * p.T + qrow) * E + h * HD + 4 * g;
#pragma unroll
                for (int n = 0; n < 8; ++n) {
                    bf16x4 ov;
#pragma unroll
                    for (int e = 0; e < 4; ++e) ov[e] = (bf16_t)(O[u][n][e] * inv);
                    *(bf16x4*)(o + 16 * n) = ov;
                }
                if (g == 0 && p.lse) p.lse[((long long)bi * p.H + h) * p.T + qrow] = (mx[u] + __builtin_amdgcn_logf(sum[u])) * LN2;
            }
        }
    }
    // ---- projected k / vv for the backward: copied out last (the images are read-only by now; the stores drain behind the kernel's tail)
    if (p.kv) {
        for (int c2 = tid; c2 < 2 * p.T * 32; c2 += NT) {
            const int im = c2 / (p.T * 32), cc = c2 - im * p.T * 32, row = cc >> 5, m = (cc >> 4) & 1, ch = cc & 15;
            if (im == 1 && !has2) continue;
            const char* src = m ? smem + P2_IMG_V + im * P2_IMG_VSZ : smem + P2_IMG_K + im * P2_IMG_KSZ;
            *(uint4*)(p.kv + (((long long)(im ? b1 : b0) * p.T + row) * 2 + m) * E + h * HD + ch * 8) = *(const uint4*)(src + row * 256 + ((ch ^ sw256(row)) << 4));
        }
    }
#ifdef AV_K17_STAMPS
    __syncthreads();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    K17_STAMP(4);
}

// ---------------------------------------------------------------------------------------------------------------
// Backward of the attention core above (no masks, no dropout; T <= 112, head_dim 128): one workgroup per (item, head), every operand of
// the sequence staged ONCE into LDS images (Q, K, V, dO: 128 KiB, 256-B rows, chunk swizzle sw256), delta = rowsum(dO o O) computed while
// dO is staged.  Then, without any further barrier (the images are read-only):
//   phase A - wavefront w owns 16 keys:    S = Q K^T and dP = dO V^T have the key on the lane and 4 queries in the registers = the B
//             fragment of dV^T[d][key] += dO^T[d][32 q] P[32 q][key] and dK^T[d][key] += Q^T[d][32 q] dS[32 q][key]
//             (dO^T, Q^T through ds_read_b64_tr_b16 from the row-major images);
//   phase B - wavefront w owns 16 queries: S^T = K Q^T, dP^T = V dO^T put the query on the lane; dQ^T[d][q] += K^T[d][32 keys] dS^T.
// P is recomputed from the saved row LSE.  No atomics, no cross-workgroup accumulation: bitwise reproducible.
// ---------------------------------------------------------------------------------------------------------------
constexpr int BW_R = 128;                                    // image rows (frames padded to 128)
constexpr int BW_IMG = BW_R * 256;                           // 32 KiB per image
constexpr int BW_LDS = 4 * BW_IMG + 2 * BW_R * 4;            // Q, K, V, dO images + log2-LSE + delta

struct FxB {
    const bf16_t *q, *kv, *o, *dout;  // q, o, dout [B, T, E]; kv [B, T, 2, E]
    const float* lse;                 // [B, H, T]
    bf16_t *dq, *dkv;                 // dq [B, T, E], dkv [B, T, 2, E]
    int B, T, H;
    float scale;
};

__device__ __forceinline__ bf16x8 img_row(const char* img, int row, int ks, int g) {        // 8 consecutive d of row `row`: d = 32 ks + 8 g ..
    return *(const bf16x8*)(img + row * 256 + (((4 * ks + g) ^ sw256(row)) << 4));
}
// A fragment [16 d = d-tile n][k = 32 rows r0 .. r0 + 31] of the transposed image, k-slot order of pack8 (rows r0 + 4 g + j, r0 + 16 + 4 g + j)
__device__ __forceinline__ bf16x8 img_tr(const char* img, int r0, int n, int lane) {
    const int row_lo = r0 + 4 * (lane >> 4) + ((lane >> 2) & 3), row_hi = row_lo + 16, pp = lane & 3, ch = 2 * n + (pp >> 1);
    const bf16x4 lo = AV_DS_READ_TR16_B64((lds_b4_t)(img + row_lo * 256 + ((ch ^ sw256(row_lo)) << 4) + 8 * (pp & 1)));
    const bf16x4 hi = AV_DS_READ_TR16_B64((lds_b4_t)(img + row_hi * 256 + ((ch ^ sw256(row_hi)) << 4) + 8 * (pp & 1)));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}
__device__ __forceinline__ bf16x8 pack8f(const f32x4& a, const f32x4& b) {
    bf16x8 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[e] = (bf16_t)a[e]; v[4 + e] = (bf16_t)b[e]; }
    return v;
}

__global__ __launch_bounds__(NT, 2) void fusion_xattn_bwd_kernel(const FxB p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Qs = smem; char* Ks = smem + BW_IMG; char* Vs = smem + 2 * BW_IMG; char* Ds = smem + 3 * BW_IMG;
    float* lse_s = (float*)(smem + 4 * BW_IMG);              // log2-domain LSE per query
    float* del_s = lse_s + BW_R;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r = lane & 15, g = lane >> 4;
    const int h = blockIdx.x, b = blockIdx.y, T = p.T;
    const bf16_t* Q = p.q + (long long)b * T * E + h * HD;
    const bf16_t* KV = p.kv + (long long)b * T * 2 * E + h * HD;
    const bf16_t* O = p.o + (long long)b * T * E + h * HD;
    const bf16_t* DO = p.dout + (long long)b * T * E + h * HD;
    // ---- stage: 16-B chunks, 16 per row; rows >= T are zero.  delta needs O only here (8 lanes x 2 chunks cover a row: shuffle sum)
    for (int c0 = tid; c0 < BW_R * 16; c0 += NT) {
        const int row = c0 >> 4, ch = c0 & 15;
        const bool ok = row < T;
        const long long ro = (long long)(ok ? row : T - 1);
        const uint4 z = make_uint4(0, 0, 0, 0);
        const uint4 vq = *(const uint4*)(Q + ro * E + ch * 8), vk = *(const uint4*)(KV + ro * 2 * E + ch * 8);
        const uint4 vv = *(const uint4*)(KV + ro * 2 * E + E + ch * 8), vd = *(const uint4*)(DO + ro * E + ch * 8), vo = *(const uint4*)(O + ro * E + ch * 8);
        const int off = row * 256 + ((ch ^ sw256(row)) << 4);
        *(uint4*)(Qs + off) = ok ? vq : z; *(uint4*)(Ks + off) = ok ? vk : z; *(uint4*)(Vs + off) = ok ? vv : z; *(uint4*)(Ds + off) = ok ? vd : z;
        const bf16x8 d8 = __builtin_bit_cast(bf16x8, vd), o8 = __builtin_bit_cast(bf16x8, vo);
        float sd = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) sd += (float)d8[e] * (float)o8[e];
        sd = ok ? sd : 0.f;
        sd += __shfl_xor(sd, 1, 64); sd += __shfl_xor(sd, 2, 64); sd += __shfl_xor(sd, 4, 64); sd += __shfl_xor(sd, 8, 64);
        if (ch == 0) del_s[row] = sd;
    }
    for (int i = tid; i < BW_R; i += NT) lse_s[i] = i < T ? p.lse[((long long)b * p.H + h) * T + i] * LOG2E : 0.f;
    __syncthreads();

    const float c = p.scale * LOG2E;
    const int nt16 = (T + 15) >> 4;
    if (w < nt16) {                                          // wave-uniform: EXEC stays all ones for the transposed reads
        // ---- phase A: my 16 keys = tile w
        {
            const int krow = w * 16 + r;
            bf16x8 kf[4], vf[4];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) { kf[ks] = img_row(Ks, krow, ks, g); vf[ks] = img_row(Vs, krow, ks, g); }
            const bool kok = krow < T;
            f32x4 dVt[8], dKt[8];
#pragma unroll
            for (int n = 0; n < 8; ++n) { dVt[n] = f32x4{0.f, 0.f, 0.f, 0.f}; dKt[n] = f32x4{0.f, 0.f, 0.f, 0.f}; }
            for (int qp = 0; qp < (T + 31) >> 5; ++qp) {
                f32x4 Pt[2], St[2];
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
                    const int q0 = 32 * qp + 16 * hf;
                    f32x4 s4 = f32x4{0.f, 0.f, 0.f, 0.f}, dp = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) {
                        s4 = AV_MFMA_F32_16X16X32_LP(img_row(Qs, q0 + r, ks, g), kf[ks], s4, 0, 0, 0);     // S[query 4g+e][key r]
                        dp = AV_MFMA_F32_16X16X32_LP(img_row(Ds, q0 + r, ks, g), vf[ks], dp, 0, 0, 0);
                    }
                    const f32x4 l4 = *(const f32x4*)(lse_s + q0 + 4 * g), d4 = *(const f32x4*)(del_s + q0 + 4 * g);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float pv = (kok && q0 + 4 * g + e < T) ? __builtin_amdgcn_exp2f(s4[e] * c - l4[e]) : 0.f;
                        Pt[hf][e] = pv;
                        St[hf][e] = pv * (dp[e] - d4[e]) * p.scale;
                    }
                }
                const bf16x8 pf = pack8f(Pt[0], Pt[1]), sf = pack8f(St[0], St[1]);
#pragma unroll
                for (int n = 0; n < 8; ++n) {
                    dVt[n] = AV_MFMA_F32_16X16X32_LP(img_tr(Ds, 32 * qp, n, lane), pf, dVt[n], 0, 0, 0);   // dV^T[d][key]
                    dKt[n] = AV_MFMA_F32_16X16X32_LP(img_tr(Qs, 32 * qp, n, lane), sf, dKt[n], 0, 0, 0);   // dK^T[d][key]
                }
            }
            if (krow < T) {
                bf16_t* ok_ = p.dkv + ((long long)b * T + krow) * 2 * E + h * HD + 4 * g;
#pragma unroll
                for (int n = 0; n < 8; ++n) {
                    bf16x4 a4, c4;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { a4[e] = (bf16_t)dKt[n][e]; c4[e] = (bf16_t)dVt[n][e]; }
                    *(bf16x4*)(ok_ + 16 * n) = a4;
                    *(bf16x4*)(ok_ + E + 16 * n) = c4;
                }
            }
        }
        // ---- phase B: my 16 queries = tile w
        {
            const int qrow = w * 16 + r;
            bf16x8 qf[4], df[4];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) { qf[ks] = img_row(Qs, qrow, ks, g); df[ks] = img_row(Ds, qrow, ks, g); }
            const bool qok = qrow < T;
            const float lq = lse_s[qrow], dq_ = del_s[qrow];
            f32x4 dQt[8];
#pragma unroll
            for (int n = 0; n < 8; ++n) dQt[n] = f32x4{0.f, 0.f, 0.f, 0.f};
            for (int tp = 0; tp < (T + 31) >> 5; ++tp) {
                f32x4 St[2];
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
                    const int k0 = 32 * tp + 16 * hf;
                    f32x4 s4 = f32x4{0.f, 0.f, 0.f, 0.f}, dp = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) {
                        s4 = AV_MFMA_F32_16X16X32_LP(img_row(Ks, k0 + r, ks, g), qf[ks], s4, 0, 0, 0);     // S^T[key 4g+e][query r]
                        dp = AV_MFMA_F32_16X16X32_LP(img_row(Vs, k0 + r, ks, g), df[ks], dp, 0, 0, 0);
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float pv = (qok && k0 + 4 * g + e < T) ? __builtin_amdgcn_exp2f(s4[e] * c - lq) : 0.f;
                        St[hf][e] = pv * (dp[e] - dq_) * p.scale;
                    }
                }
                const bf16x8 sf = pack8f(St[0], St[1]);
#pragma unroll
                for (int n = 0; n < 8; ++n) dQt[n] = AV_MFMA_F32_16X16X32_LP(img_tr(Ks, 32 * tp, n, lane), sf, dQt[n], 0, 0, 0);   // dQ^T[d][q]
            }
            if (qok) {
                bf16_t* oq = p.dq + ((long long)b * T + qrow) * E + h * HD + 4 * g;
#pragma unroll
                for (int n = 0; n < 8; ++n) {
                    bf16x4 a4;
#pragma unroll
                    for (int e = 0; e < 4; ++e) a4[e] = (bf16_t)dQt[n][e];
                    *(bf16x4*)(oq + 16 * n) = a4;
                }
            }
        }
    }
}

}  // namespace

#ifdef AV_K17_STAMPS
extern "C" int av_k17_stamps_read(unsigned long long* host_out, int n_blocks) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_k17_stamps), (size_t)n_blocks * 8 * sizeof(unsigned long long)) == hipSuccess ? 0 : 1;
}
#endif

extern "C" int av_fusion_xattn_fwd(const void* a, const void* v, const void* w_in, const float* b_in, void* q_out, void* kv_out, void* o,
                                   float* lse, int B, int T, int E_, int H, float scale, void* stream) {
    AV_CHECK(a && v && w_in && b_in && o, "av_fusion_xattn_fwd: null pointer");
    AV_CHECK(E_ == E && H * HD == E, "av_fusion_xattn_fwd: built for embed_dim 512 = 4 heads x 128 (got E=%d H=%d)", E_, H);
    AV_CHECK(B > 0 && T > 0 && T <= TQ, "av_fusion_xattn_fwd: T=%d out of range (1..%d)", T, TQ);
    AV_CHECK(((uintptr_t)a | (uintptr_t)v | (uintptr_t)w_in | (uintptr_t)b_in | (uintptr_t)o | (uintptr_t)q_out | (uintptr_t)kv_out) % 16 == 0,
             "av_fusion_xattn_fwd: operands must be 16-byte aligned");
    static bool attr = false;
    if (!attr) {
        if (hipFuncSetAttribute((const void*)fusion_xattn_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) != hipSuccess) {
            av_set_error("av_fusion_xattn_fwd: cannot raise dynamic LDS to %d", LDS_BYTES);
            return AV_ERR_LAUNCH;
        }
        attr = true;
    }
    FxP p;
    p.a = (const bf16_t*)a; p.v = (const bf16_t*)v; p.w = (const bf16_t*)w_in; p.bias = b_in;
    p.q = (bf16_t*)q_out; p.kv = (bf16_t*)kv_out; p.o = (bf16_t*)o; p.lse = lse;
    p.B = B; p.T = T; p.H = H; p.scale = scale;
    static const int pair = [] { const char* e = getenv("AVAMD_XATTN_PAIR"); return e ? atoi(e) : 1; }();     // 0: one item per workgroup (A/B runs)
    if (pair && B >= 2) {
        static bool attr2 = false;
        if (!attr2) {
            if (hipFuncSetAttribute((const void*)fusion_xattn_fwd2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, P2_LDS) != hipSuccess) {
                av_set_error("av_fusion_xattn_fwd: cannot raise dynamic LDS to %d", P2_LDS);
                return AV_ERR_LAUNCH;
            }
            attr2 = true;
        }
        const int npairs = (B + 1) / 2, ngrp = (npairs + 7) / 8;
        hipLaunchKernelGGL(fusion_xattn_fwd2_kernel, dim3((unsigned)(32 * ngrp)), dim3(NT), P2_LDS, (hipStream_t)stream, p);
        AV_LAUNCH_CHECK();
        return AV_OK;
    }
    hipLaunchKernelGGL(fusion_xattn_fwd_kernel, dim3((unsigned)H, (unsigned)B), dim3(NT), LDS_BYTES, (hipStream_t)stream, p);
    AV_LAUNCH_CHECK();
    return AV_OK;
}

extern "C" int av_fusion_xattn_bwd(const void* q, const void* kv, const void* o, const void* dout, const float* lse, void* dq, void* dkv, int B, int T,
                                   int E_, int H, float scale, void* stream) {
    AV_CHECK(q && kv && o && dout && lse && dq && dkv, "av_fusion_xattn_bwd: null pointer");
    AV_CHECK(E_ == E && H * HD == E, "av_fusion_xattn_bwd: built for embed_dim 512 = 4 heads x 128 (got E=%d H=%d)", E_, H);
    AV_CHECK(B > 0 && T > 0 && T <= TQ, "av_fusion_xattn_bwd: T=%d out of range (1..%d)", T, TQ);
    AV_CHECK(((uintptr_t)q | (uintptr_t)kv | (uintptr_t)o | (uintptr_t)dout | (uintptr_t)dq | (uintptr_t)dkv) % 16 == 0, "av_fusion_xattn_bwd: operands must be 16-byte aligned");
    static bool attr = false;
    if (!attr) {
        if (hipFuncSetAttribute((const void*)fusion_xattn_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, BW_LDS) != hipSuccess) {
            av_set_error("av_fusion_xattn_bwd: cannot raise dynamic LDS to %d", BW_LDS);
            return AV_ERR_LAUNCH;
        }
        attr = true;
    }
    FxB p;
    p.q = (const bf16_t*)q; p.kv = (const bf16_t*)kv; p.o = (const bf16_t*)o; p.dout = (const bf16_t*)dout; p.lse = lse;
    p.dq = (bf16_t*)dq; p.dkv = (bf16_t*)dkv; p.B = B; p.T = T; p.H = H; p.scale = scale;
    hipLaunchKernelGGL(fusion_xattn_bwd_kernel, dim3((unsigned)H, (unsigned)B), dim3(NT), BW_LDS, (hipStream_t)stream, p);
    AV_LAUNCH_CHECK();
    return AV_OK;
}
