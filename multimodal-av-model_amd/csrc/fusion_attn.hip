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
                for (int j = 0; j < 4; ++j) acc[0][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[i], fa[j], acc[0][i][j], 0, 0, 0);
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
                    acc[1][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fk[i], fv[j], acc[1][i][j], 0, 0, 0);
                    acc[2][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fvw[i], fv[j], acc[2][i][j], 0, 0, 0);
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
                s4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8*)(smem + IMG_K + krow * 256 + (((4 * ks + g) ^ sw256(krow)) << 4)), qf[ks], s4, 0, 0, 0);
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
                const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4_t)(smem + IMG_V + row_lo * 256 + ((ch ^ sw256(row_lo)) << 4) + 8 * (pp & 1)));
                const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4_t)(smem + IMG_V + row_hi * 256 + ((ch ^ sw256(row_hi)) << 4) + 8 * (pp & 1)));
                O[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7), pf, O[n], 0, 0, 0);
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

}  // namespace

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
    hipLaunchKernelGGL(fusion_xattn_fwd_kernel, dim3((unsigned)H, (unsigned)B), dim3(NT), LDS_BYTES, (hipStream_t)stream, p);
    AV_LAUNCH_CHECK();
    return AV_OK;
}
