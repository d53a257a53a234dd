// Shared LDS staging / MFMA helpers of the attention kernels (forward: attention.hip, backward: attention_bwd.hip).
#pragma once
#include "av_common.h"

// launch parameter blocks (global scope: shared between attention.hip, attention_bwd.hip and attention_short.hip)
struct AttnP {
    const void *q, *k, *v;
    void* o;
    float* lse;
    const int* klen;
    int B, H, Tq, Tk;
    long long q_bs, q_rs, k_bs, k_rs, v_bs, v_rs, o_bs, o_rs;
    float scale;
    int vec_ok;
    float drop_p;                 // attention-probability dropout (hf:457): P*mask/(1-p) feeds PV, the softmax sum is undropped
    unsigned drop_stream;
    unsigned long long drop_seed;
    const unsigned long long* dmask;   // optional: precomputed keep bits (attn_dropmask_kernel, attention_short.hip) read instead of Philox
};

// Attention-probability dropout: element (b, h, q, k) uses Philox element index ((b H + h) Tq + q) * Tk4 + k with the row pitch
// Tk4 = Tk rounded up to 4, so four consecutive keys of one query always share ONE Philox block (drop_mult4) in every kernel.
struct BwdP {
    const bf16_t *q, *k, *v, *o, *dout;
    const float *lse, *delta;
    bf16_t *dq, *dk, *dv;
    const int* klen;
    int B, H, Tq, Tk;
    long long q_bs, q_rs, k_bs, k_rs, v_bs, v_rs, o_bs, o_rs, do_bs, do_rs, dq_bs, dq_rs, dk_bs, dk_rs, dv_bs, dv_rs;
    float scale;
    int vec_ok;
    float drop_p;
    unsigned drop_stream;
    unsigned long long drop_seed;
    const unsigned long long* dmask;   // optional: the same precomputed keep bits
};

// whole-sequence kernels for short sequences (attention_short.hip); AV_SHORT_NOT_TAKEN = shape not covered, caller goes on
#define AV_SHORT_NOT_TAKEN (-1000)
int av_attention_short_fwd_try(const AttnP& p, int D, hipStream_t st);
int av_attention_short_bwd_try(const BwdP& p, int D, hipStream_t st);

namespace {

template <typename T> struct ACfg;
template <> struct ACfg<float> { static constexpr int VEC = 4; };
template <> struct ACfg<bf16_t> { static constexpr int VEC = 8; };

template <typename T> __device__ __forceinline__ void zero16(T* dst) { *(uint4*)dst = make_uint4(0, 0, 0, 0); }

// acc += A[16 x K] * B[16 x K]^T, both operands row-major in LDS with K contiguous
template <typename T>
__device__ __forceinline__ void mma_rows(f32x4& acc, const T* a, int lda, const T* b, int ldb, int K, int lane);
template <>
__device__ __forceinline__ void mma_rows<bf16_t>(f32x4& acc, const bf16_t* a, int lda, const bf16_t* b, int ldb, int K, int lane) {
    const int r = lane & 15, g = lane >> 4;
    for (int k0 = 0; k0 < K; k0 += 32) {
        const bf16x8 av = *(const bf16x8*)(a + r * lda + k0 + 8 * g);
        const bf16x8 bv = *(const bf16x8*)(b + r * ldb + k0 + 8 * g);
        acc = AV_MFMA_F32_16X16X32_LP(av, bv, acc, 0, 0, 0);
    }
}
template <>
__device__ __forceinline__ void mma_rows<float>(f32x4& acc, const float* a, int lda, const float* b, int ldb, int K, int lane) {
    const int r = lane & 15, g = lane >> 4;
    for (int k0 = 0; k0 < K; k0 += 16) {
        const f32x4 av = *(const f32x4*)(a + r * lda + k0 + 4 * g);
        const f32x4 bv = *(const f32x4*)(b + r * ldb + k0 + 4 * g);
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[jj], bv[jj], acc, 0, 0, 0);
    }
}

// ---- fragment-level helpers: load the A fragments of a 16-row operand ONCE, reuse them against many B tiles ---------
template <typename T, int K> struct AFrag;
template <int K> struct AFrag<bf16_t, K> {
    bf16x8 f[K / 32];
    __device__ __forceinline__ void load(const bf16_t* a, int lda, int lane) {
        const int r = lane & 15, g = lane >> 4;
#pragma unroll
        for (int i = 0; i < K / 32; ++i) f[i] = *(const bf16x8*)(a + r * lda + i * 32 + 8 * g);
    }
    __device__ __forceinline__ void mma(f32x4& acc, const bf16_t* b, int ldb, int lane) const {
        const int r = lane & 15, g = lane >> 4;
#pragma unroll
        for (int i = 0; i < K / 32; ++i) {
            const bf16x8 bv = *(const bf16x8*)(b + r * ldb + i * 32 + 8 * g);
            acc = AV_MFMA_F32_16X16X32_LP(f[i], bv, acc, 0, 0, 0);
        }
    }
};
template <int K> struct AFrag<float, K> {
    f32x4 f[K / 16];
    __device__ __forceinline__ void load(const float* a, int lda, int lane) {
        const int r = lane & 15, g = lane >> 4;
#pragma unroll
        for (int i = 0; i < K / 16; ++i) f[i] = *(const f32x4*)(a + r * lda + i * 16 + 4 * g);
    }
    __device__ __forceinline__ void mma(f32x4& acc, const float* b, int ldb, int lane) const {
        const int r = lane & 15, g = lane >> 4;
#pragma unroll
        for (int i = 0; i < K / 16; ++i) {
            const f32x4 bv = *(const f32x4*)(b + r * ldb + i * 16 + 4 * g);
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(f[i][jj], bv[jj], acc, 0, 0, 0);
        }
    }
};

// stage `rows_tile` x DK elements (zero-filled outside [row_end) x [0,D)) into dst[row][ld]
template <typename T, int D, int DK>
__device__ __forceinline__ void stage_rows(T* dst, int ld, const T* src, long long rs, int row0, int row_end, bool vec, int tid) {
    constexpr int VEC = ACfg<T>::VEC;
    constexpr int CPR = DK / VEC;
    for (int c = tid; c < 64 * CPR; c += 256) {
        const int row = c / CPR, col = (c % CPR) * VEC;
        const int gr = row0 + row;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (gr < row_end && col < D) {
            const T* p = src + (long long)gr * rs + col;
            if (vec) v = *(const uint4*)p;
            else {
                __attribute__((aligned(16))) T tmp[VEC];
#pragma unroll
                for (int e = 0; e < VEC; ++e) tmp[e] = p[e];
                v = *(const uint4*)tmp;
            }
        }
        *(uint4*)(dst + row * ld + col) = v;
    }
}

// stage V rows [row0,row0+64) transposed: dst[d][key]
template <typename T, int D>
__device__ __forceinline__ void stage_vt(T* dst, int ld, const T* src, long long rs, int row0, int row_end, bool vec, int tid) {
    constexpr int VEC = ACfg<T>::VEC;
    constexpr int CPR = D / VEC;
    for (int c = tid; c < 64 * CPR; c += 256) {
        const int key = c / CPR, col = (c % CPR) * VEC;
        const int gr = row0 + key;
        __attribute__((aligned(16))) T tmp[VEC];
        if (gr < row_end) {
            const T* p = src + (long long)gr * rs + col;
            if (vec) *(uint4*)tmp = *(const uint4*)p;
            else {
#pragma unroll
                for (int e = 0; e < VEC; ++e) tmp[e] = p[e];
            }
        } else {
#pragma unroll
            for (int e = 0; e < VEC; ++e) tmp[e] = from_f32<T>(0.f);
        }
#pragma unroll
        for (int e = 0; e < VEC; ++e) dst[(col + e) * ld + key] = tmp[e];
    }
}


}  // namespace
