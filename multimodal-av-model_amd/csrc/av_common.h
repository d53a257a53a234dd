// Shared device/host helpers for the gfx950 (MI355X, CDNA4) kernels of the audio-visual CTC path.
// Wavefront = 64 lanes; MFMA 16x16 tiles; all kernels enqueue on the caller's hipStream_t.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/av_hip.h"

// The 16-bit operand type of every kernel.  Default build (libavhip.so): bfloat16.  -DAV_HALF=1 builds the SAME sources with IEEE half
// operands (libavhip_f16.so: the reference's GPU arithmetic, torch.cuda.amp fp16 autocast, model/trainer.py:9,40,65 - BASELINE configs[4]
// "fp16 + fp32 master"): v_mfma_f32_16x16x32_f16 runs at the bf16 rate, accumulation stays fp32, and the dtype code AV_BF16 of the C-ABI means
// "this library's 16-bit type".  The name bf16_t is kept for both.
#ifdef AV_HALF
typedef _Float16 bf16_t;
typedef _Float16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 bf16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 bf16x2 __attribute__((ext_vector_type(2)));
#define AV_MFMA_F32_16X16X32_LP(A, B, C_, X, Y, Z) __builtin_amdgcn_mfma_f32_16x16x32_f16(A, B, C_, X, Y, Z)
typedef __fp16 av_gcc_half4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
__device__ __forceinline__ bf16x4 av_ds_read_tr16_b64_f16(__attribute__((address_space(3))) bf16x4* p) {
    return __builtin_bit_cast(bf16x4, __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) av_gcc_half4*)p));
}
#define AV_DS_READ_TR16_B64(P) av_ds_read_tr16_b64_f16(P)
#else
typedef __bf16 bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
#define AV_MFMA_F32_16X16X32_LP(A, B, C_, X, Y, Z) __builtin_amdgcn_mfma_f32_16x16x32_bf16(A, B, C_, X, Y, Z)
#define AV_DS_READ_TR16_B64(P) __builtin_amdgcn_ds_read_tr16_b64_v4bf16(P)
#endif
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define AV_WAVE 64

// ---- error plumbing (never abort; python turns a non-zero status into RuntimeError) ----
void av_set_error(const char* fmt, ...);
#define AV_CHECK(cond, ...)                         \
    do {                                            \
        if (!(cond)) {                              \
            av_set_error(__VA_ARGS__);              \
            return AV_ERR_ARG;                      \
        }                                           \
    } while (0)
#define AV_LAUNCH_CHECK()                                                        \
    do {                                                                         \
        hipError_t e__ = hipGetLastError();                                      \
        if (e__ != hipSuccess) {                                                 \
            av_set_error("%s:%d launch failed: %s", __FILE__, __LINE__, hipGetErrorString(e__)); \
            return AV_ERR_LAUNCH;                                                \
        }                                                                        \
    } while (0)

// ---- scalar conversions ----
template <typename T> __device__ __forceinline__ float to_f32(T v);
template <> __device__ __forceinline__ float to_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f32<bf16_t>(bf16_t v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float v) { return (bf16_t)v; }

// dtype-erased element load/store (dtype: AV_F32 / AV_BF16)
__device__ __forceinline__ float ld_any(const void* p, long long i, int dtype) {
    return dtype == AV_F32 ? ((const float*)p)[i] : (float)((const bf16_t*)p)[i];
}
__device__ __forceinline__ void st_any(void* p, long long i, int dtype, float v) {
    if (dtype == AV_F32) ((float*)p)[i] = v; else ((bf16_t*)p)[i] = (bf16_t)v;
}

// ---- wave / block reductions (64-lane wavefront) ----
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// exact (erf) GELU and its derivative — transformers/activations.py "gelu" 
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_grad_f(float x) {
    const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
    const float pdf = 0.39894228040143267794f * expf(-0.5f * x * x);
    return cdf + x * pdf;
}
// bf16 perf path (GEMM epilogues; the matrix pipe idles while they run, so every VALU cycle counts): erf(z), z = |x| / sqrt(2), as the
// odd polynomial z P(z^2) of degree 17 fitted on [0, 3] (|err| <= 2.8e-5, GELU abs err <= 5.8e-5, GELU' <= 1.4e-5: far below bf16
// resolution), 1 beyond - 9 FMAs and no transcendental (the Abramowitz-Stegun form cost an exp and a reciprocal on top of 5 FMAs)
__device__ __forceinline__ float erf_abs_poly(float z) {
    // explicit fused multiply-adds: the library is built with -ffp-contract=off (bit-exact BatchNorm / fp32-parity paths), under which
    // `p * z2 + c` is a multiply AND an add - twice the vector slots of this Horner chain while a GEMM's matrix pipes wait for its epilogue
    const float z2 = z * z;
    float p = 4.0719861260640755e-08f;
    p = __builtin_fmaf(p, z2, -1.9457509097264847e-06f);
    p = __builtin_fmaf(p, z2, 4.110950976610184e-05f);
    p = __builtin_fmaf(p, z2, -0.0005118074477650225f);
    p = __builtin_fmaf(p, z2, 0.004241328686475754f);
    p = __builtin_fmaf(p, z2, -0.025126988068223f);
    p = __builtin_fmaf(p, z2, 0.11113087832927704f);
    p = __builtin_fmaf(p, z2, -0.37536558508872986f);
    p = __builtin_fmaf(p, z2, 1.1282844543457031f);
    return z >= 3.0f ? 1.0f : p * z;
}
__device__ __forceinline__ float gelu_fast(float x) {
    const float e = erf_abs_poly(fabsf(x) * 0.70710678118654752440f);
    const float t = 0.5f * x;
    return __builtin_fmaf(fabsf(t), e, t);                 // 0.5 x (1 + sign(x) erf(|x| / sqrt 2))
}
__device__ __forceinline__ float gelu_grad_fast(float x) {
    const float e = erf_abs_poly(fabsf(x) * 0.70710678118654752440f);
    const float cdf = __builtin_fmaf(x < 0.f ? -0.5f : 0.5f, e, 0.5f);
    return __builtin_fmaf(x * 0.39894228040143267794f, __expf(-0.5f * x * x), cdf);
}
// gelu(x) and gelu'(x) from ONE erf evaluation (AV_ACT_GELU_GF)
__device__ __forceinline__ void gelu_both_fast(float x, float& gl, float& gp) {
    const float e = erf_abs_poly(fabsf(x) * 0.70710678118654752440f);
    const float t = 0.5f * x;
    gl = __builtin_fmaf(fabsf(t), e, t);
    gp = __builtin_fmaf(x * 0.39894228040143267794f, __expf(-0.5f * x * x), __builtin_fmaf(x < 0.f ? -0.5f : 0.5f, e, 0.5f));
}
// Packed-fp32 forms (v_pk_fma_f32 / v_pk_mul_f32: two elements per vector slot), the same operation sequence per element as the scalar
// functions above, so results are bit-identical to them.
typedef float av_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ av_f32x2 pk_fma(av_f32x2 a, av_f32x2 b, av_f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ av_f32x2 pk_splat(float c) { return av_f32x2{c, c}; }
__device__ __forceinline__ av_f32x2 erf_abs_poly2(av_f32x2 z) {
    const av_f32x2 z2 = z * z;
    av_f32x2 p = pk_splat(4.0719861260640755e-08f);
    p = pk_fma(p, z2, pk_splat(-1.9457509097264847e-06f));
    p = pk_fma(p, z2, pk_splat(4.110950976610184e-05f));
    p = pk_fma(p, z2, pk_splat(-0.0005118074477650225f));
    p = pk_fma(p, z2, pk_splat(0.004241328686475754f));
    p = pk_fma(p, z2, pk_splat(-0.025126988068223f));
    p = pk_fma(p, z2, pk_splat(0.11113087832927704f));
    p = pk_fma(p, z2, pk_splat(-0.37536558508872986f));
    p = pk_fma(p, z2, pk_splat(1.1282844543457031f));
    av_f32x2 e = p * z;
    e.x = z.x >= 3.0f ? 1.0f : e.x;
    e.y = z.y >= 3.0f ? 1.0f : e.y;
    return e;
}
__device__ __forceinline__ av_f32x2 pk_abs(av_f32x2 x) { return av_f32x2{fabsf(x.x), fabsf(x.y)}; }
__device__ __forceinline__ av_f32x2 gelu_fast2(av_f32x2 x) {
    const av_f32x2 e = erf_abs_poly2(pk_abs(x) * pk_splat(0.70710678118654752440f));
    const av_f32x2 t = pk_splat(0.5f) * x;
    return pk_fma(pk_abs(t), e, t);
}
__device__ __forceinline__ av_f32x2 gelu_grad_fast2(av_f32x2 x) {
    const av_f32x2 e = erf_abs_poly2(pk_abs(x) * pk_splat(0.70710678118654752440f));
    const av_f32x2 sg = av_f32x2{x.x < 0.f ? -0.5f : 0.5f, x.y < 0.f ? -0.5f : 0.5f};
    const av_f32x2 cdf = pk_fma(sg, e, pk_splat(0.5f));
    const av_f32x2 h = pk_splat(-0.5f) * x * x;
    return pk_fma(x * pk_splat(0.39894228040143267794f), av_f32x2{__expf(h.x), __expf(h.y)}, cdf);
}
__device__ __forceinline__ void gelu_both_fast2(av_f32x2 x, av_f32x2& gl, av_f32x2& gp) {
    const av_f32x2 e = erf_abs_poly2(pk_abs(x) * pk_splat(0.70710678118654752440f));
    const av_f32x2 t = pk_splat(0.5f) * x;
    gl = pk_fma(pk_abs(t), e, t);
    const av_f32x2 sg = av_f32x2{x.x < 0.f ? -0.5f : 0.5f, x.y < 0.f ? -0.5f : 0.5f};
    const av_f32x2 h = pk_splat(-0.5f) * x * x;
    gp = pk_fma(x * pk_splat(0.39894228040143267794f), av_f32x2{__expf(h.x), __expf(h.y)}, pk_fma(sg, e, pk_splat(0.5f)));
}
// The same functions over FOUR pairs at once, written coefficient by coefficient: each pair's Horner chain is 9 dependent packed FMAs, and
// evaluated pair after pair (as the per-pair functions above inline) the compiler emits four serial chains with a wait state between every two
// instructions - an epilogue that is bound by vector issue then runs at the latency of one chain.  Side by side the four chains fill each
// other's latency.  Per element the operations and their order are those of the per-pair functions: bit-identical results.
__device__ __forceinline__ void erf_abs_poly2x4(const av_f32x2 (&z)[4], av_f32x2 (&e)[4]) {
    av_f32x2 z2[4], p[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { z2[k] = z[k] * z[k]; p[k] = pk_splat(4.0719861260640755e-08f); }
    constexpr float cf[8] = {-1.9457509097264847e-06f, 4.110950976610184e-05f, -0.0005118074477650225f, 0.004241328686475754f,
                             -0.025126988068223f, 0.11113087832927704f, -0.37536558508872986f, 1.1282844543457031f};
#pragma unroll
    for (int c = 0; c < 8; ++c)
#pragma unroll
        for (int k = 0; k < 4; ++k) p[k] = pk_fma(p[k], z2[k], pk_splat(cf[c]));
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        e[k] = p[k] * z[k];
        e[k].x = z[k].x >= 3.0f ? 1.0f : e[k].x;
        e[k].y = z[k].y >= 3.0f ? 1.0f : e[k].y;
    }
}
__device__ __forceinline__ void gelu_fast2x4(const av_f32x2 (&x)[4], av_f32x2 (&g)[4]) {
    av_f32x2 z[4], e[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) z[k] = pk_abs(x[k]) * pk_splat(0.70710678118654752440f);
    erf_abs_poly2x4(z, e);
#pragma unroll
    for (int k = 0; k < 4; ++k) { const av_f32x2 t = pk_splat(0.5f) * x[k]; g[k] = pk_fma(pk_abs(t), e[k], t); }
}
__device__ __forceinline__ void gelu_both_fast2x4(const av_f32x2 (&x)[4], av_f32x2 (&gl)[4], av_f32x2 (&gp)[4]) {
    av_f32x2 z[4], e[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) z[k] = pk_abs(x[k]) * pk_splat(0.70710678118654752440f);
    erf_abs_poly2x4(z, e);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const av_f32x2 t = pk_splat(0.5f) * x[k];
        gl[k] = pk_fma(pk_abs(t), e[k], t);
        const av_f32x2 sg = av_f32x2{x[k].x < 0.f ? -0.5f : 0.5f, x[k].y < 0.f ? -0.5f : 0.5f};
        const av_f32x2 h = pk_splat(-0.5f) * x[k] * x[k];
        gp[k] = pk_fma(x[k] * pk_splat(0.39894228040143267794f), av_f32x2{__expf(h.x), __expf(h.y)}, pk_fma(sg, e[k], pk_splat(0.5f)));
    }
}
__device__ __forceinline__ float sigmoid_f(float x) { return 1.0f / (1.0f + expf(-x)); }

// ---- counter-based RNG for dropout: the mask of element `idx` of stream `stream` under `seed` is a pure function of (seed, stream, idx),
// so the backward pass regenerates exactly the forward's mask without storing it.  An aligned group of four elements draws 64 bits = FOUR
// 16-bit uniforms (the drop probability is resolved to 2^-16) from a keyed 32-bit integer mixer - two multiply-xorshift rounds with the
// "lowbias32" constants of C. Wellons' hash prospector (avalanche bias 0.17), a second key xor-ed in between them - evaluated at two inputs
// derived from the group index and a 32-bit key mixed from seed and stream.  Rounds 1-2 of this build used Philox2x32-10 here: its ten 32 x 32 -> 64-bit multiplies per group
// (v_mul_lo_u32 + v_mul_hi_u32, quarter-rate instructions) were a quarter of the FFN-up GEMM's epilogue time while its matrix pipes idled;
// the mixer needs four.  Dropout needs decorrelated, reproducible masks, not a cryptographic stream.
__device__ __forceinline__ unsigned drop_key(unsigned long long seed, unsigned stream) {
    unsigned h = (unsigned)seed * 0x9E3779B1u;
    h ^= (unsigned)(seed >> 32) * 0x85EBCA77u;
    h ^= (stream + 0x165667B1u) * 0xC2B2AE3Du;
    h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12; h *= 0x297A2D39u; h ^= h >> 15;
    return h;
}
// second key of a (seed, stream) pair, folded in BETWEEN the mixer's two rounds: with the first key alone (x = group ^ key, then a fixed
// bijection) every stream's mask would be an XOR-translated window of one 2^32-entry table, so the ~150 masks of a step would be permuted
// copies of overlapping windows; with a second, independently mixed key after the first multiply two streams are not translates of each other.
__device__ __forceinline__ unsigned drop_key2(unsigned key) {
    unsigned h = __builtin_rotateleft32(key, 16) * 0x9E3779B1u + 0x7F4A7C15u;
    h ^= h >> 13; h *= 0x5BD1E995u; h ^= h >> 15;
    return h;
}
__device__ __forceinline__ unsigned mix32(unsigned x, unsigned k2) {
    x ^= x >> 16; x *= 0x7FEB352Du; x ^= k2; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
    return x;
}
// a group index -> its two 32-bit draws (four 16-bit uniforms)
__device__ __forceinline__ void drop_draw(unsigned long long seed, unsigned stream, unsigned long long idx4, unsigned& o0, unsigned& o1) {
    const unsigned long long blk = idx4 >> 2;
    const unsigned c1 = (unsigned)(blk >> 32);                                     // non-zero only beyond 2^34 elements
    const unsigned key = drop_key(seed, stream), k2 = drop_key2(key);
    const unsigned x = (unsigned)blk ^ key ^ c1 ^ __builtin_rotateleft32(c1, 11) ^ __builtin_rotateleft32(c1, 23);
    o0 = mix32(x, k2); o1 = mix32(x + 0x9E3779B9u, k2);
}
// The keep test resolves p to thr = ceil(65536 p) sixteen-bit steps, so the kept fraction is (65536 - thr) / 65536, not 1 - p: the survivors
// are scaled by the inverse of THAT (an unbiased mask; with 1 / (1 - p) the expectation was off by up to 1.5e-5 relative).
__device__ __host__ __forceinline__ float drop_inv_keep(float p) {
    const float thr = __builtin_ceilf(p * 65536.0f);
    return 65536.0f / (65536.0f - thr);
}
// the four 16-bit uniforms (as floats in [0,1)) of the aligned group that starts at element idx4 (a multiple of 4)
__device__ __forceinline__ void drop_uniform4(unsigned long long seed, unsigned stream, unsigned long long idx4, float (&u)[4]) {
    unsigned o0, o1; drop_draw(seed, stream, idx4, o0, o1);
    u[0] = (float)(o0 & 0xFFFFu) * (1.0f / 65536.0f); u[1] = (float)(o0 >> 16) * (1.0f / 65536.0f);
    u[2] = (float)(o1 & 0xFFFFu) * (1.0f / 65536.0f); u[3] = (float)(o1 >> 16) * (1.0f / 65536.0f);
}
// uniform in [0,1) for element idx
__device__ __forceinline__ float drop_uniform(unsigned long long seed, unsigned stream, unsigned long long idx) {
    float u[4];
    drop_uniform4(seed, stream, idx & ~3ull, u);
    const int e = (int)(idx & 3);
    return e == 0 ? u[0] : e == 1 ? u[1] : e == 2 ? u[2] : u[3];
}
// multiplier of element idx: 0 (dropped) or 1/(1-p)
__device__ __forceinline__ float drop_mult(unsigned long long seed, unsigned stream, unsigned long long idx, float p, float inv_keep) {
    return drop_uniform(seed, stream, idx) >= p ? inv_keep : 0.f;
}
// out-of-line form for epilogues that are fully unrolled over accumulator registers (keeps the unrolled code small)
__device__ __noinline__ float drop_mult_call(unsigned long long seed, unsigned stream, unsigned long long idx, float p, float inv_keep) {
    return drop_mult(seed, stream, idx, p, inv_keep);
}
// keep bits (bit e = element idx4 + e is kept) of an aligned group of four under drop probability p (thr = ceil(65536 p))
__device__ __forceinline__ unsigned drop_keep4(unsigned long long seed, unsigned stream, unsigned long long idx4, unsigned thr) {
    unsigned o0, o1; drop_draw(seed, stream, idx4, o0, o1);
    return ((o0 & 0xFFFFu) >= thr ? 1u : 0u) | ((o0 >> 16) >= thr ? 2u : 0u) | ((o1 & 0xFFFFu) >= thr ? 4u : 0u) | ((o1 >> 16) >= thr ? 8u : 0u);
}
// 4 consecutive elements starting at a multiple of 4 (one generator call).  The keep test runs on the 16-bit integers: u = k / 65536 >= p
// <=> k >= ceil(65536 p) (both sides exact in fp32), which spares the int -> float conversion and the scaling of every element.
__device__ __forceinline__ void drop_mult4(unsigned long long seed, unsigned stream, unsigned long long idx4, float p, float inv_keep, float (&m)[4]) {
    unsigned o0, o1; drop_draw(seed, stream, idx4, o0, o1);
    const unsigned thr = (unsigned)__builtin_ceilf(p * 65536.0f);
    m[0] = (o0 & 0xFFFFu) >= thr ? inv_keep : 0.f; m[1] = (o0 >> 16) >= thr ? inv_keep : 0.f;
    m[2] = (o1 & 0xFFFFu) >= thr ? inv_keep : 0.f; m[3] = (o1 >> 16) >= thr ? inv_keep : 0.f;
}

static inline int av_cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }
