// 3x3 / stride 1 / pad 1 convolution, 64 -> 64 channels, NHWC bf16: the four convolutions of ResNet-18 layer1 in the lip
// encoder (model/encoder.py:44-57 of the reference; 2 x 3200 frames of 24 x 24 x 64 per step at configs[1]).
// With only 64 output channels the implicit-GEMM kernel of gemm_fast.hip re-stages the 128-pixel A tile for each of the 9
// taps (9 x the input through L2 -> LDS) and is bound by that traffic (~420 TFLOP/s).  This kernel is weights-stationary:
//   * a persistent workgroup keeps ALL 9 x 64 x 64 filter taps in LDS (72 KiB) for its whole life;
//   * the input of a 256-pixel tile is staged by LDS-DMA ONCE, as a window of 256 + 2 (W + 1) consecutive flattened pixels; tap
//     (ky, kx) of output pixel m reads window row m + ky * W + kx, so the 9 taps share one copy of the input (1.2 x instead of
//     9 x through L2 -> LDS) and a window is worth 288 MFMAs per wavefront between barriers; windows are double buffered;
//   * border taps (zero padding, image seams inside a window) are handled by zeroing the pixel fragment of the lanes whose
//     tap falls outside the image - the staged neighbour data is finite and never multiplied;
//   * MFMA operands are swapped (D^T = W P^T) so a lane holds 4 consecutive output channels of one pixel: results go through
//     the consumed window buffer (swizzled) and leave as full 128-B rows; train-mode BatchNorm partial sums (model/trainer.py:54)
//     are reduced from the fp32 accumulators.
#include "av_common.h"

namespace {

constexpr int CM = 256;                        // output pixels per tile (4 wavefronts x 64)
constexpr int NINSTR = 40;                     // LDS-DMA wave instructions per window (10 per wavefront): 320 rows >= CM + 2 (W + 1)
constexpr int MAXW = 31;
constexpr int W_BYTES = 9 * 64 * 128;          // filter taps  [tap][n][64 c]
constexpr int WIN_BYTES = NINSTR * 1024;       // one window   [320 rows][64 c]; the consumed window doubles as the output staging buffer
constexpr int ST_BYTES = 4 * 2 * 64 * 4;       // per-wave BatchNorm partials
constexpr int LDS_TOTAL = W_BYTES + 2 * WIN_BYTES + ST_BYTES;

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// workgroup barrier that orders LDS traffic only: __syncthreads() also drains vmcnt, i.e. it would wait for the LDS-DMA prefetch of
// the NEXT window at every barrier.  Global-memory ordering in this kernel is by the counted vmcnt in front of the first barrier.
__device__ __forceinline__ void wg_barrier() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0xC07F);          // lgkmcnt(0)
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// sum over the 16 lanes of a DPP row (row_shr 1, 2, 4, 8 with zero fill): the total lands in lane 15 of the row
__device__ __forceinline__ float row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x112, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x114, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x118, 0xf, 0xf, true));
    return v;
}

struct CP {
    const bf16_t* x; const bf16_t* w; bf16_t* y; float* stats;
    const float* in_scale; const float* in_shift; const float* in_slope;             // XF: per-channel affine + PReLU applied to the input
    int M, H, W, ntiles;
};

// XF: the input is the RAW output of the previous convolution; its BatchNorm-apply + PReLU (the mid-block activation of a
// BasicBlock, model/encoder.py:27-31) runs in place on the staged window - one pass over 40 KiB of LDS per 256-pixel tile instead
// of a 236 MB read + 236 MB write through HBM per convolution.  Same float32 operations as av_bn_act => the same bf16 values.
template <bool XF>
__global__ __launch_bounds__(256) void conv3x3_c64_kernel(const CP p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Wl = smem;
    char* Win = smem + W_BYTES;
    float* St = (float*)(smem + W_BYTES + 2 * WIN_BYTES);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int r = lane & 15, g = lane >> 4;
    const int G = gridDim.x;
    if ((int)blockIdx.x >= p.ntiles) return;

    // filter -> LDS once: global row n holds k = tap*64 + c; LDS row (tap, n) = 128 B, 16-B chunk XOR-swizzled with n & 7
    for (int i = tid; i < 9 * 64 * 8; i += 256) {
        const int row = i >> 3, ch = i & 7;
        const int tap = row >> 6, n = row & 63;
        *(uint4*)(Wl + row * 128 + ((ch ^ (n & 7)) << 4)) = *(const uint4*)(p.w + (long long)n * 576 + tap * 64 + ch * 8);
    }

    // window t of this workgroup = the input rows of tile first + t * G: flattened pixels p0 - (W+1) .. p0 + CM + W  (row of
    // output pixel m, tap (ky, kx) = m + ky * W + kx)
    const int T = (p.ntiles - (int)blockIdx.x + G - 1) / G;
    const int sub = lane >> 3, choff = ((lane & 7) ^ sub) << 3;
    float xs[8], xb[8], xl[8];                                                        // XF: parameters of MY channel chunk (fixed per thread:
    if constexpr (XF) {                                                               // chunk id = k * 256 + tid => slot tid & 7, row & 7 = (tid >> 3) & 7)
        const int cg = ((tid & 7) ^ ((tid >> 3) & 7)) * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) { xs[e] = p.in_scale[cg + e]; xb[e] = p.in_shift[cg + e]; xl[e] = p.in_slope ? p.in_slope[cg + e] : 1.f; }
    }
    // k-th of this wavefront's 10 LDS-DMA instructions of window t.  Window 0 goes out as one burst; window t+1 is issued ONE
    // instruction per MFMA step inside tile t's loop: a burst holds the issuing wavefront for hundreds of clocks at the CU's address
    // unit (measured on the GEMM: tools/v3_stamps.cpp), and issued behind tile t's output stores it also made the next tile's
    // vmcnt wait cover those stores.
    auto stage_k = [&](int t, int k) {
        const int tile = (int)blockIdx.x + t * G;
        const long long q0 = (long long)tile * CM - (p.W + 1);
        char* buf = Win + (t & 1) * WIN_BYTES;
        const int ins = k * 4 + w;
        long long q = q0 + ins * 8 + sub;
        q = q < 0 ? 0 : (q > p.M - 1 ? p.M - 1 : q);                                    // out-of-range rows are never used unmasked
        const bf16_t* src = p.x + q * 64 + choff;
        const unsigned off = __builtin_amdgcn_readfirstlane((unsigned)(ins * 1024));
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(buf + off), 16, 0, 0);
    };
#pragma unroll
    for (int k = 0; k < NINSTR / 4; ++k) stage_k(0, k);

    const float invW = 1.0f / (float)p.W, invH = 1.0f / (float)p.H;
    for (int t = 0; t < T; ++t) {
        const int tile = (int)blockIdx.x + t * G;
        const int p0 = tile * CM;
        f32x4 acc[4][4];                                                                 // [channel tile j][pixel tile i]
        unsigned vm[4];                                                                  // per pixel tile: bits 0-2 ky valid, 3-5 kx valid
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int px = p0 + w * 64 + i * 16 + r;
            int row_ = (int)((float)px * invW), xx = px - row_ * p.W;                    // px < 2^24: one correction step is enough
            if (xx < 0) { xx += p.W; --row_; } else if (xx >= p.W) { xx -= p.W; ++row_; }
            int img_ = (int)((float)row_ * invH), yy = row_ - img_ * p.H;
            if (yy < 0) yy += p.H; else if (yy >= p.H) yy -= p.H;
            unsigned m = 0;
            if (px < p.M) m = (yy > 0 ? 1u : 0u) | 2u | (yy < p.H - 1 ? 4u : 0u) | (xx > 0 ? 8u : 0u) | 16u | (xx < p.W - 1 ? 32u : 0u);
            vm[i] = m;
        }
        // my 10 LDS-DMA instructions of window t have landed - they were issued during tile t-1's MFMA loop; the output /
        // statistics stores of tile t-1 behind them may still be in flight - then everyone's
        if (t > 0) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");             // 8 output stores per wavefront are newer than the window (+ 1 statistics store in wavefronts 0, 1)
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        wg_barrier();
        const bool more = t + 1 < T;
        char* win = Win + (t & 1) * WIN_BYTES;
        if constexpr (XF) {                                                           // 320 rows x 8 chunks = 10 chunks per thread, in place
            uint4 tv[NINSTR / 4];
#pragma unroll
            for (int k = 0; k < NINSTR / 4; ++k) tv[k] = *(const uint4*)(win + (k * 256 + tid) * 16);
#pragma unroll
            for (int k = 0; k < NINSTR / 4; ++k) {
                const bf16x8 v = __builtin_bit_cast(bf16x8, tv[k]);
                bf16x8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float f = (float)v[e] * xs[e] + xb[e];
                    o[e] = (bf16_t)(f >= 0.f ? f : f * xl[e]);
                }
                *(uint4*)(win + (k * 256 + tid) * 16) = __builtin_bit_cast(uint4, o);
            }
            wg_barrier();
        }
        // 18 steps (ky, kx, k-half), fragments double buffered in registers: the LDS reads of step s+1 are issued before the 16
        // MFMAs of step s (with one wavefront per SIMD nothing else would hide the LDS latency)
        bf16x8 wfA[4], pfA[4], wfB[4], pfB[4];
#define C64_LOAD(WF, PF, STEP)                                                                                           \
    {                                                                                                                    \
        const int ky_ = (STEP) / 6, kx_ = ((STEP) / 2) % 3, ks_ = (STEP) & 1;                                            \
        const char* wt_ = Wl + (ky_ * 3 + kx_) * 64 * 128;                                                               \
        _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                                    \
            WF[j] = *(const bf16x8*)(wt_ + (j * 16 + r) * 128 + (((ks_ * 4 + g) ^ (r & 7)) << 4));                        \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                                  \
            const int row_ = w * 64 + i * 16 + r + ky_ * p.W + kx_;                                                       \
            PF[i] = *(const bf16x8*)(win + row_ * 128 + (((ks_ * 4 + g) ^ (row_ & 7)) << 4));                             \
        }                                                                                                                \
    }
#define C64_MMA(WF, PF, STEP)                                                                                            \
    {                                                                                                                    \
        const int ky_ = (STEP) / 6, kx_ = ((STEP) / 2) % 3;                                                              \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                                    \
            if (!(((vm[i] >> ky_) & 1u) && ((vm[i] >> (3 + kx_)) & 1u))) PF[i] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};          \
        _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                                    \
            _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                                \
                acc[j][i] = AV_MFMA_F32_16X16X32_LP(WF[j], PF[i], acc[j][i], 0, 0, 0);                   \
    }
        C64_LOAD(wfA, pfA, 0)
#pragma unroll
        for (int st = 0; st < 18; st += 2) {
            __builtin_amdgcn_s_waitcnt(0xC07F);                                          // lgkmcnt(0): set A is complete
            C64_LOAD(wfB, pfB, st + 1)
            __builtin_amdgcn_sched_barrier(0);
            C64_MMA(wfA, pfA, st)
            if (more) stage_k(t + 1, st / 2);                                            // windows t+1: instructions 0 .. 8 here, 9 below
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_waitcnt(0xC07F);                                          // set B is complete
            if (st + 2 < 18) C64_LOAD(wfA, pfA, st + 2)
            __builtin_amdgcn_sched_barrier(0);
            C64_MMA(wfB, pfB, st + 1)
            if (more && st == 0) stage_k(t + 1, 9);
            __builtin_amdgcn_sched_barrier(0);
        }
#undef C64_LOAD
#undef C64_MMA
        wg_barrier();                                                                // every wave is done reading window t: it becomes
        // ---- the output staging buffer.  Lane (r, g) holds channels 16j + 4g + e of pixel 64w + 16i + r
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int rl = w * 64 + i * 16 + r;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                bf16x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (bf16_t)acc[j][i][e];
                *(bf16x4*)(win + rl * 128 + (((2 * j + (g >> 1)) ^ (rl & 7)) << 4) + 8 * (g & 1)) = o;
            }
        }
        if (p.stats) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float s1 = 0.f, s2 = 0.f;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float v = (vm[i] & 2u) ? acc[j][i][e] : 0.f;                   // bit 1 is set for every pixel < M
                        s1 += v; s2 += v * v;
                    }
                    s1 = row16_sum(s1); s2 = row16_sum(s2);                                // over the 16 pixels on the lanes (DPP, no LDS)
                    if (r == 15) {
                        St[(w * 2 + 0) * 64 + j * 16 + 4 * g + e] = s1;
                        St[(w * 2 + 1) * 64 + j * 16 + 4 * g + e] = s2;
                    }
                }
        }
        wg_barrier();
        {   // 256 rows x 8 chunks, full 128-B rows per 8 lanes.  All LDS reads first, into distinct registers: a store whose data
            // register is reused by the next read makes the compiler drain vmcnt(0) in between (one store round trip per chunk)
            uint4 ov[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int id = k * 256 + tid, rl = id >> 3, ch = id & 7;
                ov[k] = *(const uint4*)(win + rl * 128 + ((ch ^ (rl & 7)) << 4));
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) asm volatile("" :: "v"(ov[k].x), "v"(ov[k].y), "v"(ov[k].z), "v"(ov[k].w));   // pin: no sinking into the stores
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int id = k * 256 + tid, rl = id >> 3, ch = id & 7;
                const long long px = (long long)p0 + rl;
                if (px < p.M) *(uint4*)(p.y + px * 64 + ch * 8) = ov[k];
            }
        }
        if (p.stats && tid < 128) {
            const int which = tid >> 6, c = tid & 63;
            p.stats[((long long)tile * 2 + which) * 64 + c] =
                St[(0 * 2 + which) * 64 + c] + St[(1 * 2 + which) * 64 + c] + St[(2 * 2 + which) * 64 + c] + St[(3 * 2 + which) * 64 + c];
        }
        wg_barrier();                                                                // staging buffer drained: tile t+1's loop refills it
    }
}

}  // namespace

extern "C" int av_conv3x3_c64(const void* x, const void* w, void* y, float* stats, int n_img, int H, int W, const float* in_scale,
                              const float* in_shift, const float* in_slope, void* stream) {
    AV_CHECK(x && w && y, "av_conv3x3_c64: null pointer");
    AV_CHECK((in_scale == nullptr) == (in_shift == nullptr) && (in_scale || !in_slope), "av_conv3x3_c64: in_scale / in_shift come together");
    AV_CHECK(n_img > 0 && H > 0 && W > 0 && W <= MAXW && (long long)n_img * H * W < (1ll << 24), "av_conv3x3_c64: bad shape n=%d H=%d W=%d (W <= %d, n*H*W < 2^24)", n_img, H, W, MAXW);
    AV_CHECK((uintptr_t)x % 16 == 0 && (uintptr_t)w % 16 == 0 && (uintptr_t)y % 16 == 0, "av_conv3x3_c64: operands must be 16-byte aligned");
    static int ncu = 0;
    if (!ncu) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu <= 0) ncu = 256;
        if (hipFuncSetAttribute((const void*)conv3x3_c64_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL) != hipSuccess ||
            hipFuncSetAttribute((const void*)conv3x3_c64_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL) != hipSuccess) {
            ncu = 0;
            av_set_error("av_conv3x3_c64: cannot raise dynamic LDS to %d", LDS_TOTAL);
            return AV_ERR_LAUNCH;
        }
    }
    CP p;
    p.x = (const bf16_t*)x; p.w = (const bf16_t*)w; p.y = (bf16_t*)y; p.stats = stats;
    p.in_scale = in_scale; p.in_shift = in_shift; p.in_slope = in_slope;
    p.M = n_img * H * W; p.H = H; p.W = W; p.ntiles = (p.M + CM - 1) / CM;
    const int G = p.ntiles < ncu ? p.ntiles : ncu;
    if (in_scale) hipLaunchKernelGGL(conv3x3_c64_kernel<true>, dim3(G), dim3(256), LDS_TOTAL, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(conv3x3_c64_kernel<false>, dim3(G), dim3(256), LDS_TOTAL, (hipStream_t)stream, p);
    AV_LAUNCH_CHECK();
    return AV_OK;
}
