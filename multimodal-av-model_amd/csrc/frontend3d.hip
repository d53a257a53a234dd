// Lip front-end Conv3d(1 -> 64, k (5,7,7), stride (1,2,2), pad (2,3,3), no bias) as an implicit GEMM on MFMA,
// bf16 perf mode (model/encoder.py:61).  With one input channel an im2col matrix would be 245x the input, so the A
// operand is never materialised: a workgroup keeps the 5 x 21 x 40 input patch of an 8 x 16 output tile in LDS and
// every lane builds its MFMA A fragment straight from it.  The K axis is re-laid out as 35 (kt,ky) tap rows x 8
// (kx padded 7 -> 8 with a zero weight) = 280 -> 288, so that one 16x16x32 fragment (8 consecutive k) is 8
// CONTIGUOUS pixels of one patch row (4 x ds_read_b32, 4-byte aligned because the stride-2 window starts at an even
// column).  The [64 x 288] weight image stays in LDS for the 6 tiles a workgroup processes.  Epilogue: fp32 LDS
// image -> 16-B row-contiguous stores of the channel-last output + per-tile BatchNorm partial sums.
#include "av_common.h"

namespace {

constexpr int KT = 5, KH = 7, KW = 7, TR = KT * KH;         // 35 tap rows
constexpr int KPAD = 288, WLD = 296;                        // padded K, LDS weight row (bf16 elements)
constexpr int TOY = 8, TOX = 16;                            // output tile
constexpr int PH = 2 * TOY + 5, PW = 40;                    // patch rows / padded row width (needs 2*15+8 = 38)
constexpr int CLD = 68;
constexpr int W_BYTES = 64 * WLD * 2;                       // 37 888
constexpr int R2_BYTES = 128 * CLD * 4;                     // 34 816 (>= patch 5*21*40*2 = 8 400)

struct FrontP {
    const float* x;      // [B*T][H][W] fp32 (single channel)
    const bf16_t* w;     // [64][KPAD] : k = (kt*7+ky)*8 + kx
    bf16_t* y;           // [B*T][Ho][Wo][64]
    float* stats;        // [tiles][2][64] or null
    int T, H, W, Ho, Wo;
    int prefetch;
};

__global__ __launch_bounds__(256, 2) void conv3d_front_kernel(const FrontP p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16_t* Wl = (bf16_t*)smem;
    bf16_t* patch = (bf16_t*)(smem + W_BYTES);
    float* cs = (float*)(smem + W_BYTES);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int r = lane & 15, g = lane >> 4;
    // XCD-aware order: workgroups are dealt round-robin to the 8 XCDs (private L2s) in launch order, and an output frame reads input
    // frames t-2 .. t+2 - with the natural order every XCD fetches (almost) every input frame (PMC: 951 MB read for 118 MB of input).
    // Bijective remap: XCD x works through a CONTIGUOUS range of frames, so the 4 shared input frames of consecutive outputs hit in its L2.
    int oxt, bt;
    {
        const int nx = gridDim.x, total = nx * gridDim.y;
        const int lin = blockIdx.y * nx + blockIdx.x;
        const int q = total >> 3, rem = total & 7, xcd = lin & 7, slot = lin >> 3;
        const int v = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + slot;
        bt = v / nx; oxt = v - bt * nx;
    }
    const int t = bt % p.T;
    const int ox0 = oxt * TOX;

    for (int i = tid; i < 64 * KPAD / 8; i += 256) {            // weights -> LDS, 16 B per thread
        const int row = i / (KPAD / 8), ch = i % (KPAD / 8);
        *(uint4*)(Wl + row * WLD + ch * 8) = *(const uint4*)(p.w + row * KPAD + ch * 8);
    }
    // tap rows handled by this lane: tr = 4*ks + g  (tr >= 35 multiplies zero weights: clamp the address)
    int aoff[9];
#pragma unroll
    for (int ks = 0; ks < 9; ++ks) {
        int tr = 4 * ks + g;
        if (tr > TR - 1) tr = TR - 1;
        const int kt = tr / KH, ky = tr - kt * KH;
        aoff[ks] = (kt * PH + ky) * PW + 2 * r;                 // + 2*oy_l*PW added per row tile
    }
    const int ntile = p.Ho / TOY;
    // 4 consecutive patch pixels per thread and iteration.  The loads of tile i+1 are issued right after tile i's patch is in LDS and
    // stay in flight (20 VGPRs) under tile i's MFMAs, epilogue and stores: their HBM / L2 round trip is off the per-tile critical path.
    // Loads are unconditional from clamped (always valid) addresses + select afterwards: predicated loads compile to one branch +
    // vmcnt(0) each, i.e. 20 serial memory round trips per tile.
    constexpr int NPATCH = KT * PH * (PW / 4), NIT = (NPATCH + 255) / 256;
    float pv[NIT][4];
    auto issue_loads = [&](int oy0) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int i = it * 256 + tid;
            const int pc = i % (PW / 4), q = i / (PW / 4);
            const int py = q % PH, kt = q / PH;
            const int ti = t + kt - 2, iy = 2 * oy0 - 3 + py, ix0 = 2 * ox0 - 3 + pc * 4;
            const int tic = ti < 0 ? 0 : (ti > p.T - 1 ? p.T - 1 : ti), iyc = iy < 0 ? 0 : (iy > p.H - 1 ? p.H - 1 : iy);
            const float* src = p.x + ((long long)(bt - t + tic) * p.H + iyc) * p.W;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int ix = ix0 + e;
                const int ixc = ix < 0 ? 0 : (ix > p.W - 1 ? p.W - 1 : ix);
                pv[it][e] = src[ixc];
            }
        }
    };
    auto write_patch = [&](int oy0) {
#pragma unroll
        for (int it = 0; it < NIT; ++it)                              // pin: keeps the loads unconditional and batched
            asm volatile("" :: "v"(pv[it][0]), "v"(pv[it][1]), "v"(pv[it][2]), "v"(pv[it][3]));
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int i = it * 256 + tid;
            if (i < NPATCH) {
                const int pc = i % (PW / 4), q = i / (PW / 4);
                const int py = q % PH, kt = q / PH;
                const int ti = t + kt - 2, iy = 2 * oy0 - 3 + py, ix0 = 2 * ox0 - 3 + pc * 4;
                const bool rowok = ti >= 0 && ti < p.T && iy >= 0 && iy < p.H;
                bf16x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (bf16_t)((rowok && ix0 + e >= 0 && ix0 + e < p.W) ? pv[it][e] : 0.f);
                *(bf16x4*)(patch + (kt * PH + py) * PW + pc * 4) = o;
            }
        }
    };
    if (p.prefetch) issue_loads(0);
    for (int oyt = 0; oyt < ntile; ++oyt) {
        const int oy0 = oyt * TOY;
        __syncthreads();                                        // previous tile's image reads are done
        if (!p.prefetch) issue_loads(oy0);
        write_patch(oy0);
        if (p.prefetch && oyt + 1 < ntile) issue_loads(oy0 + TOY);
        __syncthreads();
        f32x4 acc[2][4];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 9; ++ks) {
            bf16x8 b[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = *(const bf16x8*)(Wl + (j * 16 + r) * WLD + ks * 32 + 8 * g);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const unsigned* ap = (const unsigned*)(patch + aoff[ks] + 2 * (2 * w + i) * PW);
                uint4 av = make_uint4(ap[0], ap[1], ap[2], ap[3]);
                const bf16x8 a = __builtin_bit_cast(bf16x8, av);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = AV_MFMA_F32_16X16X32_LP(a, b[j], acc[i][j], 0, 0, 0);
            }
        }
        __syncthreads();                                        // all patch reads done: the image may overwrite it
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    cs[((2 * w + i) * 16 + 4 * g + e) * CLD + j * 16 + r] = acc[i][j][e];   // pixel = oy_l*16 + ox_l
        __syncthreads();
        {   // all LDS reads first, into distinct registers: a store whose data registers are reused by the next read costs a
            // vmcnt(0) round trip per iteration
            uint4 ov[4];
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int id = it * 256 + tid;
                const int pix = id >> 3, cc = (id & 7) * 8;
                const f32x4 v0 = *(const f32x4*)(cs + pix * CLD + cc), v1 = *(const f32x4*)(cs + pix * CLD + cc + 4);
                bf16x8 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) { o[e] = (bf16_t)v0[e]; o[4 + e] = (bf16_t)v1[e]; }
                ov[it] = __builtin_bit_cast(uint4, o);
            }
#pragma unroll
            for (int it = 0; it < 4; ++it) asm volatile("" :: "v"(ov[it].x), "v"(ov[it].y), "v"(ov[it].z), "v"(ov[it].w));
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int id = it * 256 + tid;
                const int pix = id >> 3, cc = (id & 7) * 8;
                const int oy = oy0 + (pix >> 4), ox = ox0 + (pix & 15);
                *(uint4*)(p.y + (((long long)bt * p.Ho + oy) * p.Wo + ox) * 64 + cc) = ov[it];
            }
        }
        if (p.stats) {                                           // 4 row groups x 64 columns, combined through LDS
            const int c = tid & 63, rg = tid >> 6;
            float s1 = 0.f, s2 = 0.f;
            for (int rr = rg * 32; rr < rg * 32 + 32; ++rr) { const float v = cs[rr * CLD + c]; s1 += v; s2 += v * v; }
            __syncthreads();                                    // image reads of the store loop are done
            float* red = cs;                                    // reuse the image: [4][2][64]
            red[(rg * 2 + 0) * 64 + c] = s1; red[(rg * 2 + 1) * 64 + c] = s2;
            __syncthreads();
            if (tid < 128) {
                const int which = tid >> 6;
                float* o = p.stats + (((long long)bt * ntile + oyt) * gridDim.x + oxt) * 128;
                o[which * 64 + c] = red[(0 * 2 + which) * 64 + c] + red[(1 * 2 + which) * 64 + c] + red[(2 * 2 + which) * 64 + c] + red[(3 * 2 + which) * 64 + c];
            }
        }
    }
}

}  // namespace

extern "C" int av_conv3d_front(const float* x, const void* w, void* y, float* stats, int B, int T, int H, int W, void* stream) {
    AV_CHECK(x && w && y && B > 0 && T > 0, "av_conv3d_front: bad args");
    AV_CHECK(H % 16 == 0 && W % 32 == 0, "av_conv3d_front: H=%d must be a multiple of 16 and W=%d of 32", H, W);
    static const int prefetch = [] { const char* e = getenv("AVAMD_FRONT_PREFETCH"); return (e && e[0] == '0') ? 0 : 1; }();
    FrontP p{x, (const bf16_t*)w, (bf16_t*)y, stats, T, H, W, H / 2, W / 2, prefetch};
    static bool done = false;
    const int lds = W_BYTES + R2_BYTES;
    if (!done) {
        if (hipFuncSetAttribute((const void*)conv3d_front_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) {
            av_set_error("av_conv3d_front: cannot raise dynamic LDS to %d", lds);
            return AV_ERR_LAUNCH;
        }
        done = true;
    }
    dim3 grid((unsigned)(p.Wo / TOX), (unsigned)(B * T));
    hipLaunchKernelGGL(conv3d_front_kernel, grid, dim3(256), lds, (hipStream_t)stream, p);
    AV_LAUNCH_CHECK();
    return AV_OK;
}
