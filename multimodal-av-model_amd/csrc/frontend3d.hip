// Lip front-end Conv3d(1 -> 64, k (5,7,7), stride (1,2,2), pad (2,3,3), no bias) as an implicit GEMM on MFMA,
// bf16 perf mode (model/encoder.py:61).  With one input channel an im2col matrix would be 245x the input, so the A
// operand is never materialised: a workgroup keeps the 5 x 21 x 40 input patch of an 8 x 16 output tile in LDS and
// every lane builds its MFMA A fragment straight from it.  The K axis is re-laid out as 35 (kt,ky) tap rows x 8
// (kx padded 7 -> 8 with a zero weight) = 280 -> 288, so that one 16x16x32 fragment (8 consecutive k) is 8
// CONTIGUOUS pixels of one patch row (4 x ds_read_b32, 4-byte aligned because the stride-2 window starts at an even
// column).  The [64 x 288] weight image stays in LDS for the 6 tiles a workgroup processes.  Epilogue: fp32 LDS
// image -> 16-B row-contiguous stores of the channel-last output + per-tile BatchNorm partial sums.
//
// POOL form (av_conv3d_front_pool): the front-end's BatchNorm3d + PReLU + MaxPool3d((1,3,3),(1,2,2),(0,1,1)) (model/encoder.py:62-64) read the
// 1.9 GB conv output of a 64 x 100-frame batch once more and shrink it four-fold.  BatchNorm (an affine map per channel, known only after the
// whole batch) followed by PReLU is monotone or V-shaped per channel, so max over a pooling window of prelu(bn(x)) = max(prelu(bn(max x)),
// prelu(bn(min x))) EXACTLY (fp32 rounding keeps each piece monotone): the kernel pools the raw (bf16-rounded) conv output instead - per 3 x 3
// window the per-channel MAX and MIN - and writes those two quarter-size tensors; av_bn_prelu_minmax finishes the job once the statistics
// exist.  The conv output itself never reaches HBM (1.9 GB written + 1.9 GB read -> 2 x 0.47 GB written + read).  Windows overlap by one
// pixel: the row above a tile comes from the previous tile of the same workgroup (it walks a 16-column strip top to bottom; a carry row in
// LDS), the column left of the strip is RECOMPUTED here (a ninth 16-pixel MFMA row tile per K-step, one n-tile per wavefront: +12.5 %
// MFMAs, the patch is 2 input columns wider - exactly the 40 it already had room for).
#include "av_common.h"

namespace {

constexpr int KT = 5, KH = 7, KW = 7, TR = KT * KH;         // 35 tap rows
constexpr int KPAD = 288, WLD = 296;                        // padded K, LDS weight row (bf16 elements)
constexpr int TOY = 8, TOX = 16;                            // output tile
constexpr int PH = 2 * TOY + 5, PW = 40;                    // patch rows / padded row width (needs 2*15+8 = 38)
constexpr int CLD = 68;
constexpr int W_BYTES = 64 * WLD * 2;                       // 37 888
constexpr int R2_BYTES = 128 * CLD * 4;                     // 34 816 (>= patch 5*21*40*2 = 8 400)
constexpr int R2P_BYTES = 136 * CLD * 4;                    // POOL: + the 8 pixels of the halo column
constexpr int CARRY_BYTES = 17 * CLD * 4;                   // POOL: last conv row of the previous tile (halo column + 16)

struct FrontP {
    const float* x;      // [B*T][H][W] fp32 (single channel)
    const bf16_t* w;     // [64][KPAD] : k = (kt*7+ky)*8 + kx
    bf16_t* y;           // [B*T][Ho][Wo][64]
    float* stats;        // [tiles][2][64] or null
    int T, H, W, Ho, Wo;
    int prefetch;
    bf16_t *ymax, *ymin; // POOL: [B*T][Ho/2][Wo/2][64] per-window maximum / minimum of the conv output
};

template <bool POOL>
__global__ __launch_bounds__(256, 2) void conv3d_front_kernel(const FrontP p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16_t* Wl = (bf16_t*)smem;
    bf16_t* patch = (bf16_t*)(smem + W_BYTES);
    float* cs = (float*)(smem + W_BYTES);
    float* carry = (float*)(smem + W_BYTES + R2P_BYTES);        // POOL only
    constexpr int XS = POOL ? 2 : 0;                            // POOL: the patch starts 2 input columns further left (halo output column)
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
        aoff[ks] = (kt * PH + ky) * PW + 2 * r + XS;            // + 2*oy_l*PW added per row tile
    }
    int aoffh[9];                                               // POOL: halo column (output column ox0 - 1), lane r < 8 = output row r of the tile
    if constexpr (POOL) {
#pragma unroll
        for (int ks = 0; ks < 9; ++ks) {
            int tr = 4 * ks + g;
            if (tr > TR - 1) tr = TR - 1;
            const int kt = tr / KH, ky = tr - kt * KH;
            aoffh[ks] = (kt * PH + ky + 2 * (r < 8 ? r : 7)) * PW;
        }
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
            const int ti = t + kt - 2, iy = 2 * oy0 - 3 + py, ix0 = 2 * ox0 - 3 - XS + pc * 4;
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
                const int ti = t + kt - 2, iy = 2 * oy0 - 3 + py, ix0 = 2 * ox0 - 3 - XS + pc * 4;
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
        f32x4 acch = f32x4{0.f, 0.f, 0.f, 0.f};                 // POOL: halo column tile, n-tile w of it
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
            if constexpr (POOL) {
                const unsigned* ap = (const unsigned*)(patch + aoffh[ks]);
                uint4 av = make_uint4(ap[0], ap[1], ap[2], ap[3]);
                const bf16x8 bw = *(const bf16x8*)(Wl + (w * 16 + r) * WLD + ks * 32 + 8 * g);      // (b[w] with a run-time w would go to scratch)
                acch = AV_MFMA_F32_16X16X32_LP(__builtin_bit_cast(bf16x8, av), bw, acch, 0, 0, 0);
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
        if constexpr (POOL) {
            if (g < 2) {                                        // halo pixels 128 + oy_l (rows 8 .. 15 of that MFMA tile are duplicates)
#pragma unroll
                for (int e = 0; e < 4; ++e) cs[(128 + 4 * g + e) * CLD + w * 16 + r] = acch[e];
            }
        }
        __syncthreads();
        if constexpr (POOL) {
            // 4 x 8 pooled pixels x 8 channel chunks = 256 threads: window rows 2 pyl - 1 .. + 1 (row -1 = the carry row; none above the first
            // tile), columns 2 pxl - 1 .. + 1 (column -1 = the halo pixels; none left of the first strip).  The conv output is rounded to bf16
            // before it is compared - exactly the values the unfused path stores and pools
            const int pp = tid >> 3, pyl = pp >> 3, pxl = pp & 7, cc = (tid & 7) * 8;
            float mx[8], mn[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) { mx[e] = -INFINITY; mn[e] = INFINITY; }
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                const int ry = 2 * pyl - 1 + dy;
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    const int cx = 2 * pxl - 1 + dx;
                    const bool ok = (ry >= 0 || oyt > 0) && (cx >= 0 || ox0 > 0);
                    const float* src = ry >= 0 ? (cx >= 0 ? cs + (ry * 16 + cx) * CLD : cs + (128 + ry) * CLD) : carry + (cx + 1) * CLD;
                    const f32x4 v0 = *(const f32x4*)(src + cc), v1 = *(const f32x4*)(src + cc + 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float a0 = (float)(bf16_t)v0[e], a1 = (float)(bf16_t)v1[e];
                        mx[e] = ok ? fmaxf(mx[e], a0) : mx[e]; mn[e] = ok ? fminf(mn[e], a0) : mn[e];
                        mx[4 + e] = ok ? fmaxf(mx[4 + e], a1) : mx[4 + e]; mn[4 + e] = ok ? fminf(mn[4 + e], a1) : mn[4 + e];
                    }
                }
            }
            bf16x8 omx, omn;
#pragma unroll
            for (int e = 0; e < 8; ++e) { omx[e] = (bf16_t)mx[e]; omn[e] = (bf16_t)mn[e]; }
            const long long po = (((long long)bt * (p.Ho / 2) + oy0 / 2 + pyl) * (p.Wo / 2) + ox0 / 2 + pxl) * 64 + cc;
            *(bf16x8*)(p.ymax + po) = omx;
            *(bf16x8*)(p.ymin + po) = omn;
        } else
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
        {                                                        // BatchNorm partials: 4 row groups x 64 columns, combined through LDS
            const int c = tid & 63, rg = tid >> 6;
            float s1 = 0.f, s2 = 0.f;
            if (p.stats)
                for (int rr = rg * 32; rr < rg * 32 + 32; ++rr) { const float v = cs[rr * CLD + c]; s1 += v; s2 += v * v; }
            if (p.stats || POOL) __syncthreads();               // image reads of the store / pooling loop and of the sums are done
            if constexpr (POOL) {
                // carry = conv row 7 of this tile (halo pixel 135, then pixels 112 .. 127) for the first pooled row of the next tile; the
                // partial-sum scratch below lives in image rows 0 .. 7 only
                for (int i = tid; i < 17 * 16; i += 256) {
                    const int px = i >> 4, c4 = (i & 15) * 4;
                    *(f32x4*)(carry + px * CLD + c4) = *(const f32x4*)(cs + (px == 0 ? 135 : 111 + px) * CLD + c4);
                }
            }
            if (p.stats) {
                float* red = cs;                                // reuse the image: [4][2][64]
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
}

// finishes the POOL form once the batch statistics exist: out = max(prelu(bn(max)), prelu(bn(min))) per channel (see the file header)
__global__ __launch_bounds__(256) void bn_prelu_minmax_kernel(const bf16x8* __restrict__ ymax, const bf16x8* __restrict__ ymin, const float* __restrict__ scale,
                                                              const float* __restrict__ shift, const float* __restrict__ slope, bf16x8* __restrict__ out, long long n8) {
    const long long stride = (long long)gridDim.x * blockDim.x;      // a multiple of 8 chunks: fixed channel group per thread
    long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int cg = (int)(e & 7) * 8;
    float sc[8], sh[8], sl[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { sc[i] = scale[cg + i]; sh[i] = shift[cg + i]; sl[i] = slope[cg + i]; }
    for (; e < n8; e += 4 * stride) {
        bf16x8 a[4], b[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const long long i = e + u * stride < n8 ? e + u * stride : e; a[u] = ymax[i]; b[u] = ymin[i]; }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (e + u * stride < n8) {
                bf16x8 o;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    float va = (float)a[u][i] * sc[i] + sh[i], vb = (float)b[u][i] * sc[i] + sh[i];      // the unfused kernel's operation order
                    va = va >= 0.f ? va : va * sl[i];
                    vb = vb >= 0.f ? vb : vb * sl[i];
                    o[i] = (bf16_t)fmaxf(va, vb);
                }
                out[e + u * stride] = o;
            }
        }
    }
}

}  // namespace

static int front_launch(const FrontP& p, int B, bool pool, hipStream_t st) {
    static bool done[2] = {false, false};
    const int lds = pool ? W_BYTES + R2P_BYTES + CARRY_BYTES : W_BYTES + R2_BYTES;
    if (!done[pool]) {
        const hipError_t e = pool ? hipFuncSetAttribute((const void*)conv3d_front_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds)
                                  : hipFuncSetAttribute((const void*)conv3d_front_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) {
            av_set_error("av_conv3d_front: cannot raise dynamic LDS to %d", lds);
            return AV_ERR_LAUNCH;
        }
        done[pool] = true;
    }
    dim3 grid((unsigned)(p.Wo / TOX), (unsigned)(B * p.T));
    if (pool) hipLaunchKernelGGL(conv3d_front_kernel<true>, grid, dim3(256), lds, st, p);
    else hipLaunchKernelGGL(conv3d_front_kernel<false>, grid, dim3(256), lds, st, p);
    AV_LAUNCH_CHECK();
    return AV_OK;
}

extern "C" int av_conv3d_front(const float* x, const void* w, void* y, float* stats, int B, int T, int H, int W, void* stream) {
    AV_CHECK(x && w && y && B > 0 && T > 0, "av_conv3d_front: bad args");
    AV_CHECK(H % 16 == 0 && W % 32 == 0, "av_conv3d_front: H=%d must be a multiple of 16 and W=%d of 32", H, W);
    static const int prefetch = [] { const char* e = getenv("AVAMD_FRONT_PREFETCH"); return (e && e[0] == '0') ? 0 : 1; }();
    FrontP p{x, (const bf16_t*)w, (bf16_t*)y, stats, T, H, W, H / 2, W / 2, prefetch, nullptr, nullptr};
    return front_launch(p, B, false, (hipStream_t)stream);
}

extern "C" int av_conv3d_front_pool(const float* x, const void* w, void* ymax, void* ymin, float* stats, int B, int T, int H, int W, void* stream) {
    AV_CHECK(x && w && ymax && ymin && B > 0 && T > 0, "av_conv3d_front_pool: bad args");
    AV_CHECK(H % 16 == 0 && W % 32 == 0, "av_conv3d_front_pool: H=%d must be a multiple of 16 and W=%d of 32", H, W);
    AV_CHECK((uintptr_t)ymax % 16 == 0 && (uintptr_t)ymin % 16 == 0, "av_conv3d_front_pool: outputs must be 16-byte aligned");
    static const int prefetch = [] { const char* e = getenv("AVAMD_FRONT_PREFETCH"); return (e && e[0] == '0') ? 0 : 1; }();
    FrontP p{x, (const bf16_t*)w, nullptr, stats, T, H, W, H / 2, W / 2, prefetch, (bf16_t*)ymax, (bf16_t*)ymin};
    return front_launch(p, B, true, (hipStream_t)stream);
}

extern "C" int av_bn_prelu_minmax(const void* ymax, const void* ymin, const float* scale, const float* shift, const float* slope, void* out,
                                  long long n, void* stream) {
    AV_CHECK(ymax && ymin && scale && shift && slope && out && n >= 0 && n % 64 == 0, "av_bn_prelu_minmax: bad args (n must be a multiple of the 64 channels)");
    AV_CHECK((uintptr_t)ymax % 16 == 0 && (uintptr_t)ymin % 16 == 0 && (uintptr_t)out % 16 == 0, "av_bn_prelu_minmax: 16-byte alignment");
    if (n == 0) return AV_OK;
    const long long n8 = n / 8;
    long long blocks = (n8 + 4 * 256 - 1) / (4 * 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(bn_prelu_minmax_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const bf16x8*)ymax, (const bf16x8*)ymin, scale, shift,
                       slope, (bf16x8*)out, n8);
    AV_LAUNCH_CHECK();
    return AV_OK;
}
