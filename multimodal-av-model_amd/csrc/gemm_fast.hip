// Fast path of av_gemm for the dominant shape class: bf16, C = epilogue(alpha * A[M,K] x B[N,K]^T), both operands
// K-contiguous (nn.Linear forward; dX and dW are brought to this form with cached W^T / one HBM-bound transpose).
//
//   * 128 x 128 x 64 block tile, 256 threads = 4 wavefronts (2 x 2), 64 x 64 per wave = 4 x 4 MFMA 16x16x32 tiles;
//   * global -> LDS with direct `global_load_lds_dwordx4` (no VGPR staging): one wave instruction fills 8 tile rows
//     of 128 B; two LDS buffers, the loads of K-tile t+1 are issued BEFORE the MFMA phase of tile t and retired by
//     the single barrier per K-tile (guide T3/T4 "minimum 2-phase" structure);
//   * LDS rows are 128 B (no padding is possible with LDS-DMA), so the 16-B chunk index is XOR-swizzled with
//     (row & 7): applied on the per-lane SOURCE address when staging and on the ds_read_b128 fragment address
//     (both-sides rule) -> conflict-free fragment reads;
//   * XCD-aware bijective remap of blockIdx so that tiles sharing an A panel run on one XCD (private L2);
//   * epilogue through an fp32 LDS image so that bias / GELU / gelu' x aux / residual / C2 and the stores are
//     16-32 B per lane and row-contiguous.
#include <cstdlib>

#include "av_common.h"

namespace {

constexpr int BM = 128, BK = 64, NT = 256;
constexpr int TILE_A = BM * BK * 2;       // 16 KiB A tile per buffer

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __attribute__((aligned(256))) unsigned char g_zero_line[256];   // source of padded / out-of-range rows

// Diagnostic build only (-DAV_GEMM_STAMPS, tools/gemm_stamps.py): wall-clock stamps (s_memrealtime, 100 MHz) of every workgroup of the
// 8-phase kernel at entry / first MFMA phase / end of the main loop / exit, into a buffer nothing else reads.  No stamp exists in the
// product build.
#ifdef AV_GEMM_STAMPS
__device__ unsigned long long g_gemm_stamps[8192 * 4];
#define AV_STAMP(SLOT) do { if (threadIdx.x == 0 && blockIdx.x < 8192) g_gemm_stamps[blockIdx.x * 4 + (SLOT)] = __builtin_amdgcn_s_memrealtime(); } while (0)
// persistent kernel (v7): [workgroup < 256][tile of the workgroup < 16][tile start, loop begin, loop end, stores issued]
__device__ unsigned long long g_v7_stamps[256 * 16 * 4];
#define AV_STAMP7(SEQ, SLOT) do { if (threadIdx.x == 0 && blockIdx.x < 256 && (SEQ) < 16) g_v7_stamps[(blockIdx.x * 16 + (SEQ)) * 4 + (SLOT)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define AV_STAMP(SLOT) do { } while (0)
#define AV_STAMP7(SEQ, SLOT) do { } while (0)
#endif

// NWI wave-instructions per wave, each filling 8 tile rows x 128 B (lane -> row sub = lane>>3, physical chunk lane&7;
// the global chunk is pch ^ sub: XOR swizzle applied on the SOURCE side, LDS image stays lane-linear)
template <int NWI>
__device__ __forceinline__ void stage_rows(const bf16_t* __restrict__ base, long long ld, int row0, int nrows, int k0, char* tile,
                                           int w, int lane) {
    const int sub = lane >> 3, pch = lane & 7;
#pragma unroll
    for (int i = 0; i < NWI; ++i) {
        const int row = (w * NWI + i) * 8 + sub;
        int gr = row0 + row;
        if (gr > nrows - 1) gr = nrows - 1;                      // clamped rows are computed but never stored
        const bf16_t* src = base + (long long)gr * ld + k0 + ((pch ^ sub) << 3);
        const unsigned off = __builtin_amdgcn_readfirstlane((unsigned)((w * NWI + i) * 1024));
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(tile + off), 16, 0, 0);
    }
}


// ---- k-major operands (element (k, c) at base[k * ld + c]: the A of dW = dY^T X, the B of dX = dY W) --------------------
// Tile = 64 k-rows x 128 columns (256-B rows), filled by LDS-DMA (a wave instruction = 4 rows); MFMA fragments come out of it
// with ds_read_b64_tr_b16: a 16-lane group reads 4 k-rows x 16 columns and receives them column-major, so lane (c = lane & 15,
// g = lane >> 4) gets k = 8g .. 8g+7 of its column from two reads (rows 8g + 0..3 and 8g + 4..7) - the natural k order of the
// 16x16x32 operand, i.e. compatible with a row-major partner read with ds_read_b128.  The 16-B chunk of a row is XOR-swizzled
// with s(row) = ((row & 3) << 2) | ((row >> 2) & 3) (applied to the DMA source), which makes those reads bank-conflict free.
typedef __attribute__((address_space(3))) bf16x4* lds_b4_t;

template <int NWI = 4>                                       // wave instructions per wavefront: 4 (four wavefronts fill the tile) or 2 (eight)
__device__ __forceinline__ void stage_kmajor(const bf16_t* __restrict__ base, long long ld, int col0, int ncols, int k0, int K, char* tile,
                                             int w, int lane) {
    const int rsub = lane >> 4, pos = lane & 15;
#pragma unroll
    for (int i = 0; i < NWI; ++i) {
        const int row = (w * NWI + i) * 4 + rsub;
        const int sx = ((row & 3) << 2) | ((row >> 2) & 3);
        const int col = col0 + ((pos ^ sx) << 3), k = k0 + row;
        const bf16_t* src = (k < K && col < ncols) ? base + (long long)k * ld + col : (const bf16_t*)g_zero_line;
        const unsigned off = __builtin_amdgcn_readfirstlane((unsigned)((w * NWI + i) * 1024));
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(tile + off), 16, 0, 0);
    }
}

// ---- k-major half-tiles of the 8-phase kernel (gemm_nt_bf16_v4_kernel<false, true>) ---------------------------------------------------------
// Image of a half-tile [64 k-rows][128 columns]: 8-row x 32-column subtiles of 512 B,
//     off(row, ch) = 2048 (row >> 3) + 512 (ch >> 2) + 64 (row & 7) + 16 ((ch & 3) ^ ((row >> 2) & 3))          (ch = 16-B chunk of the row)
// - a 32-lane half of a transposed read (8 k-rows x 32 B) covers all 64 banks once, and the offset of column tile ct, k-step ks splits into a
// per-lane base that depends on ct only through its parity and a COMPILE-TIME part 8192 ks + 512 (ct >> 1): the reads take it as the
// instruction's immediate offset (4 address adds per half-tile instead of one per read).
// LDS-DMA: wave instruction q = 2 w + i of a half-tile fills bytes [1024 q, 1024 q + 1024) = rows 8 (q >> 1) .. + 7 x columns 64 (q & 1) .. + 63,
// each lane quad 64 contiguous bytes of one row (lane l: row 8 (q >> 1) + ((l >> 2) & 7), chunk 4 (2 (q & 1) + (l >> 5)) + ((l & 3) ^ x)).
// The lane-dependent part of the source address is hoisted out of the K loop (row * ld + col as a 32-bit element offset, the K-tile adds the
// wave-uniform k0 * ld), and the common path has NO per-lane select: the instructions between a phase's fragment reads and its barrier are on
// the critical path of the staggered schedule - with the (k < K && col < ncols ? src : zero line) select of stage_kmajor in front of every
// LDS-DMA (compare, mask, two v_cndmask, a 64-bit multiply) the DMA-only loop ran at 1.78 us per K-tile, without it at 1.19 (row-major form: 1.03).
struct KmLane { int off[2]; };
__device__ __forceinline__ void km_lane_init(KmLane& kl, long long ld, int col0, int ncols, int w, int lane) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int q = w * 2 + i;
        const int row = 8 * (q >> 1) + ((lane >> 2) & 7);
        const int x = (((q >> 1) & 1) << 1) | ((lane >> 4) & 1);                    // (row >> 2) & 3
        int col = col0 + 8 * (4 * (2 * (q & 1) + (lane >> 5)) + ((lane & 3) ^ x));
        if (col > ncols - 8) col = ncols - 8;                    // columns past the operand: clamped re-reads, multiplied and never stored (ncols % 8 == 0)
        kl.off[i] = row * (int)ld + col;
    }
}
__device__ __forceinline__ void stage_kmajor8(const bf16_t* __restrict__ base, long long ld, const KmLane& kl, int k0, int K, char* tile, int w, int lane) {
    const bf16_t* tb = base + (long long)k0 * ld;                // wave-uniform
    if (k0 + 64 <= K) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const unsigned off = __builtin_amdgcn_readfirstlane((unsigned)((w * 2 + i) * 1024));
            __builtin_amdgcn_global_load_lds((gptr_t)(tb + kl.off[i]), (lptr_t)(tile + off), 16, 0, 0);
        }
    } else {                                                     // a slice's ragged last K-tile: k-rows beyond it come from the zero line
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = 8 * w + ((lane >> 2) & 7);           // 8 ((2 w + i) >> 1) + ...
            const bf16_t* src = k0 + row < K ? tb + kl.off[i] : (const bf16_t*)g_zero_line;
            const unsigned off = __builtin_amdgcn_readfirstlane((unsigned)((w * 2 + i) * 1024));
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(tile + off), 16, 0, 0);
        }
    }
}
// per-lane base offsets of the transposed reads: [parity of the column tile][rows 8 g + 0..3 | 8 g + 4..7]; `fold` = the wave-uniform part of
// 512 (ct >> 1) (A: 1024 wr, B: 512 wc)
struct Km8Base { unsigned b[2][2]; };
__device__ __forceinline__ void km8_base_init(Km8Base& kb, int fold, int lane) {
    const int g = lane >> 4, qq = (lane >> 2) & 3, pp = lane & 3;
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int hi = 0; hi < 2; ++hi)
            kb.b[e][hi] = (unsigned)(fold + 2048 * g + 64 * qq + 256 * hi + 32 * (e ^ (g & 1)) + 16 * ((pp >> 1) ^ hi) + 8 * (pp & 1));
}
template <int IMM>
__device__ __forceinline__ bf16x4 ds_read_tr16_b64_imm(unsigned addr) {
    bf16x4 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(IMM) : "memory");
    return v;
}
// The two halves of a fragment stay SEPARATE values until the wait that retires the reads: the compiler takes an asm result as ready at once,
// so a join (register copies into one 4-register MFMA operand) placed between the read and the wait would copy stale registers.  The kernel
// ties the halves to its wait (km8_tie*: empty asm with the halves as in/out operands) and joins them after it.
template <int IMM>
__device__ __forceinline__ void km8_read(unsigned lo_addr, unsigned hi_addr, bf16x4 (&d)[2]) {
    d[0] = ds_read_tr16_b64_imm<IMM>(lo_addr); d[1] = ds_read_tr16_b64_imm<IMM>(hi_addr);
}
__device__ __forceinline__ void km8_tie8(bf16x4 (&f)[2][2][2]) {     // [tile][k-step][half]
    asm volatile("" : "+v"(f[0][0][0]), "+v"(f[0][0][1]), "+v"(f[0][1][0]), "+v"(f[0][1][1]), "+v"(f[1][0][0]), "+v"(f[1][0][1]), "+v"(f[1][1][0]), "+v"(f[1][1][1]));
}
__device__ __forceinline__ bf16x8 km8_join(const bf16x4 (&d)[2]) { return __builtin_shufflevector(d[0], d[1], 0, 1, 2, 3, 4, 5, 6, 7); }

// per-lane byte offsets (inside a k-major tile, k-step 0) of the two transposed reads of column tile `ct`
struct KmOff { int lo, hi; };
__device__ __forceinline__ KmOff kmajor_off(int ct, int lane) {
    const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
    const int r1 = 8 * g + q, r2 = r1 + 4;
    const int s1 = ((r1 & 3) << 2) | ((r1 >> 2) & 3), s2 = ((r2 & 3) << 2) | ((r2 >> 2) & 3);
    const int ch = 2 * ct + (pp >> 1);
    KmOff o;
    o.lo = r1 * 256 + ((ch ^ s1) << 4) + 8 * (pp & 1);
    o.hi = r2 * 256 + ((ch ^ s2) << 4) + 8 * (pp & 1);
    return o;
}
// The transposed reads are issued as inline asm, not through the builtin: next to LDS-DMA the compiler puts `s_waitcnt vmcnt(0)` in front of every
// group of builtin transposed reads (it cannot tell which in-flight global_load_lds they depend on), i.e. the whole prefetch queue drained before
// each fragment read - measured: the 128 x 128 kernel below ran its k-major products load -> compute serialised (22 % MFMA-busy), and the
// 8-phase form at 2.7 us per K-tile instead of 1.55.  The asm form is invisible to that pass: the CALLER waits (lgkmcnt(0) before the MFMAs that
// consume the fragments; vmcnt for the DMA as for the row-major tiles).
__device__ __forceinline__ bf16x4 ds_read_tr16_b64_asm(unsigned addr) {
    bf16x4 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(addr) : "memory");
    return v;
}
// (the two halves of a fragment stay separate values until the caller's wait: kmajor_tie4 carries them through it as in/out operands, so the
// register copies that join them into one MFMA operand cannot be scheduled before the data has arrived)
__device__ __forceinline__ void kmajor_read(const char* tile, const KmOff& o, int ks, bf16x4 (&d)[2]) {
    const unsigned base = (unsigned)(unsigned long long)(lds_b4_t)(tile + ks * 32 * 256);
    d[0] = ds_read_tr16_b64_asm(base + (unsigned)o.lo);
    d[1] = ds_read_tr16_b64_asm(base + (unsigned)o.hi);
}
__device__ __forceinline__ void kmajor_tie4(bf16x4 (&f)[4][2]) {
    asm volatile("" : "+v"(f[0][0]), "+v"(f[0][1]), "+v"(f[1][0]), "+v"(f[1][1]), "+v"(f[2][0]), "+v"(f[2][1]), "+v"(f[3][0]), "+v"(f[3][1]));
}

// Position-major pixel order (av_gemm_args.cNF / cPM): decode an output row and address an input pixel.
// Blocked: images are taken in blocks of cNF; inside a block the order is [position][image of the block], i.e. image q's pixel `pos` of a P-pixel
// map is row ((q / cNF) * P + pos) * cNF + q % cNF.  A row tile then lies at ONE position, and consecutive row tiles are the neighbouring
// positions of the SAME images: their input windows overlap and are re-read from this XCD's L2 (with one block = the whole batch every tap
// of every tile fetched 256 fresh pixel rows from beyond L2 and the skipped taps bought only 6 %).
__device__ __forceinline__ void conv_decode(const av_gemm_args& p, int m, int& q, int& oy, int& ox) {
    if (p.cPM & 2) {
        const int per = p.cOh * p.cOw * p.cNF, blk = m / per, rem = m - blk * per, pos = rem / p.cNF;
        q = blk * p.cNF + (rem - pos * p.cNF); oy = pos / p.cOw; ox = pos - oy * p.cOw;
    } else { ox = m % p.cOw; int t = m / p.cOw; oy = t % p.cOh; q = t / p.cOh; }
}
__device__ __forceinline__ long long conv_pixel(const av_gemm_args& p, int q, int iy, int ix) {        // row index of input pixel (q, iy, ix)
    if (p.cPM & 1) { const int blk = q / p.cNF; return ((long long)blk * (p.cH * p.cW) + iy * p.cW + ix) * p.cNF + (q - blk * p.cNF); }
    return ((long long)q * p.cH + iy) * p.cW + ix;
}
// K-tile sequence of a row tile: all rows of [m0, m0 + rows) share one output position (position-major output, tile inside one position's
// block of cNF rows) => the taps outside the image for that position are dropped: seq holds the remaining tap indices (4 bits each).
// Otherwise every tap stays.  K-tile t covers tap seq[t >> sh], channels 64 (t & (cpt - 1)) ..; cpt = Cin / 64 = 1 << sh.
struct TapSeq { unsigned long long seq; int ntap, sh; };
__device__ __forceinline__ TapSeq tap_seq(const av_gemm_args& p, int m0, int rows) {
    TapSeq ts;
    const int ntaps = p.cKh * p.cKw, cpt = p.cCin >> 6;
    ts.sh = 31 - __builtin_clz(cpt);
    unsigned mask = ntaps >= 32 ? 0xffffffffu : ((1u << ntaps) - 1u);
    int mlast = m0 + rows - 1;
    if (mlast > p.M - 1) mlast = p.M - 1;
    if ((p.cPM & 2) && ntaps <= 16 && (1 << ts.sh) == cpt && m0 / p.cNF == mlast / p.cNF) {
        const int pos = (m0 / p.cNF) % (p.cOh * p.cOw), oy = pos / p.cOw, ox = pos - oy * p.cOw;
        const int iy0 = oy * p.cSh - p.cPh, ix0 = ox * p.cSw - p.cPw;
        mask = 0u;
        for (int ky = 0; ky < p.cKh; ++ky)
            for (int kx = 0; kx < p.cKw; ++kx)
                if (iy0 + ky >= 0 && iy0 + ky < p.cH && ix0 + kx >= 0 && ix0 + kx < p.cW) mask |= 1u << (ky * p.cKw + kx);
    }
    ts.seq = 0ull; ts.ntap = 0;
    if (ntaps <= 16 && (1 << ts.sh) == cpt) {
        for (int t = 0; t < ntaps; ++t)
            if ((mask >> t) & 1u) { ts.seq |= (unsigned long long)t << (4 * ts.ntap); ++ts.ntap; }
    } else { ts.ntap = -1; }                                 // no table: K-tile t is simply channels 64 t of the tap-major K axis
    return ts;
}
__device__ __forceinline__ int tap_k0(const av_gemm_args& p, const TapSeq& ts, int t) {
    if (ts.ntap < 0) return t * BK;
    const int tap = (int)((ts.seq >> (4 * (t >> ts.sh))) & 15ull);
    return tap * p.cCin + ((t & ((1 << ts.sh) - 1)) << 6);
}

// per tile row handled by this lane: pointer to the (ky = 0, kx = 0) tap pixel (+ the lane's swizzled 16-B chunk) and a bit mask of
// the taps that fall inside the image (bit ky * Kw + kx for Kh * Kw <= 32; the valid ky range for longer 1-D filters).  A K-step then costs one 64-bit add and a select per row.
struct ConvRows { const bf16_t* rowp[4]; unsigned valid[4]; };

__device__ __forceinline__ void conv_rows_init(ConvRows& cr, const bf16_t* __restrict__ base, const av_gemm_args& p, int m0, int w, int lane) {
    const int sub = lane >> 3, choff = ((lane & 7) ^ sub) << 3;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + (w * 4 + i) * 8 + sub;
        cr.rowp[i] = base;
        cr.valid[i] = 0u;
        if (m < p.M) {
            int q, oy, ox;
            conv_decode(p, m, q, oy, ox);
            const int iy0 = oy * p.cSh - p.cPh, ix0 = ox * p.cSw - p.cPw;
            cr.rowp[i] = base + conv_pixel(p, q, iy0, ix0) * p.cCtot + p.cCoff + choff;
            unsigned v = 0u;
            if (p.cKh * p.cKw <= 32) {
                for (int ky = 0; ky < p.cKh; ++ky)
                    for (int kx = 0; kx < p.cKw; ++kx)
                        if (iy0 + ky >= 0 && iy0 + ky < p.cH && ix0 + kx >= 0 && ix0 + kx < p.cW) v |= 1u << (ky * p.cKw + kx);
            } else {                                            // long 1-D filters (Kw = 1, positional conv): valid ky range [lo, hi)
                int lo = -iy0, hi = p.cH - iy0;
                lo = lo < 0 ? 0 : lo; hi = hi > p.cKh ? p.cKh : hi; hi = hi < lo ? lo : hi;
                v = (unsigned)lo | ((unsigned)hi << 16);
            }
            cr.valid[i] = v;
        }
    }
}

// implicit im2col of an NHWC image: one K tile (64 channels) lies inside ONE filter tap (Cin % 64 == 0), so every
// tile row is a contiguous 128-B run of the input pixel (or the zero line for padding / rows >= M)
__device__ __forceinline__ void stage_conv(const av_gemm_args& p, const ConvRows& cr, int k0, char* tile, int w, int lane) {
    const int sub = lane >> 3, pch = lane & 7;
    const int tap = k0 / p.cCin, c0 = k0 - tap * p.cCin;
    const int ky = tap / p.cKw, kx = tap - ky * p.cKw;
    const long long toff = ((long long)ky * p.cW + kx) * ((p.cPM & 1) ? p.cNF : 1) * p.cCtot + c0;          // wave-uniform
    const bf16_t* zl = (const bf16_t*)g_zero_line + ((pch ^ sub) << 3);
    const bool ranged = p.cKh * p.cKw > 32;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const bool ok = ranged ? (tap >= (int)(cr.valid[i] & 0xffffu) && tap < (int)(cr.valid[i] >> 16)) : ((cr.valid[i] >> tap) & 1u);
        const bf16_t* src = ok ? cr.rowp[i] + toff : zl;
        const unsigned off = __builtin_amdgcn_readfirstlane((unsigned)((w * 4 + i) * 1024));
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(tile + off), 16, 0, 0);
    }
}

struct FastFlags { int c_vec, r_vec, aux_vec; };

// shared tail of the coalesced epilogue: optional pre-activation copy, activation, dropout, residual, store (8 columns)
// VO ("vector only", v7's specialised instantiations): the caller guarantees whole, 16-B aligned chunks for C / C2 / R / aux (N % 8 == 0 and the
// FastFlags all set), so the element-wise fallbacks are not generated
// The second output of the FFN-up product (the saved gradient factor) is read again only in the backward: it takes the non-temporal store path and
// does not evict operand panels from L2 (+0.6 % on the step; non-temporal LOADS of the saved factor / the fp32 residual were 10-20 % slower -
// those operands are cache hits; profiles/r04_nontemporal_ab.txt).  -DAV_EPI_NT=0: A/B builds.
#ifndef AV_EPI_NT
#define AV_EPI_NT 1
#endif
template <bool VO>
__device__ __forceinline__ void epilogue_store_t(const av_gemm_args& p, const FastFlags& fl0, float (&v)[8], long long off, int gm, int gn, bool full0,
                                                 const float* R) {
    const bool full = VO ? true : full0;
    FastFlags fl = fl0;
    if constexpr (VO) { fl.c_vec = 1; fl.r_vec = 1; fl.aux_vec = 1; }
    if (p.act == AV_ACT_GELU_GF) {
        // activation + dropout site whose backward is a plain multiply: C2 = gelu'(v) * m, v = gelu(v) * m (m = dropout multiplier); the dX
        // product of the layer above then ends in AV_ACT_MUL_AUX - no erf / exp / Philox while its matrix pipe waits
        float m[8], gf[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) m[e] = 1.f;
        if (p.drop_p > 0.f) {
            const float ik = drop_inv_keep(p.drop_p);
            if (VO || (off & 3) == 0) {                       // VO: chunks are 16-B aligned (8-element offsets) by contract
                float m4[4];
                drop_mult4(p.drop_seed, p.drop_stream, (unsigned long long)off, p.drop_p, ik, m4);
#pragma unroll
                for (int e = 0; e < 4; ++e) m[e] = m4[e];
                drop_mult4(p.drop_seed, p.drop_stream, (unsigned long long)off + 4, p.drop_p, ik, m4);
#pragma unroll
                for (int e = 0; e < 4; ++e) m[4 + e] = m4[e];
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) m[e] = drop_mult(p.drop_seed, p.drop_stream, (unsigned long long)(off + e), p.drop_p, ik);
            }
        }
        {
            av_f32x2 x4[4], gl4[4], gp4[4];                     // the four pairs side by side (av_common.h: gelu_both_fast2x4)
#pragma unroll
            for (int k = 0; k < 4; ++k) x4[k] = av_f32x2{v[2 * k], v[2 * k + 1]};
            gelu_both_fast2x4(x4, gl4, gp4);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const av_f32x2 mm = av_f32x2{m[2 * k], m[2 * k + 1]};
                const av_f32x2 gl = gl4[k] * mm, gp = gp4[k] * mm;
                v[2 * k] = gl.x; v[2 * k + 1] = gl.y; gf[2 * k] = gp.x; gf[2 * k + 1] = gp.y;
            }
        }
        if (p.C2) {
            if (full && fl.c_vec && p.out_dtype == AV_BF16) {
                bf16x8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = (bf16_t)gf[e];
                {
#if AV_EPI_NT        // the gradient factor is read again only in the backward: non-temporal store (it does not push operand panels out of L2)
                    __builtin_nontemporal_store(o, (bf16x8*)((bf16_t*)p.C2 + off));
#else
                    *(bf16x8*)((bf16_t*)p.C2 + off) = o;
#endif
                }
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) if (gn + e < p.N) st_any(p.C2, off + e, p.out_dtype, gf[e]);
            }
        }
    } else
    if (p.C2) {
        if (full && fl.c_vec) {
            if (p.out_dtype == AV_BF16) {
                bf16x8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = (bf16_t)v[e];
                *(bf16x8*)((bf16_t*)p.C2 + off) = o;
            } else {
                *(f32x4*)((float*)p.C2 + off) = f32x4{v[0], v[1], v[2], v[3]};
                *(f32x4*)((float*)p.C2 + off + 4) = f32x4{v[4], v[5], v[6], v[7]};
            }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) if (gn + e < p.N) st_any(p.C2, off + e, p.out_dtype, v[e]);
        }
    }
    if (p.act == AV_ACT_GELU) {
        av_f32x2 x4[4], g4[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) x4[k] = av_f32x2{v[2 * k], v[2 * k + 1]};
        gelu_fast2x4(x4, g4);
#pragma unroll
        for (int k = 0; k < 4; ++k) { v[2 * k] = g4[k].x; v[2 * k + 1] = g4[k].y; }
    } else if (p.act == AV_ACT_MUL_AUX) {
        if (full && fl.aux_vec && p.aux_dtype == AV_BF16) {
            const bf16x8 u = *(const bf16x8*)((const bf16_t*)p.aux + off);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] *= (float)u[e];
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) if (gn + e < p.N) v[e] *= ld_any(p.aux, off + e, p.aux_dtype);
        }
    } else if (p.act == AV_ACT_MUL_GELU_GRAD) {
        if (full && fl.aux_vec && p.aux_dtype == AV_BF16) {
            const bf16x8 u = *(const bf16x8*)((const bf16_t*)p.aux + off);
#pragma unroll
            for (int e = 0; e < 8; e += 2) { const av_f32x2 g = gelu_grad_fast2(av_f32x2{(float)u[e], (float)u[e + 1]}); v[e] *= g.x; v[e + 1] *= g.y; }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) if (gn + e < p.N) v[e] *= gelu_grad_fast(ld_any(p.aux, off + e, p.aux_dtype));
        }
    }
    if (p.drop_p > 0.f && p.act != AV_ACT_GELU_GF) {
        const float ik = drop_inv_keep(p.drop_p);
        if (VO || (off & 3) == 0) {
            float m4[4];
            drop_mult4(p.drop_seed, p.drop_stream, (unsigned long long)off, p.drop_p, ik, m4);
#pragma unroll
            for (int e = 0; e < 4; e += 2) { const av_f32x2 t = av_f32x2{v[e], v[e + 1]} * av_f32x2{m4[e], m4[e + 1]}; v[e] = t.x; v[e + 1] = t.y; }
            drop_mult4(p.drop_seed, p.drop_stream, (unsigned long long)off + 4, p.drop_p, ik, m4);
#pragma unroll
            for (int e = 0; e < 4; e += 2) { const av_f32x2 t = av_f32x2{v[4 + e], v[5 + e]} * av_f32x2{m4[e], m4[e + 1]}; v[4 + e] = t.x; v[5 + e] = t.y; }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] *= drop_mult(p.drop_seed, p.drop_stream, (unsigned long long)(off + e), p.drop_p, ik);
        }
    }
    if (R) {
        const long long roff = (long long)gm * p.ldr + gn;
        if (full && fl.r_vec) {
            const f32x4 r0 = *(const f32x4*)(R + roff), r1 = *(const f32x4*)(R + roff + 4);
#pragma unroll
            for (int e = 0; e < 4; e += 2) {
                const av_f32x2 a = av_f32x2{v[e], v[e + 1]} + av_f32x2{r0[e], r0[e + 1]}, b = av_f32x2{v[4 + e], v[5 + e]} + av_f32x2{r1[e], r1[e + 1]};
                v[e] = a.x; v[e + 1] = a.y; v[4 + e] = b.x; v[5 + e] = b.y;
            }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) if (gn + e < p.N) v[e] += R[roff + e];
        }
    }
    if (full && fl.c_vec) {
        if (p.out_dtype == AV_BF16) {
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (bf16_t)v[e];
            *(bf16x8*)((bf16_t*)p.C + off) = o;
        } else {
            *(f32x4*)((float*)p.C + off) = f32x4{v[0], v[1], v[2], v[3]};
            *(f32x4*)((float*)p.C + off + 4) = f32x4{v[4], v[5], v[6], v[7]};
        }
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) if (gn + e < p.N) st_any(p.C, off + e, p.out_dtype, v[e]);
    }
}

__device__ __forceinline__ void epilogue_store(const av_gemm_args& p, const FastFlags& fl, float (&v)[8], long long off, int gm, int gn, bool full,
                                               const float* R) {
    epilogue_store_t<false>(p, fl, v, off, gm, gn, full, R);
}


// BNT = 128: waves 2(M) x 2(N), 64 x 64 each.   BNT = 64: waves 4(M) x 1(N), 32 x 64 each (N <= 64 problems:
// ResNet layer1, grouped positional conv).   CONV: A operand is the implicit im2col of an NHWC image.
// AKM / BKM: that operand is k-major (A stored [K][M], B stored [K][N]); BNT = 128, no CONV.
template <int BNT, bool CONV, bool AKM = false, bool BKM = false>
__global__ __launch_bounds__(NT, 2) void gemm_nt_bf16_kernel(const av_gemm_args p, const int nbM, const int nbN, const FastFlags fl) {
    constexpr int WM_T = BNT == 128 ? 4 : 2;                 // m-tiles per wave
    constexpr int TILE_BB = BNT * BK * 2;                    // B tile bytes
    constexpr int STAGE = TILE_A + TILE_BB;
    constexpr int CLD = BNT + 4;                             // fp32 epilogue image leading dimension
    constexpr int EPI_BYTES = BM * CLD * 4;                  // the BatchNorm partial scratch (2 KiB) sits behind the image
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wrow = BNT == 128 ? (w >> 1) * 64 : w * 32;
    const int wcol = BNT == 128 ? (w & 1) * 64 : 0;
    const int r = lane & 15, g = lane >> 4;

    // XCD-aware bijective remap: consecutive tile ids (sharing an A panel) land on the same XCD
    const int nwg = nbM * nbN;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, rem = nwg & 7, xcd = bid & 7, slot = bid >> 3;
        bid = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + slot;
    }
    // grouped ordering inside the XCD's contiguous id range: 8 row panels x all column panels, column-panel major,
    // so that the 8 A panels AND each B panel are re-read from this XCD's L2 instead of HBM / Infinity Cache
    constexpr int GM = 8;
    const int per_group = GM * nbN;
    const int grp = bid / per_group, in_grp = bid - grp * per_group;
    const int first_m = grp * GM;
    const int gsz = nbM - first_m < GM ? nbM - first_m : GM;
    const int mb = first_m + in_grp % gsz, nb = in_grp / gsz;
    const int m0 = mb * BM, n0 = nb * BNT;
    const int z = blockIdx.z;
    const int zo = p.batch_inner > 0 ? z / p.batch_inner : 0;
    const int zi = p.batch_inner > 0 ? z % p.batch_inner : z;
    const bf16_t* A = (const bf16_t*)p.A + (long long)zo * p.oA + (long long)zi * p.sA;
    const bf16_t* B = (const bf16_t*)p.B + (long long)zo * p.oB + (long long)zi * p.sB;

    ConvRows cr;
    if constexpr (CONV) conv_rows_init(cr, A, p, m0, w, lane);
    int Kz = p.K;                                            // split-K: this batch's slice of k_total (the last one may be shorter)
    if constexpr (AKM && BKM) {
        if (p.k_total > 0 && p.k_total - z * p.K < Kz) Kz = p.k_total - z * p.K;
    }

    f32x4 acc[WM_T][4];
#pragma unroll
    for (int i = 0; i < WM_T; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    TapSeq ts;
    ts.seq = 0ull; ts.ntap = -1; ts.sh = 0;
    if constexpr (CONV) { if (p.cPM) ts = tap_seq(p, m0, BM); }
    auto stage = [&](int kt, char* buf) {
        int k0 = kt * BK;
        if constexpr (CONV) { k0 = tap_k0(p, ts, kt); stage_conv(p, cr, k0, buf, w, lane); }
        else if constexpr (AKM) stage_kmajor(A, p.lda, m0, p.M, kt * BK, Kz, buf, w, lane);
        else stage_rows<4>(A, p.lda, m0, p.M, kt * BK, buf, w, lane);
        if constexpr (BKM) stage_kmajor(B, p.ldb, n0, p.N, kt * BK, Kz, buf + TILE_A, w, lane);
        else stage_rows<BNT / 32>(B, p.ldb, n0, p.N, k0, buf + TILE_A, w, lane);
    };

    int nk = (Kz + BK - 1) / BK;                             // a ragged last K-step only with k-major operands (rows >= K are zero lines)
    if constexpr (CONV) { if (ts.ntap >= 0) nk = ts.ntap << ts.sh; }
    KmOff ao[WM_T], bo[4];
    if constexpr (AKM) {
#pragma unroll
        for (int i = 0; i < WM_T; ++i) ao[i] = kmajor_off(wrow / 16 + i, lane);
    }
    if constexpr (BKM) {
#pragma unroll
        for (int j = 0; j < 4; ++j) bo[j] = kmajor_off(wcol / 16 + j, lane);
    }
    stage(0, smem);
    __syncthreads();

    // per-lane fragment addressing: row = base + 16*t + r (row & 7 == r & 7), chunk = 4*ks + g
    const int sw = r & 7;
    const char* a_base = smem + (wrow + r) * 128;
    const char* b_base = smem + TILE_A + (wcol + r) * 128;
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) stage(kt + 1, smem + (cur ^ 1) * STAGE);
        const char* ab = a_base + cur * STAGE;
        const char* bb = b_base + cur * STAGE;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int choff = ((ks * 4 + g) ^ sw) << 4;
            bf16x8 a[WM_T], b[4];
            bf16x4 ah[4][2], bh[4][2];                           // halves of the k-major fragments until the wait below
#pragma unroll
            for (int i = 0; i < WM_T; ++i) {
                if constexpr (AKM) kmajor_read(smem + cur * STAGE, ao[i], ks, ah[i]);
                else a[i] = *(const bf16x8*)(ab + i * 16 * 128 + choff);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if constexpr (BKM) kmajor_read(smem + cur * STAGE + TILE_A, bo[j], ks, bh[j]);
                else b[j] = *(const bf16x8*)(bb + j * 16 * 128 + choff);
            }
            if constexpr (AKM || BKM) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                 // the asm transposed reads are not tracked by the compiler
                if constexpr (AKM) {
                    static_assert(WM_T == 4, "k-major A: 128-wide tiles");
                    kmajor_tie4(ah);
#pragma unroll
                    for (int i = 0; i < WM_T; ++i) a[i] = __builtin_shufflevector(ah[i][0], ah[i][1], 0, 1, 2, 3, 4, 5, 6, 7);
                }
                if constexpr (BKM) {
                    kmajor_tie4(bh);
#pragma unroll
                    for (int j = 0; j < 4; ++j) b[j] = __builtin_shufflevector(bh[j][0], bh[j][1], 0, 1, 2, 3, 4, 5, 6, 7);
                }
            }
#pragma unroll
            for (int i = 0; i < WM_T; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = AV_MFMA_F32_16X16X32_LP(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }

    // ---------------- epilogue: accumulators -> fp32 LDS image -> row-contiguous vector stores ----------------
    float* cs = (float*)smem;
#pragma unroll
    for (int i = 0; i < WM_T; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e)
                cs[(wrow + i * 16 + 4 * g + e) * CLD + wcol + j * 16 + r] = acc[i][j][e] * p.alpha;
    if (p.stats) {
        // train-mode BatchNorm partials (per-column sum / sum of squares over this block's rows) straight from the accumulators: rows
        // beyond M were staged from the zero line, so they contribute exact zeros.  Lane (r, g) sums its 4 x WM_T rows of column
        // wcol + 16 j + r, the four g groups meet through two shuffles, the wavefronts of a column range through a 2-KiB LDS scratch.
        float* red = (float*)(smem + EPI_BYTES);
        const int wslot = BNT == 128 ? (w >> 1) : w;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int i = 0; i < WM_T; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float v = acc[i][j][e] * p.alpha; s1 += v; s2 += v * v; }
            s1 += __shfl_xor(s1, 16, 64); s1 += __shfl_xor(s1, 32, 64);
            s2 += __shfl_xor(s2, 16, 64); s2 += __shfl_xor(s2, 32, 64);
            if (g == 0) {
                red[(wslot * 2) * BNT + wcol + j * 16 + r] = s1;
                red[(wslot * 2 + 1) * BNT + wcol + j * 16 + r] = s2;
            }
        }
    }
    __syncthreads();

    const long long cbase = (long long)zo * p.oC + (long long)zi * p.sC;
    const float* R = p.R ? p.R + (long long)zi * p.sR : nullptr;
    const float* bias = p.bias ? p.bias + (long long)zi * p.sBias : nullptr;
    constexpr int CPR = BNT / 8;                             // 8-column chunks per tile row
    float bv[8];                                             // the thread's 8 columns are the same in every iteration
    {
        const int gnt = n0 + (tid % CPR) * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) bv[e] = (bias && gnt + e < p.N) ? bias[gnt + e] : 0.f;
    }
    // Fast tails: a store whose data registers are reused by the next iteration's LDS reads makes the compiler drain vmcnt(0)
    // in between, i.e. one store round trip per iteration.  For the common epilogues all rows are therefore produced first
    // (distinct registers) and stored together.
    const bool fast_tail = n0 + BNT <= p.N && fl.c_vec && !p.C2 && p.act == AV_ACT_NONE && p.drop_p <= 0.f && !R && p.out_dtype == AV_BF16;
    if (fast_tail) {
        constexpr int NIT = BM * CPR / NT;
        uint4 ov[NIT];
        const int cc = (tid % CPR) * 8;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int row = (it * NT + tid) / CPR;
            const f32x4 v0 = *(const f32x4*)(cs + row * CLD + cc), v1 = *(const f32x4*)(cs + row * CLD + cc + 4);
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) { o[e] = (bf16_t)(v0[e] + bv[e]); o[4 + e] = (bf16_t)(v1[e] + bv[4 + e]); }
            ov[it] = __builtin_bit_cast(uint4, o);
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) asm volatile("" :: "v"(ov[it].x), "v"(ov[it].y), "v"(ov[it].z), "v"(ov[it].w));
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int row = (it * NT + tid) / CPR;
            if (m0 + row < p.M) *(uint4*)((bf16_t*)p.C + cbase + (long long)(m0 + row) * p.ldc + n0 + cc) = ov[it];
        }
    }
    if (!fast_tail)
#pragma unroll
    for (int it = 0; it < BM * CPR / NT; ++it) {
        const int id = it * NT + tid;
        const int row = id / CPR, cc = (id % CPR) * 8;
        const int gm = m0 + row, gn = n0 + cc;
        if (gm >= p.M || gn >= p.N) continue;
        float v[8];
        const f32x4 v0 = *(const f32x4*)(cs + row * CLD + cc), v1 = *(const f32x4*)(cs + row * CLD + cc + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] = v0[e]; v[4 + e] = v1[e]; }
        const bool full = gn + 8 <= p.N;
        const long long off = cbase + (long long)gm * p.ldc + gn;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += bv[e];
        if (p.act == AV_ACT_GELU_GF) {
            // activation + dropout site whose backward is a plain multiply: C2 = gelu'(v) * m, v = gelu(v) * m (m = dropout multiplier); the dX
            // product of the layer above then ends in AV_ACT_MUL_AUX - no erf / exp / Philox while its matrix pipe waits
            float m[8], gf[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) m[e] = 1.f;
            if (p.drop_p > 0.f) {
                const float ik = drop_inv_keep(p.drop_p);
                if ((off & 3) == 0) {
                    float m4[4];
                    drop_mult4(p.drop_seed, p.drop_stream, (unsigned long long)off, p.drop_p, ik, m4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) m[e] = m4[e];
                    drop_mult4(p.drop_seed, p.drop_stream, (unsigned long long)off + 4, p.drop_p, ik, m4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) m[4 + e] = m4[e];
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) m[e] = drop_mult(p.drop_seed, p.drop_stream, (unsigned long long)(off + e), p.drop_p, ik);
                }
            }
#pragma unroll
            for (int e = 0; e < 8; e += 2) {
                av_f32x2 gl, gp;
                const av_f32x2 mm = av_f32x2{m[e], m[e + 1]};
                gelu_both_fast2(av_f32x2{v[e], v[e + 1]}, gl, gp);
                gl = gl * mm; gp = gp * mm;
                v[e] = gl.x; v[e + 1] = gl.y; gf[e] = gp.x; gf[e + 1] = gp.y;
            }
            if (p.C2) {
                if (full && fl.c_vec && p.out_dtype == AV_BF16) {
                    bf16x8 o;
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = (bf16_t)gf[e];
                    *(bf16x8*)((bf16_t*)p.C2 + off) = o;
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) if (gn + e < p.N) st_any(p.C2, off + e, p.out_dtype, gf[e]);
                }
            }
        } else
        if (p.C2) {
            if (full && fl.c_vec) {
                if (p.out_dtype == AV_BF16) {
                    bf16x8 o;
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = (bf16_t)v[e];
                    *(bf16x8*)((bf16_t*)p.C2 + off) = o;
                } else {
                    *(f32x4*)((float*)p.C2 + off) = f32x4{v[0], v[1], v[2], v[3]};
                    *(f32x4*)((float*)p.C2 + off + 4) = f32x4{v[4], v[5], v[6], v[7]};
                }
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) if (gn + e < p.N) st_any(p.C2, off + e, p.out_dtype, v[e]);
            }
        }
        if (p.act == AV_ACT_GELU) {
#pragma unroll
            for (int e = 0; e < 8; e += 2) { const av_f32x2 g = gelu_fast2(av_f32x2{v[e], v[e + 1]}); v[e] = g.x; v[e + 1] = g.y; }
        } else if (p.act == AV_ACT_MUL_AUX) {
            if (full && fl.aux_vec && p.aux_dtype == AV_BF16) {
                const bf16x8 u = *(const bf16x8*)((const bf16_t*)p.aux + off);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] *= (float)u[e];
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) if (gn + e < p.N) v[e] *= ld_any(p.aux, off + e, p.aux_dtype);
            }
        } else if (p.act == AV_ACT_MUL_GELU_GRAD) {
            if (full && fl.aux_vec && p.aux_dtype == AV_BF16) {
                const bf16x8 u = *(const bf16x8*)((const bf16_t*)p.aux + off);
#pragma unroll
                for (int e = 0; e < 8; e += 2) { const av_f32x2 g = gelu_grad_fast2(av_f32x2{(float)u[e], (float)u[e + 1]}); v[e] *= g.x; v[e + 1] *= g.y; }
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) if (gn + e < p.N) v[e] *= gelu_grad_fast(ld_any(p.aux, off + e, p.aux_dtype));
            }
        }
        if (p.drop_p > 0.f && p.act != AV_ACT_GELU_GF) {        // off is a multiple of 4 whenever ldc % 4 == 0 (gn % 8 == 0)
            const float ik = drop_inv_keep(p.drop_p);
            if ((off & 3) == 0) {
                float m4[4];
                drop_mult4(p.drop_seed, p.drop_stream, (unsigned long long)off, p.drop_p, ik, m4);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] *= m4[e];
                drop_mult4(p.drop_seed, p.drop_stream, (unsigned long long)off + 4, p.drop_p, ik, m4);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[4 + e] *= m4[e];
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] *= drop_mult(p.drop_seed, p.drop_stream, (unsigned long long)(off + e), p.drop_p, ik);
            }
        }
        if (R) {
            const long long roff = (long long)gm * p.ldr + gn;
            if (full && fl.r_vec) {
                const f32x4 r0 = *(const f32x4*)(R + roff), r1 = *(const f32x4*)(R + roff + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[e] += r0[e]; v[4 + e] += r1[e]; }
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) if (gn + e < p.N) v[e] += R[roff + e];
            }
        }
        if (full && fl.c_vec) {
            if (p.out_dtype == AV_BF16) {
                bf16x8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = (bf16_t)v[e];
                *(bf16x8*)((bf16_t*)p.C + off) = o;
            } else {
                *(f32x4*)((float*)p.C + off) = f32x4{v[0], v[1], v[2], v[3]};
                *(f32x4*)((float*)p.C + off + 4) = f32x4{v[4], v[5], v[6], v[7]};
            }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) if (gn + e < p.N) st_any(p.C, off + e, p.out_dtype, v[e]);
        }
    }
    if (p.stats) {
        // combine the two wavefronts that share a column range (partials were left in the scratch behind the image by the block above)
        if (tid < BNT && n0 + tid < p.N) {
            const float* red = (const float*)(smem + EPI_BYTES);
            constexpr int NR = BNT == 128 ? 2 : 4;            // wavefronts per column range
            float t1 = 0.f, t2 = 0.f;
#pragma unroll
            for (int q = 0; q < NR; ++q) { t1 += red[(q * 2) * BNT + tid]; t2 += red[(q * 2 + 1) * BNT + tid]; }
            float* o = p.stats + (long long)mb * 2 * p.N;
            o[n0 + tid] = t1;
            o[p.N + n0 + tid] = t2;
        }
    }
}

// ---- 2-D transpose with optional zero padding of the new inner dimension (out[C][Rpad]) -------------------
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void transpose_kernel(const TI* __restrict__ in, TO* __restrict__ out, int R, int C, long long ldi,
                                                        int Rpad) {
    __shared__ float tile[64][65];
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int i = ty; i < 64; i += 4) {
        const int rr = r0 + i, cc = c0 + tx;
        tile[i][tx] = (rr < R && cc < C) ? to_f32<TI>(in[(long long)rr * ldi + cc]) : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 64; i += 4) {
        const int cc = c0 + i, rr = r0 + tx;
        if (cc < C && rr < Rpad) out[(long long)cc * Rpad + rr] = from_f32<TO>(tile[tx][i]);
    }
}

bool al16(const void* p) { return ((uintptr_t)p % 16) == 0; }

}  // namespace

// ---------------------------------------------------------------------------------------------------------------
// v2: 256 x 128 x 64 tile, 512 threads = 8 wavefronts (4 x 2, 64 x 64 each), THREE LDS stages with the loads of K-tile
// t+2 in flight while tile t is multiplied: each wave retires only its OLDEST stage with a counted s_waitcnt vmcnt(N)
// (N = its LDS-DMA instructions of the newer stage), then ONE raw s_barrier per K-tile publishes the tile and frees the
// stage read in the previous iteration (read-after-wait-and-barrier / restage-after-barrier: guide §5 "Pipelining across
// barriers").  144 KiB of LDS => one workgroup per CU, two waves per SIMD.
// ---------------------------------------------------------------------------------------------------------------
constexpr int V2_BM = 256, V2_BN = 128, V2_NT = 512;
constexpr int V2_STAGE = (V2_BM + V2_BN) * BK * 2;          // 49 152 B
constexpr int V2_CLD = V2_BN + 4;
constexpr int V2_LDS = 3 * V2_STAGE;                        // 147 456 B >= epilogue image 256*132*4 = 135 168 B

__global__ __launch_bounds__(V2_NT, 2) void gemm_nt_bf16_v2_kernel(const av_gemm_args p, const int nbM, const int nbN, const FastFlags fl) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;       // w in 0..7
    const int wrow = (w >> 1) * 64, wcol = (w & 1) * 64;
    const int r = lane & 15, g = lane >> 4;
    const int nwg = nbM * nbN;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, rem = nwg & 7, xcd = bid & 7, slot = bid >> 3;
        bid = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + slot;
    }
    constexpr int GM = 4;
    const int per_group = GM * nbN;
    const int grp = bid / per_group, in_grp = bid - grp * per_group;
    const int first_m = grp * GM;
    const int gsz = nbM - first_m < GM ? nbM - first_m : GM;
    const int mb = first_m + in_grp % gsz, nb = in_grp / gsz;
    const int m0 = mb * V2_BM, n0 = nb * V2_BN;
    const int z = blockIdx.z;
    const int zo = p.batch_inner > 0 ? z / p.batch_inner : 0;
    const int zi = p.batch_inner > 0 ? z % p.batch_inner : z;
    const bf16_t* A = (const bf16_t*)p.A + (long long)zo * p.oA + (long long)zi * p.sA;
    const bf16_t* B = (const bf16_t*)p.B + (long long)zo * p.oB + (long long)zi * p.sB;

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // per wave and stage: 4 LDS-DMA instructions for A (256 rows / 8 waves / 8 rows) + 2 for B = 6
    auto stage = [&](int kt, char* buf) {
        stage_rows<4>(A, p.lda, m0, p.M, kt * BK, buf, w, lane);
        stage_rows<2>(B, p.ldb, n0, p.N, kt * BK, buf + V2_BM * BK * 2, w, lane);
    };
    const int nk = p.K / BK;
    stage(0, smem);
    if (nk > 1) stage(1, smem + V2_STAGE);

    const int sw = r & 7;
    const int a_off = (wrow + r) * 128, b_off = V2_BM * BK * 2 + (wcol + r) * 128;
    const int ch0 = ((0 * 4 + g) ^ sw) << 4, ch1 = ((1 * 4 + g) ^ sw) << 4;
    // fragment double buffering: F0 = k-step 0, F1 = k-step 1 of a K-tile.  Every ds_read group is issued one MFMA
    // group (16 MFMAs) ahead of its use, the tile hand-over barrier sits between the two MFMA groups.
    bf16x8 a0[4], b0[4], a1[4], b1[4];
#define LOADF(A_, B_, STG, CH)                                                                 \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) A_[i] = *(const bf16x8*)(smem + (STG) * V2_STAGE + a_off + i * 16 * 128 + (CH)); \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) B_[j] = *(const bf16x8*)(smem + (STG) * V2_STAGE + b_off + j * 16 * 128 + (CH));
#define MMAF(A_, B_)                                                                           \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                              \
        _Pragma("unroll") for (int j = 0; j < 4; ++j)                                          \
            acc[i][j] = AV_MFMA_F32_16X16X32_LP(A_[i], B_[j], acc[i][j], 0, 0, 0);

    if (nk > 1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                           // tile 0 landed for everyone
    LOADF(a0, b0, 0, ch0)
    int cur = 0;
    for (int kt = 0; kt + 1 < nk; ++kt) {                                   // all tiles but the last (peeled: no join on the wait state)
        if (kt + 2 < nk) {                                                  // stage (kt+2)%3 was released by the previous barrier
            int nxt = cur + 2; if (nxt >= 3) nxt -= 3;
            stage(kt + 2, smem + nxt * V2_STAGE);
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);                                 // lgkmcnt(0): F0 (issued one MFMA group ago) is complete
        LOADF(a1, b1, cur, ch1)                                             // in flight under the MFMAs of F0
        __builtin_amdgcn_sched_barrier(0);
        MMAF(a0, b0)
        __builtin_amdgcn_sched_barrier(0);
        const int nstage = cur + 1 == 3 ? 0 : cur + 1;
        __builtin_amdgcn_s_waitcnt(0xC07F);                                 // lgkmcnt(0): my reads of tile kt are complete
        if (kt + 2 < nk) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");   // my loads of tile kt+1 landed (tile kt+2 may fly)
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                       // tile kt+1 complete for everyone; stage cur is free
        LOADF(a0, b0, nstage, ch0)                                          // in flight under the MFMAs of F1
        __builtin_amdgcn_sched_barrier(0);
        MMAF(a1, b1)
        __builtin_amdgcn_sched_barrier(0);
        cur = nstage;
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);                                     // last tile
    LOADF(a1, b1, cur, ch1)
    __builtin_amdgcn_sched_barrier(0);
    MMAF(a0, b0)
    __builtin_amdgcn_sched_barrier(0);
    MMAF(a1, b1)
#undef LOADF
#undef MMAF
    __syncthreads();                                         // every wave is done reading the last stage

    float* cs = (float*)smem;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e)
                cs[(wrow + i * 16 + 4 * g + e) * V2_CLD + wcol + j * 16 + r] = acc[i][j][e] * p.alpha;
    __syncthreads();

    const long long cbase = (long long)zo * p.oC + (long long)zi * p.sC;
    const float* R = p.R ? p.R + (long long)zi * p.sR : nullptr;
    const float* bias = p.bias ? p.bias + (long long)zi * p.sBias : nullptr;
    constexpr int CPR = V2_BN / 8;
    float bv[8];
    {
        const int gnt = n0 + (tid % CPR) * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) bv[e] = (bias && gnt + e < p.N) ? bias[gnt + e] : 0.f;
    }
#pragma unroll
    for (int it = 0; it < V2_BM * CPR / V2_NT; ++it) {
        const int id = it * V2_NT + tid;
        const int row = id / CPR, cc = (id % CPR) * 8;
        const int gm = m0 + row, gn = n0 + cc;
        if (gm >= p.M || gn >= p.N) continue;
        float v[8];
        const f32x4 v0 = *(const f32x4*)(cs + row * V2_CLD + cc), v1 = *(const f32x4*)(cs + row * V2_CLD + cc + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] = v0[e] + bv[e]; v[4 + e] = v1[e] + bv[4 + e]; }
        const bool full = gn + 8 <= p.N;
        const long long off = cbase + (long long)gm * p.ldc + gn;
        epilogue_store(p, fl, v, off, gm, gn, full, R);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// v4: 256 x 256 x 64 tile with HALF-TILE granularity (the guide's "256^2 8-phase" schedule): 8 wavefronts as 2 (M) x 4 (N); the block
// tile is four 128 x 128 quadrants (A halves 0/1 x B halves 0/1) and every wavefront owns a 64 x 32 piece of EACH quadrant
// (128 accumulator registers).  A K-tile is four phases, one quadrant each: (A0,B0) (A0,B1) (A1,B1) (A1,B0); a phase is
//     { fragment reads of the half-tile(s) the quadrant adds (12 / 4 / 8 / 4 ds_read_b128) ; 2 LDS-DMA instructions = ONE 16-KiB
//       half-tile of the prefetch ; s_barrier ; lgkmcnt(0) ; 16 MFMAs ; s_barrier }.
// LDS = 2 K-tiles x 4 half-tile slots (A0, B1, A1, B0 in issue order) = 128 KiB.  The DMA runs 6 half-tiles ahead of the phase that
// issues it: phase p of K-tile t issues half-tile 4t+p+6, i.e. (t+1: A1), (t+1: B0), (t+2: A0), (t+2: B1) - every slot is re-filled
// TWO phases after its last fragment read (A0: read in phase 0, B1: 1, A1: 2, B0: 0 and 3), which is what the staggered wavefront
// groups need (below), and ONE counted wait per K-tile (vmcnt(4) in phase 3: everything but the two youngest half-tiles has landed =
// all of K-tile t+1) keeps the queue from ever draining inside the loop.
// Stagger: wavefronts 4-7 run one barrier behind 0-3 (one extra s_barrier at entry, balanced at exit), so on every SIMD one wavefront
// is in its MFMA segment while its partner issues reads / DMA: the matrix pipe sees back-to-back 16-MFMA clusters.
// RAW: a half-tile is read at the earliest one phase after the counted wait + barrier that retired it.  WAR: the lagging group's
// reads of phase c complete (lgkmcnt(0)) inside the leading group's phase c+1 read segment, hence the two-phase refill distance.
// ---------------------------------------------------------------------------------------------------------------
constexpr int V4_BM = 256, V4_BN = 256, V4_NT = 512;
constexpr int V4_HALF = 128 * BK * 2;                       // 16 384 B: one half-tile slot (128 rows x 128 B)
constexpr int V4_KT = 4 * V4_HALF;                          // 65 536 B per K-tile
// Ring of V4_NS half-tile slots, LDS-DMA issued V4_NS - 2 half-tiles ahead of the one being consumed (the slot overwritten at phase p of K-tile
// t is always the one of half-tile 4 t + p - 2, for any ring size).  8 slots = two K-tiles (96 KB in flight at most); 10 slots = all 160 KB
// of the CU's LDS, 128 KB in flight: with every CU busy the L2 -> LDS stream (44 GB/s per CU) needs more than 96 KB outstanding to cover its
// latency.  -DAV_V4_NS=8 builds the old ring (A/B).
#ifndef AV_V4_NS
#define AV_V4_NS 10
#endif
constexpr int V4_NS = AV_V4_NS, V4_LEAD = V4_NS - 2;
static_assert(V4_NS == 8 || V4_NS == 10, "ring of 8 or 10 half-tile slots");
constexpr int V4_CLD = V4_BN + 4;
constexpr int V4_EPI = 128 * V4_CLD * 4;                    // 133 120 B
constexpr int V4_LDS = V4_NS * V4_HALF > V4_EPI + 4096 ? V4_NS * V4_HALF : V4_EPI + 4096;   // ring | epilogue image + BatchNorm partial scratch

// Tail jobs of the v4 grid: ONE 128 x 128 quadrant of a 256 x 256 tile per workgroup (same wavefront layout: 2 x 4, 64 x 32 each, 32
// accumulator registers).  When the tile count leaves a short last round (R = tiles mod 256 <= 128), those R tiles are cut into 4 R
// quadrant jobs that fill the chip instead of R of 256 CUs working a whole tile time.  A K-tile is ONE phase here: an A half-tile and
// a B half-tile per K-tile in a ring of four 32-KiB slots, refilled two phases after the last read (staggered groups, as above), the
// DMA two K-tiles ahead, vmcnt(4) per phase.
__device__ __forceinline__ void v4_quadrant_job(const av_gemm_args& p, const FastFlags& fl, char* smem, const bf16_t* A, const bf16_t* B,
                                                const int m0, const int n0, const int m_end, const long long cbase, const float* R, const float* bias) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wr = __builtin_amdgcn_readfirstlane(w >> 2), wc = w & 3;
    const int r = lane & 15, g = lane >> 4;
    f32x4 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int nk = p.K / BK;
    auto issue = [&](int t) {
        char* slot = smem + (t & 3) * (2 * V4_HALF);
        stage_rows<2>(A, p.lda, m0, p.M, t * BK, slot, w, lane);
        stage_rows<2>(B, p.ldb, n0, p.N, t * BK, slot + V4_HALF, w, lane);
    };
    issue(0);
    if (nk > 1) { issue(1); asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (wr == 1) __builtin_amdgcn_s_barrier();
    const int sw = r & 7;
    const int ch0 = (g ^ sw) << 4, ch1 = ((4 + g) ^ sw) << 4;
    const int a_row = (wr * 64 + r) * 128, b_row = V4_HALF + (wc * 32 + r) * 128;
    bf16x8 fa[4][2], fb[2][2];
    for (int t = 0; t < nk; ++t) {
        const char* kb = smem + (t & 3) * (2 * V4_HALF);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            fb[j][0] = *(const bf16x8*)(kb + b_row + j * 2048 + ch0);
            fb[j][1] = *(const bf16x8*)(kb + b_row + j * 2048 + ch1);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            fa[i][0] = *(const bf16x8*)(kb + a_row + i * 2048 + ch0);
            fa[i][1] = *(const bf16x8*)(kb + a_row + i * 2048 + ch1);
        }
        if (t + 2 < nk) { issue(t + 2); asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }     // K-tile t+1 has landed, t+2 may fly
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = AV_MFMA_F32_16X16X32_LP(fa[i][ks], fb[j][ks], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
    if (wr == 0) __builtin_amdgcn_s_barrier();

    constexpr int QLD = 128 + 4;
    float* cs = (float*)smem;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e)
                cs[(wr * 64 + i * 16 + 4 * g + e) * QLD + wc * 32 + j * 16 + r] = acc[i][j][e] * p.alpha;
    __syncthreads();
    constexpr int CPR = 128 / 8;
    float bv[8];
    {
        const int gnt = n0 + (tid % CPR) * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) bv[e] = (bias && gnt + e < p.N) ? bias[gnt + e] : 0.f;
    }
    for (int it = 0; it < 128 * CPR / V4_NT; ++it) {
        const int id = it * V4_NT + tid;
        const int row = id / CPR, cc = (id % CPR) * 8;
        const int gm = m0 + row, gn = n0 + cc;
        if (gm >= m_end || gn >= p.N) continue;
        float v[8];
        const f32x4 v0 = *(const f32x4*)(cs + row * QLD + cc), v1 = *(const f32x4*)(cs + row * QLD + cc + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] = v0[e] + bv[e]; v[4 + e] = v1[e] + bv[4 + e]; }
        const bool full = gn + 8 <= p.N;
        const long long off = cbase + (long long)gm * p.ldc + gn;
        epilogue_store(p, fl, v, off, gm, gn, full, R);
    }
}

// implicit im2col for the v4 kernel: a half-tile is 128 output pixels, a wavefront stages 2 row groups of 8 pixels per half
struct ConvRows2 { const bf16_t* rowp[2]; unsigned valid[2]; };
__device__ __forceinline__ void conv_rows2_init(ConvRows2& cr, const bf16_t* __restrict__ base, const av_gemm_args& p, int mbase, int w, int lane) {
    const int sub = lane >> 3, choff = ((lane & 7) ^ sub) << 3;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = mbase + (w * 2 + i) * 8 + sub;
        cr.rowp[i] = base;
        cr.valid[i] = 0u;
        if (m < p.M) {
            int q, oy, ox;
            conv_decode(p, m, q, oy, ox);
            const int iy0 = oy * p.cSh - p.cPh, ix0 = ox * p.cSw - p.cPw;
            cr.rowp[i] = base + conv_pixel(p, q, iy0, ix0) * p.cCtot + p.cCoff + choff;
            unsigned v = 0u;
            for (int ky = 0; ky < p.cKh; ++ky)
                for (int kx = 0; kx < p.cKw; ++kx)
                    if (iy0 + ky >= 0 && iy0 + ky < p.cH && ix0 + kx >= 0 && ix0 + kx < p.cW) v |= 1u << (ky * p.cKw + kx);
            cr.valid[i] = v;
        }
    }
}
// all_in: every row of the tile is a real pixel whose window contains this tap (a position-major tile after tap_seq dropped the taps outside the
// image for its position): no per-lane select in front of the LDS-DMA - in the staggered schedule the instructions between a phase's fragment
// reads and its barrier are on the critical path (see stage_kmajor8)
__device__ __forceinline__ void stage_conv2(const av_gemm_args& p, const ConvRows2& cr, int k0, char* tile, int w, int lane, bool all_in) {
    const int sub = lane >> 3, pch = lane & 7;
    const int tap = k0 / p.cCin, c0 = k0 - tap * p.cCin;
    const int ky = tap / p.cKw, kx = tap - ky * p.cKw;
    const long long toff = ((long long)ky * p.cW + kx) * ((p.cPM & 1) ? p.cNF : 1) * p.cCtot + c0;          // wave-uniform
    if (all_in) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const unsigned off = __builtin_amdgcn_readfirstlane((unsigned)((w * 2 + i) * 1024));
            __builtin_amdgcn_global_load_lds((gptr_t)(cr.rowp[i] + toff), (lptr_t)(tile + off), 16, 0, 0);
        }
        return;
    }
    const bf16_t* zl = (const bf16_t*)g_zero_line + ((pch ^ sub) << 3);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const bf16_t* src = ((cr.valid[i] >> tap) & 1u) ? cr.rowp[i] + toff : zl;
        const unsigned off = __builtin_amdgcn_readfirstlane((unsigned)((w * 2 + i) * 1024));
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(tile + off), 16, 0, 0);
    }
}

// KM: BOTH operands k-major (A stored [K][M], B stored [K][N]: dW = dY^T X with the tokens as K).  A half-tile slot then holds 64 k-rows x 128
// columns (the same 16 KiB), filled by stage_kmajor<2> and read with the transposed LDS reads of kmajor_frag (two ds_read_b64_tr_b16 per
// fragment); ring, phases, waits and epilogue are unchanged.  blockIdx.z is a K slice of k_total (split-K: av_gemm_args.k_total), the last
// one possibly shorter and ragged (k-rows beyond it come from the zero line).
template <bool CONV, bool KM = false>
__global__ __launch_bounds__(V4_NT, 2) void gemm_nt_bf16_v4_kernel(const av_gemm_args p, const int nbM, const int nbN, const FastFlags fl, const int nfull, const int bm_eff) {
    static_assert(!(CONV && KM), "k-major operands: plain products only");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    AV_STAMP(0);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;       // w in 0..7
    const int wr = __builtin_amdgcn_readfirstlane(w >> 2), wc = w & 3;
    const int r = lane & 15, g = lane >> 4;
    // jobs: blocks [0, nfull) are whole tiles, blocks nfull + 4 i + q the four quadrants of tile nfull + i (the short last round).
    // Each section gets the XCD-aware bijective remap of its own id range; tile ids are grouped 4 row panels x all column panels.
    const int ntile = nbM * nbN;
    int bid = blockIdx.x, quad = -1;
    if (bid < nfull) {
        const int q = nfull >> 3, rem = nfull & 7, xcd = bid & 7, slot = bid >> 3;
        bid = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + slot;
    } else {
        const int nq = 4 * (ntile - nfull);
        int j = bid - nfull;
        const int q = nq >> 3, rem = nq & 7, xcd = j & 7, slot = j >> 3;
        j = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + slot;
        bid = nfull + (j >> 2);
        quad = j & 3;
    }
    int z = blockIdx.z;
    if constexpr (KM) {
        // K slices ride in the 1-D grid (nfull = tiles x slices jobs): the XCD-aware remap above hands each XCD a contiguous range of
        // (slice, tile) jobs, i.e. tiles of ONE slice that share their operand panels in that XCD's L2 (with the slices in blockIdx.z every
        // XCD worked on all slices at once: four times the L2-miss traffic)
        z = bid / ntile;
        bid -= z * ntile;
    }
    constexpr int GM = 4;
    const int per_group = GM * nbN;
    const int grp = bid / per_group, in_grp = bid - grp * per_group;
    const int first_m = grp * GM;
    const int gsz = nbM - first_m < GM ? nbM - first_m : GM;
    const int mb = first_m + in_grp % gsz, nb = in_grp / gsz;
    // Row tiles are bm_eff <= 256 rows apart (a multiple of 16): the tile still stages 256 rows, but 16-row tiles beyond bm_eff (or beyond M) get
    // neither fragment reads nor MFMAs nor stores.  The host picks bm_eff so that the tile count fills the 256 CUs (M = 64 x 199 tokens, N = 1024:
    // 62 x 4 = 248 tiles of 208 rows instead of 50 x 4 = 200 of 256); the two wavefront groups alternate on a SIMD, so the skipped MFMAs
    // of one group shorten the K-tile for both.
    const int m0 = mb * bm_eff, n0 = nb * V4_BN;
    const int rows_here = p.M - m0 < bm_eff ? p.M - m0 : bm_eff;
    const int zo = p.batch_inner > 0 ? z / p.batch_inner : 0;
    const int zi = p.batch_inner > 0 ? z % p.batch_inner : z;
    const bf16_t* A = (const bf16_t*)p.A + (long long)zo * p.oA + (long long)zi * p.sA;
    const bf16_t* B = (const bf16_t*)p.B + (long long)zo * p.oB + (long long)zi * p.sB;
    if (quad >= 0) {                                         // block-uniform
        const int qm = m0 + (quad >> 1) * 128, qn = n0 + (quad & 1) * 128;
        if (qm < m0 + rows_here && qn < p.N)
            v4_quadrant_job(p, fl, smem, A, B, qm, qn, m0 + rows_here, (long long)zo * p.oC + (long long)zi * p.sC,
                            p.R ? p.R + (long long)zi * p.sR : nullptr, p.bias ? p.bias + (long long)zi * p.sBias : nullptr);
        return;
    }
    // 16-row tiles of mine that hold rows of this tile: A half h starts at row 128 h + 64 wr (wave-uniform counts in 0..4)
    auto ntiles = [&](int start) { int n = (rows_here - start + 15) >> 4; return n < 0 ? 0 : (n > 4 ? 4 : n); };
    const int nmt0 = __builtin_amdgcn_readfirstlane(ntiles(wr * 64)), nmt1 = __builtin_amdgcn_readfirstlane(ntiles(128 + wr * 64));

    f32x4 acc[2][2][4][2];                                   // [A half][B half][m-tile][n-tile]
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[a][b][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    int nk = p.K / BK;
    int Kz = p.K;                                            // KM: this slice's share of k_total
    if constexpr (KM) {
        if (p.k_total > 0 && p.k_total - z * p.K < Kz) Kz = p.k_total - z * p.K;
        nk = (Kz + BK - 1) / BK;
    }
    ConvRows2 cr0, cr1;                                      // CONV: tap-0 pixel pointers + in-image tap masks of my rows of A0 / A1
    TapSeq ts;
    ts.seq = 0ull; ts.ntap = -1; ts.sh = 0;
    if constexpr (CONV) {
        conv_rows2_init(cr0, A, p, m0, w, lane); conv_rows2_init(cr1, A, p, m0 + 128, w, lane);
        if (p.cPM) { ts = tap_seq(p, m0, V4_BM); if (ts.ntap >= 0) nk = ts.ntap << ts.sh; }       // taps outside the image for this tile's position: skipped
    }
    // CONV: a whole tile of real rows at ONE output position whose out-of-image taps were dropped above => every staged row is in the image
    const bool conv_all_in = CONV && (p.cPM & 2) && ts.ntap >= 0 && m0 + V4_BM <= p.M && m0 / p.cNF == (m0 + V4_BM - 1) / p.cNF;
    KmLane kl[4];                                            // KM: per-lane source offsets of the four half-tile kinds (index = issue-order slot j)
    if constexpr (KM) {
        km_lane_init(kl[0], p.lda, m0, p.M, w, lane); km_lane_init(kl[2], p.lda, m0 + 128, p.M, w, lane);
        km_lane_init(kl[3], p.ldb, n0, p.N, w, lane); km_lane_init(kl[1], p.ldb, n0 + 128, p.N, w, lane);
    }
    const int nh = 4 * nk;                                   // half-tiles in issue order: slot j = s & 3 : 0 = A0, 1 = B1, 2 = A1, 3 = B0
    auto issue = [&](int t, int j, int sidx) {               // half-tile 4 t + j into ring slot sidx = (4 t + j) % V4_NS; j is a compile-time constant
#ifdef AV_ABL_NODMA
        return;
#endif
        char* slot = smem + sidx * V4_HALF;
        int k0 = t * BK;
#ifdef AV_ABL_KSCRAMBLE        // diagnostic (timing only): K-tiles in a scrambled order - no two consecutive K-tiles read adjacent lines of a row
        k0 = (int)(((long long)t * 37) % nk) * BK;
#endif
        if constexpr (CONV) k0 = tap_k0(p, ts, t);
        if constexpr (KM) {
            if (j == 0) stage_kmajor8(A, p.lda, kl[0], k0, Kz, slot, w, lane);
            else if (j == 2) stage_kmajor8(A, p.lda, kl[2], k0, Kz, slot, w, lane);
            else if (j == 3) stage_kmajor8(B, p.ldb, kl[3], k0, Kz, slot, w, lane);
            else stage_kmajor8(B, p.ldb, kl[1], k0, Kz, slot, w, lane);
            return;
        }
        if (j == 0) { if constexpr (CONV) stage_conv2(p, cr0, k0, slot, w, lane, conv_all_in); else stage_rows<2>(A, p.lda, m0, p.M, k0, slot, w, lane); }
        else if (j == 2) { if constexpr (CONV) stage_conv2(p, cr1, k0, slot, w, lane, conv_all_in); else stage_rows<2>(A, p.lda, m0 + 128, p.M, k0, slot, w, lane); }
        else if (j == 3) stage_rows<2>(B, p.ldb, n0, p.N, k0, slot, w, lane);
        else stage_rows<2>(B, p.ldb, n0 + 128, p.N, k0, slot, w, lane);
    };
    // prologue: half-tiles 0 .. V4_LEAD - 1 (K-tile 0 and the first half-tiles of the next ones); K-tile 0 must have landed
    issue(0, 0, 0); issue(0, 1, 1); issue(0, 2, 2); issue(0, 3, 3);
    if (nk > 1) {
        issue(1, 0, 4); issue(1, 1, 5);
        if constexpr (V4_LEAD == 8) { issue(1, 2, 6); issue(1, 3, 7); asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }
        else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                            // K-tile 0 landed for everyone
    asm volatile("" ::: "memory");
    if (wr == 1) __builtin_amdgcn_s_barrier();       // wavefronts 4-7 run one barrier behind
    AV_STAMP(1);

    const int sw = r & 7;
    const int ch0 = (g ^ sw) << 4, ch1 = ((4 + g) ^ sw) << 4;
    const int a_row = (wr * 64 + r) * 128, b_row = (wc * 32 + r) * 128;
    bf16x8 fa[4][2], fb[2][2];
#ifdef AV_ABL_NOREAD
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) { fa[i][ks] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0}; fb[i >> 1][ks] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0}; asm volatile("" : "+v"(fa[i][ks]), "+v"(fb[i >> 1][ks])); }
#endif
    bf16x8 fb0[2][2];                                        // KM: the B0 fragments of the current K-tile (phases 0 and 3)
    bf16x4 kfa[2][2][2][2], kfb[2][2][2];                    // KM: halves of the fragments in flight: [m-tile pair][parity][k-step][half], [n-tile][k-step][half]
    Km8Base kab, kbb;                                        // KM: transposed-read bases of my A column tiles 4 wr + i / B column tiles 2 wc + j
    const unsigned lds0 = (unsigned)(unsigned long long)(lds_b4_t)smem;
    if constexpr (KM) { km8_base_init(kab, 1024 * wr, lane); km8_base_init(kbb, 512 * wc, lane); }

// Diagnostic builds only (tools/gemm_ablate.py; never the product library): -DAV_ABL_NOMFMA keeps every LDS-DMA, fragment read, wait and
// barrier of the main loop but issues no MFMA (the fragments are kept alive), -DAV_ABL_NODMA keeps the MFMAs and reads but stages nothing.
// Timing the three builds on one shape says which resource paces the K-tile.
#ifdef AV_ABL_NOMFMA
#define V4_MFMA_OP(ACC, A, B) asm volatile("" :: "v"(A), "v"(B))
#else
#define V4_MFMA_OP(ACC, A, B) ACC = AV_MFMA_F32_16X16X32_LP(A, B, ACC, 0, 0, 0)
#endif
// (no per-m-tile skipping of short tiles: an `if (i < nmt)` inside these unrolled loops compiles to a scalar branch around every pair of MFMAs /
// fragment reads - a dozen taken branches inside a 16-MFMA cluster cost more than the MFMAs they save; rows past a tile's bm_eff rows are
// clamped re-reads of its last row, computed and never stored)
#ifdef AV_ABL_NOREAD
#define V4_ABL_NR 0
#else
#define V4_ABL_NR 4
#endif
#define V4_READ_A(SLOT, NMT)                                                                                       \
    if constexpr (KM) {                                                                                            \
        if (V4_ABL_NR) {                                                                                           \
            const unsigned sb_ = lds0 + sl[SLOT] * V4_HALF;                                                        \
            const unsigned a00_ = sb_ + kab.b[0][0], a01_ = sb_ + kab.b[0][1], a10_ = sb_ + kab.b[1][0], a11_ = sb_ + kab.b[1][1];    \
            km8_read<0>(a00_, a01_, kfa[0][0][0]);       km8_read<8192>(a00_, a01_, kfa[0][0][1]);                 \
            km8_read<0>(a10_, a11_, kfa[0][1][0]);       km8_read<8192>(a10_, a11_, kfa[0][1][1]);                 \
            km8_read<512>(a00_, a01_, kfa[1][0][0]);     km8_read<8192 + 512>(a00_, a01_, kfa[1][0][1]);           \
            km8_read<512>(a10_, a11_, kfa[1][1][0]);     km8_read<8192 + 512>(a10_, a11_, kfa[1][1][1]);           \
        }                                                                                                          \
    } else {                                                                                                       \
    _Pragma("unroll") for (int i = 0; i < V4_ABL_NR; ++i) {                                                        \
            fa[i][0] = *(const bf16x8*)(smem + sl[SLOT] * V4_HALF + a_row + i * 2048 + ch0);                       \
            fa[i][1] = *(const bf16x8*)(smem + sl[SLOT] * V4_HALF + a_row + i * 2048 + ch1); } }
#define V4_READ_B(SLOT, FB, KFB)                                                                                   \
    if constexpr (KM) {                                                                                            \
        if (V4_ABL_NR) {                                                                                           \
            const unsigned sb_ = lds0 + sl[SLOT] * V4_HALF;                                                        \
            const unsigned b00_ = sb_ + kbb.b[0][0], b01_ = sb_ + kbb.b[0][1], b10_ = sb_ + kbb.b[1][0], b11_ = sb_ + kbb.b[1][1];    \
            km8_read<0>(b00_, b01_, KFB[0][0]);          km8_read<8192>(b00_, b01_, KFB[0][1]);                    \
            km8_read<0>(b10_, b11_, KFB[1][0]);          km8_read<8192>(b10_, b11_, KFB[1][1]);                    \
        }                                                                                                          \
    } else {                                                                                                       \
    _Pragma("unroll") for (int j = 0; j < V4_ABL_NR / 2; ++j) {                                                    \
            FB[j][0] = *(const bf16x8*)(smem + sl[SLOT] * V4_HALF + b_row + j * 2048 + ch0);                       \
            FB[j][1] = *(const bf16x8*)(smem + sl[SLOT] * V4_HALF + b_row + j * 2048 + ch1); } }
#define V4_JOIN_A                                                                                                  \
    if constexpr (KM) { if (V4_ABL_NR) {                                                                           \
        km8_tie8(kfa[0]); km8_tie8(kfa[1]);                                                                        \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) { fa[i][0] = km8_join(kfa[i >> 1][i & 1][0]); fa[i][1] = km8_join(kfa[i >> 1][i & 1][1]); } } }
#define V4_JOIN_B(FB, KFB)                                                                                         \
    if constexpr (KM) { if (V4_ABL_NR) {                                                                           \
        km8_tie8(KFB);                                                                                             \
        _Pragma("unroll") for (int j = 0; j < 2; ++j) { FB[j][0] = km8_join(KFB[j][0]); FB[j][1] = km8_join(KFB[j][1]); } } }
#define V4_JOIN_NONE
#define V4_MMA(QA, QB, FB, JOIN)                                                                                   \
    __builtin_amdgcn_s_barrier();                                                                                  \
    __builtin_amdgcn_s_waitcnt(0xC07F);                                                                            \
    JOIN                                                                                                           \
    __builtin_amdgcn_sched_barrier(0);                                                                             \
    __builtin_amdgcn_s_setprio(1);                                                                                 \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                               \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                              \
            _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                          \
                V4_MFMA_OP(acc[QA][QB][i][j], fa[i][ks], FB[j][ks]);                                               \
    __builtin_amdgcn_s_setprio(0);                                                                                 \
    __builtin_amdgcn_sched_barrier(0);                                                                             \
    __builtin_amdgcn_s_barrier();                                                                                  \
    asm volatile("" ::: "memory");

    int b4 = 0;                                              // (4 t) % V4_NS: ring slot of half-tile (t, 0)
    for (int t = 0; t < nk; ++t) {
        int sl[4], si[4];                                    // slots of this K-tile's half-tiles; slots the four phases refill (half-tile 4 t + V4_LEAD + p)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int x = b4 + j; x = x >= V4_NS ? x - V4_NS : x; sl[j] = x;
            int y = b4 + j - 2; y = y < 0 ? y + V4_NS : y; si[j] = y;
        }
        b4 += 4; b4 = b4 >= V4_NS ? b4 - V4_NS : b4;
        // phase 0: quadrant (A0, B0)
        if constexpr (KM) { V4_READ_B(3, fb0, kfb) } else { V4_READ_B(3, fb, kfb) }      // k-major form: B0 stays in its own registers for phase 3 (8 fewer transposed reads per K-tile)
        __builtin_amdgcn_sched_barrier(0);
        V4_READ_A(0, nmt0)
        if (4 * t + V4_LEAD + 0 < nh) issue(t + (V4_LEAD + 0) / 4, (V4_LEAD + 0) % 4, si[0]);
        if constexpr (KM) { V4_MMA(0, 0, fb0, V4_JOIN_A V4_JOIN_B(fb0, kfb)) } else { V4_MMA(0, 0, fb, V4_JOIN_NONE) }
        // phase 1: (A0, B1)
        V4_READ_B(1, fb, kfb)
        if (4 * t + V4_LEAD + 1 < nh) issue(t + (V4_LEAD + 1) / 4, (V4_LEAD + 1) % 4, si[1]);
        V4_MMA(0, 1, fb, V4_JOIN_B(fb, kfb))
        // phase 2: (A1, B1)
        V4_READ_A(2, nmt1)
        if (4 * t + V4_LEAD + 2 < nh) issue(t + (V4_LEAD + 2) / 4, (V4_LEAD + 2) % 4, si[2]);
        V4_MMA(1, 1, fb, V4_JOIN_A)
        // phase 3: (A1, B0); the K-tile's one counted wait: all of K-tile t+1 has landed, the V4_LEAD - 4 youngest half-tiles (t+2) may fly
        if constexpr (!KM) { V4_READ_B(3, fb, kfb) }
        if (4 * t + V4_LEAD + 3 < nh) {
            issue(t + (V4_LEAD + 3) / 4, (V4_LEAD + 3) % 4, si[3]);
            if constexpr (V4_LEAD == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if constexpr (KM) { V4_MMA(1, 0, fb0, V4_JOIN_NONE) } else { V4_MMA(1, 0, fb, V4_JOIN_B(fb, kfb)) }
    }
#undef V4_READ_A
#undef V4_READ_B
#undef V4_MMA
#undef V4_JOIN_A
#undef V4_JOIN_B
#undef V4_JOIN_NONE
#undef V4_MFMA_OP
    if (wr == 0) __builtin_amdgcn_s_barrier();       // balance the entry barrier of wavefronts 4-7
    AV_STAMP(2);

    float* cs = (float*)smem;
    const long long cbase = (long long)zo * p.oC + (long long)zi * p.sC;
    const float* R = p.R ? p.R + (long long)zi * p.sR : nullptr;
    const float* bias = p.bias ? p.bias + (long long)zi * p.sBias : nullptr;
    constexpr int CPR = V4_BN / 8;                                          // 32 chunks of 8 columns per row
    float bv[8];
    {
        const int gnt = n0 + (tid % CPR) * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) bv[e] = (bias && gnt + e < p.N) ? bias[gnt + e] : 0.f;
    }
#pragma unroll
    for (int half = 0; half < 2; ++half) {                                  // rows half * 128 .. + 127 of the block tile: every wavefront owns 64 x 64 of them
        __syncthreads();                                                    // stages / previous half's image are free
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        cs[(wr * 64 + i * 16 + 4 * g + e) * V4_CLD + b * 128 + wc * 32 + j * 16 + r] = acc[half][b][i][j][e] * p.alpha;
        if (CONV && p.stats) {
            // train-mode BatchNorm partials from the accumulators (rows beyond M were staged from the zero line: exact zeros): lane (r, g)
            // sums its 16 rows of column b 128 + wc 32 + 16 j + r, two shuffles join the g groups, the two wavefront rows meet in LDS
            float* red = (float*)(smem + V4_EPI);
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    float s1 = 0.f, s2 = 0.f;
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int e = 0; e < 4; ++e) { const float v = acc[half][b][i][j][e] * p.alpha; s1 += v; s2 += v * v; }
                    s1 += __shfl_xor(s1, 16, 64); s1 += __shfl_xor(s1, 32, 64);
                    s2 += __shfl_xor(s2, 16, 64); s2 += __shfl_xor(s2, 32, 64);
                    if (g == 0) {
                        const int col = b * 128 + wc * 32 + j * 16 + r;
                        red[(wr * 2) * 256 + col] = s1;
                        red[(wr * 2 + 1) * 256 + col] = s2;
                    }
                }
        }
        __syncthreads();
        for (int it = 0; it < 128 * CPR / V4_NT; ++it) {
            const int id = it * V4_NT + tid;
            const int row = id / CPR, cc = (id % CPR) * 8;
            const int gm = m0 + half * 128 + row, gn = n0 + cc;
            if (half * 128 + row >= rows_here || gn >= p.N) continue;
            float v[8];
            const f32x4 v0 = *(const f32x4*)(cs + row * V4_CLD + cc), v1 = *(const f32x4*)(cs + row * V4_CLD + cc + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] = v0[e] + bv[e]; v[4 + e] = v1[e] + bv[4 + e]; }
            const bool full = gn + 8 <= p.N;
            const long long off = cbase + (long long)gm * p.ldc + gn;
            epilogue_store(p, fl, v, off, gm, gn, full, R);
        }
        if (CONV && p.stats && tid < 256 && n0 + tid < p.N && m0 + half * 128 < p.M) {
            // row block 2 mb + half of the caller's [ceil(M / 128)][2][N] buffer: the two wavefront rows' partials (scratch behind the image)
            const float* red = (const float*)(smem + V4_EPI);
            float* o = p.stats + (long long)(2 * mb + half) * 2 * p.N;
            o[n0 + tid] = red[tid] + red[512 + tid];
            o[p.N + n0 + tid] = red[256 + tid] + red[768 + tid];
        }
    }
#ifdef AV_GEMM_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // diagnostic: the exit stamp includes the completion of this workgroup's stores
#endif
    AV_STAMP(3);
}


// ---------------------------------------------------------------------------------------------------------------------------------------
// v7: the 8-phase kernel above as a PERSISTENT workgroup with a register-direct epilogue (plain NT products; not CONV).
//   * one workgroup per CU walks its list of 256 x 256 tiles (the same XCD-aware order the dispatcher gave the v4 grid: XCD label = block id
//     mod 8 owns a contiguous range of tile ids, its workgroups take every S-th of them); the short last round still runs as quadrant jobs;
//   * the MFMA operands are swapped (D = B-fragment x A-fragment): a lane then holds 4 CONSECUTIVE COLUMNS of one output row per accumulator
//     instead of 4 rows of one column, and the B half-tiles are staged with their rows permuted (LDS row [wc:2][j:1][g:2][e:2] of half b holds
//     tile column wc 64 + b 32 + g 8 + j 4 + e; the permutation sits on the LDS-DMA source address, fragment reads and their bank pattern are
//     unchanged) so that the two n-tiles of a lane are adjacent: a lane owns 8 consecutive columns = one 16-B (bf16) / two 16-B (fp32) stores,
//     a wavefront instruction writes 16 rows x 64 B and the two B halves complete the 128-B lines.  The epilogue therefore needs NO LDS image
//     and no barrier: bias / activation / dropout / residual run on the accumulator registers (same operation order per element as the
//     image epilogue: results are bit-identical to the v4 kernel);
//   * with the ring free at the end of the main loop, the first V4_LEAD half-tiles of the workgroup's NEXT tile are requested before the
//     epilogue of the current one and land while it computes and stores: a tile no longer starts with an empty ring (v4: 1.8 us entry +
//     cold first K-tiles per tile), and the stores of tile i drain under the main loop of tile i + 1 (vmcnt counts loads, LDS-DMA and stores
//     in one in-order counter: every counted wait of the loop only ever waits for MORE than it needs, never less).
// ---------------------------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int v7_bcol(int rho, int b) {    // LDS row of B half-tile b -> column of the 256-wide block tile
    return ((rho >> 5) << 6) | (b << 5) | (((rho >> 2) & 3) << 3) | (((rho >> 4) & 1) << 2) | (rho & 3);
}
__device__ __forceinline__ void v7_stage_b(const bf16_t* __restrict__ base, long long ld, int n0, int nrows, int k0, char* tile, int w, int lane, int b) {
    const int sub = lane >> 3, pch = lane & 7;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int rho = (w * 2 + i) * 8 + sub;
        int gr = n0 + v7_bcol(rho, b);
        if (gr > nrows - 1) gr = nrows - 1;                      // clamped columns are computed but never stored
        const bf16_t* src = base + (long long)gr * ld + k0 + ((pch ^ sub) << 3);
        const unsigned off = __builtin_amdgcn_readfirstlane((unsigned)((w * 2 + i) * 1024));
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(tile + off), 16, 0, 0);
    }
}

// s_waitcnt vmcnt(8 + S) for a run-time, wave-uniform S in 0..32 (rounded DOWN to an even number: a smaller count waits longer, never shorter).
// Loads, stores and LDS-DMA share ONE in-order counter: after a tile's epilogue the S youngest entries are its stores, behind them the 8
// LDS-DMA instructions of the next tile's K-tile 1, behind those the 8 of K-tile 0.  `vmcnt(8)` at the next tile's start therefore also waited
// for all but 8 of the STORES (6.7 us per tile boundary in the FFN-up product with two outputs, profiles/r04_v7_stamps.txt); with the store
// count in the immediate the stores drain under K-tiles 0 and 1 (K-tile 2's data is younger than they are: its wait retires them).
__device__ __forceinline__ void v7_wait_dma8_plus(int S) {
#define V7_WCASE(H) case H: asm volatile("s_waitcnt vmcnt(%0)" :: "n"(8 + 2 * (H)) : "memory"); break;
    switch (S >> 1) {
        V7_WCASE(16) V7_WCASE(15) V7_WCASE(14) V7_WCASE(13) V7_WCASE(12) V7_WCASE(11) V7_WCASE(10) V7_WCASE(9) V7_WCASE(8) V7_WCASE(7) V7_WCASE(6)
        V7_WCASE(5) V7_WCASE(4) V7_WCASE(3) V7_WCASE(2) V7_WCASE(1)
        default: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    }
#undef V7_WCASE
}

// one 8-column chunk of a lane: c0 / c1 = the accumulators of n-tiles j = 0 / 1 (columns gn .. gn + 3 / gn + 4 .. gn + 7 of row gm)
template <bool VO>
__device__ __forceinline__ void v7_chunk(const av_gemm_args& p, const FastFlags& fl, const f32x4& c0, const f32x4& c1, const float (&bv)[8],
                                         int gm, int gn, int m_end, long long cbase, const float* R) {
    if (gm >= m_end || gn >= p.N) return;
    float v[8];
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[e] = c0[e] * p.alpha + bv[e]; v[4 + e] = c1[e] * p.alpha + bv[4 + e]; }
    epilogue_store_t<VO>(p, fl, v, cbase + (long long)gm * p.ldc + gn, gm, gn, gn + 8 <= p.N, R);
}

// ACT / ODT >= 0: the activation / output type are compile-time constants of this instantiation (the epilogue body shrinks to what the class
// needs); -1: read from the arguments at run time.  VO: see epilogue_store_t.
// NM1: m-tiles per wavefront in the SECOND 128-row A half, a compile-time constant of the instantiation (4 = 256-row tiles; 3 = 224-row tiles:
// both wavefront groups own three of its six live 16-row tiles, so no run-time test sits inside a cluster - see V7_MMA).
template <int ACT, int ODT, bool VO, int NM1>
__global__ __launch_bounds__(V4_NT, 2) void gemm_nt_bf16_v7_kernel(const av_gemm_args pa, const int nbM, const int nbN, const FastFlags fl, const int nfull, const int bm_eff) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    static_assert(V4_LEAD == 8, "v7 assumes the 10-slot ring");
    av_gemm_args p = pa;
    if constexpr (ACT >= 0) p.act = ACT;
    if constexpr (ODT >= 0) p.out_dtype = ODT;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wr = __builtin_amdgcn_readfirstlane(w >> 2), wc = w & 3;
    const int r = lane & 15, g = lane >> 4;
    const int ntile = nbM * nbN;
    // my share of the full tiles: XCD label x owns ids [cs, cs + cn) (bijective split of [0, nfull)), its S workgroups take local ids slot, slot + S, ...
    const int G = (int)gridDim.x;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int S = (G - xcd + 7) >> 3;
    int cs, cn;
    { const int q = nfull >> 3, rem = nfull & 7; cn = q + (xcd < rem ? 1 : 0); cs = xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q; }
    auto coords = [&](int id, int& m0, int& n0, int& mb) {
        constexpr int GM = 4;
        const int per_group = GM * nbN;
        const int grp = id / per_group, in_grp = id - grp * per_group;
        const int first_m = grp * GM;
        const int gsz = nbM - first_m < GM ? nbM - first_m : GM;
        mb = first_m + in_grp % gsz;
        m0 = mb * bm_eff; n0 = (in_grp / gsz) * V4_BN;
    };
    const int z = blockIdx.z;
    const int zo = p.batch_inner > 0 ? z / p.batch_inner : 0;
    const int zi = p.batch_inner > 0 ? z % p.batch_inner : z;
    const bf16_t* A = (const bf16_t*)p.A + (long long)zo * p.oA + (long long)zi * p.sA;
    const bf16_t* B = (const bf16_t*)p.B + (long long)zo * p.oB + (long long)zi * p.sB;
    const long long cbase = (long long)zo * p.oC + (long long)zi * p.sC;
    const float* R = p.R ? p.R + (long long)zi * p.sR : nullptr;
    const float* bias = p.bias ? p.bias + (long long)zi * p.sBias : nullptr;
    const int nk = p.K / BK, nh = 4 * nk;

    // half-tile (t, j) of the tile at (tm0, tn0) into ring slot sidx; j: 0 = A0, 1 = B1, 2 = A1, 3 = B0 (compile-time constant at every call)
    auto issue = [&](int tm0, int tn0, int t, int j, int sidx) {
        char* sl_ = smem + sidx * V4_HALF;
        const int k0 = t * BK;
        // rows of the 256-row window beyond this tile's bm_eff rows belong to the next row tile and get no MFMAs: their lanes re-read the
        // tile's last row (one L1-resident line per instruction) instead of streaming 128 B each from L2
        const int a_end = p.M - tm0 < bm_eff ? p.M : tm0 + bm_eff;
        if (j == 0) stage_rows<2>(A, p.lda, tm0, a_end, k0, sl_, w, lane);
        else if (j == 2) stage_rows<2>(A, p.lda, tm0 + 128, a_end, k0, sl_, w, lane);
        else if (j == 3) v7_stage_b(B, p.ldb, tn0, p.N, k0, sl_, w, lane, 0);
        else v7_stage_b(B, p.ldb, tn0, p.N, k0, sl_, w, lane, 1);
    };
    auto prefetch = [&](int tm0, int tn0) {                  // ring slots 0 .. 7 <- K-tile 0 and (if any) K-tile 1
        issue(tm0, tn0, 0, 0, 0); issue(tm0, tn0, 0, 1, 1); issue(tm0, tn0, 0, 2, 2); issue(tm0, tn0, 0, 3, 3);
        if (nk > 1) { issue(tm0, tn0, 1, 0, 4); issue(tm0, tn0, 1, 1, 5); issue(tm0, tn0, 1, 2, 6); issue(tm0, tn0, 1, 3, 7); }
    };
    // bias of the 64 columns of my wavefront: lane l holds column wc 64 + l (ONE register across the main loop; the epilogue fetches the 16
    // values a lane needs - columns b 32 + g 8 + e - from their lanes by ds_bpermute)
    auto load_bias = [&](int tn0) -> float {
        const int gn = tn0 + wc * 64 + lane;
        return (bias && gn < p.N) ? bias[gn] : 0.f;
    };

    const int sw = r & 7;
    const int ch0 = (g ^ sw) << 4, ch1 = ((4 + g) ^ sw) << 4;
    // m-tiles of a 128-row A half alternate between the two wavefront groups (group wr owns 16-row tiles wr, 2 + wr, 4 + wr, 6 + wr): a tile of
    // bm_eff < 256 rows then takes its missing m-tiles evenly from both groups (the groups alternate on a SIMD and meet at every barrier, so
    // the K-tile lasts as long as the group with MORE tiles: 208 rows = 7 + 6 tiles instead of 8 + 5)
    const int a_row = (wr * 16 + r) * 128, b_row = (wc * 32 + r) * 128;

    int li = slot;
    int m0 = 0, n0 = 0, mb = 0;
    float bnext = 0.f;                                       // bias (my lane's column) of the tile about to start (loop-carried)
    if (li < cn) {
        coords(cs + li, m0, n0, mb);
        bnext = load_bias(n0);                               // BEFORE the prefetch: the consume below then needs the load only (the 16 LDS-DMA are younger)
        prefetch(m0, n0);
        asm volatile("" : "+v"(bnext));                      // no load is pending in a loop-carried register (the compiler would wait vmcnt(0) at its every use)
    }
    int pend = 0;                                            // stores of the previous tile's epilogue that may still be in flight (a lower bound of their count)
    static const bool v7_decouple = true;
    int seq = 0;                                             // (diagnostic stamps only)
    while (li < cn) {
        AV_STAMP7(seq, 0);
        const int nli = li + S;
        const bool has_next = nli < cn;
        int nm0 = 0, nn0 = 0, nmb = 0;
        if (has_next) coords(cs + nli, nm0, nn0, nmb);
        const int rows_here = p.M - m0 < bm_eff ? p.M - m0 : bm_eff;

        f32x4 acc[2][2][4][2];                               // [A half][B half][m-tile 2 i + wr][n-tile]; lane (g, r): row r of the m-tile, n-tile columns 4 g + e
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[a][b][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

        // K-tile 0: the 8 oldest LDS-DMA instructions of this tile; K-tile 1's eight and the previous tile's stores may stay in flight
        if (nk > 1) v7_wait_dma8_plus(pend); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // this tile's bias was loaded at the START of the previous tile (first tile: in the prologue) and consumed at the end of its main loop:
        // a consume HERE would make the compiler wait for everything in flight, i.e. for the previous tile's stores
        const float bcur = bnext;
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (wr == 1) __builtin_amdgcn_s_barrier();           // wavefronts 4-7 run one barrier behind
        AV_STAMP7(seq, 1);
        float bload = 0.f;
        if (has_next) bload = load_bias(nn0);                // next tile's bias: in flight under this tile's main loop

        bf16x8 fa[4][2], fb[2][2];
        // Three-m-tile instantiation (222 registers): the B0 fragments keep their own 16 registers from phase 0 to phase 3 instead of being
        // re-read (4 of the 27 ds_read_b128 of a K-tile; the four-m-tile form has no registers to spare).  -DAV_V7_KEEPB0=0: A/B builds.
#ifndef AV_V7_KEEPB0
#define AV_V7_KEEPB0 1
#endif
        constexpr bool KEEPB0 = NM1 == 3 && AV_V7_KEEPB0;
        bf16x8 fb0[2][2];
// Straight-line clusters: the count of m-tiles is a compile-time constant per A half (4 in half 0, NM1 in half 1).  A per-tile run-time test
// (`if (i < nmt)`, the first form of the short-tile feature) compiled to a scalar branch around EVERY pair of MFMAs and fragment reads - a dozen
// taken branches inside a 16-MFMA cluster that is meant to issue back to back (profiles/r04_branch_free_clusters_ab.txt: 15-20 % of the loop).
#define V7_READ_A(SLOT, N)                                                                                         \
    _Pragma("unroll") for (int i = 0; i < (N); ++i) {                                                              \
        fa[i][0] = *(const bf16x8*)(smem + sl[SLOT] * V4_HALF + a_row + i * 4096 + ch0);                           \
        fa[i][1] = *(const bf16x8*)(smem + sl[SLOT] * V4_HALF + a_row + i * 4096 + ch1); }
#define V7_READ_B(SLOT, FB)                                                                                         \
    _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                                \
        FB[j][0] = *(const bf16x8*)(smem + sl[SLOT] * V4_HALF + b_row + j * 2048 + ch0);                           \
        FB[j][1] = *(const bf16x8*)(smem + sl[SLOT] * V4_HALF + b_row + j * 2048 + ch1); }
#define V7_MMA(QA, QB, N, FB)                                                                                       \
    __builtin_amdgcn_s_barrier();                                                                                  \
    __builtin_amdgcn_s_waitcnt(0xC07F);                                                                            \
    __builtin_amdgcn_sched_barrier(0);                                                                             \
    __builtin_amdgcn_s_setprio(1);                                                                                 \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                               \
        _Pragma("unroll") for (int i = 0; i < (N); ++i)                                                            \
            _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                          \
                acc[QA][QB][i][j] = AV_MFMA_F32_16X16X32_LP(FB[j][ks], fa[i][ks], acc[QA][QB][i][j], 0, 0, 0);     \
    __builtin_amdgcn_s_setprio(0);                                                                                 \
    __builtin_amdgcn_sched_barrier(0);                                                                             \
    __builtin_amdgcn_s_barrier();                                                                                  \
    asm volatile("" ::: "memory");

        int b4 = 0;                                          // (4 t) % V4_NS: ring slot of half-tile (t, 0)
        for (int t = 0; t < nk; ++t) {
            int sl[4], si[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                int x = b4 + j; x = x >= V4_NS ? x - V4_NS : x; sl[j] = x;
                int y = b4 + j - 2; y = y < 0 ? y + V4_NS : y; si[j] = y;
            }
            b4 += 4; b4 = b4 >= V4_NS ? b4 - V4_NS : b4;
            if constexpr (KEEPB0) { V7_READ_B(3, fb0) } else { V7_READ_B(3, fb) }
            __builtin_amdgcn_sched_barrier(0);
            V7_READ_A(0, 4)
            if (4 * t + V4_LEAD + 0 < nh) issue(m0, n0, t + 2, 0, si[0]);
            if constexpr (KEEPB0) { V7_MMA(0, 0, 4, fb0) } else { V7_MMA(0, 0, 4, fb) }
            V7_READ_B(1, fb)
            if (4 * t + V4_LEAD + 1 < nh) issue(m0, n0, t + 2, 1, si[1]);
            V7_MMA(0, 1, 4, fb)
            V7_READ_A(2, NM1)
            if (4 * t + V4_LEAD + 2 < nh) issue(m0, n0, t + 2, 2, si[2]);
            V7_MMA(1, 1, NM1, fb)
            if constexpr (!KEEPB0) { V7_READ_B(3, fb) }
            if (4 * t + V4_LEAD + 3 < nh) {
                issue(m0, n0, t + 2, 3, si[3]);
                // all of K-tile t + 1 has landed.  In K-tile 0 that data is OLDER than the previous tile's stores (it was requested before
                // them): they may stay in flight; from K-tile 1 on the awaited data is younger than the stores and the wait retires them
                if (t == 0 && pend > 0) v7_wait_dma8_plus(pend); else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if constexpr (KEEPB0) { V7_MMA(1, 0, NM1, fb0) } else { V7_MMA(1, 0, NM1, fb) }
        }
#undef V7_READ_A
#undef V7_READ_B
#undef V7_MMA
        if (wr == 0) __builtin_amdgcn_s_barrier();           // balance the entry barrier of wavefronts 4-7: every fragment read of this tile has completed
        asm volatile("" ::: "memory");
        AV_STAMP7(seq, 2);

        // the ring is free: request the next tile's first two K-tiles, then store this tile from the registers
        asm volatile("" : "+v"(bload));                      // next tile's bias (requested a main loop ago): nothing else of mine is in flight here
        bnext = bload;
        if (has_next) prefetch(nm0, nn0);
        const int m_end = m0 + rows_here;
        {
            // stores this wavefront is about to issue, counted conservatively: a chunk (16 rows x 64 columns of the wavefront) stores iff its first
            // row and first column are inside the tile; per chunk one 16-B store per bf16 output, two per fp32 output
            int na = 0;
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int i = 0; i < 4; ++i) na += (m0 + a * 128 + (2 * i + wr) * 16 < m_end) ? 1 : 0;
            const int nb = (n0 + wc * 64 < p.N ? 1 : 0) + (n0 + wc * 64 + 32 < p.N ? 1 : 0);
            const int spc = (p.out_dtype == AV_F32 ? 2 : 1) * (p.C2 ? 2 : 1);
            pend = v7_decouple ? __builtin_amdgcn_readfirstlane(na * nb * spc) : 0;
            if (pend > 32) pend = 32;
        }
        float bv[8], bo[8];                                  // bias of my columns of B half 0 / 1 (rolled form: the two windows swap)
#pragma unroll
        for (int e = 0; e < 8; ++e) { bv[e] = __shfl(bcur, g * 8 + e, 64); bo[e] = __shfl(bcur, 32 + g * 8 + e, 64); }
        constexpr bool ROLL = ACT != AV_ACT_MUL_AUX;
        if constexpr (!ROLL) {
            // saved-factor multiply (the dX product above an FFN activation): all 16 chunks unrolled, so that the factor loads of later chunks are
            // in flight while earlier chunks are stored - the rolled form below waits for a load round trip per quadrant (150 -> 136 us on
            // 12736 x 4096 x 1024, = the plain product).  The fp32-residual classes did not gain from it (their epilogue moves 2 x 52 MB per
            // round at the memory rate whatever the order) and stay rolled with the rest
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int b = 0; b < 2; ++b)
                        v7_chunk<VO>(p, fl, acc[a][b][i][0], acc[a][b][i][1], b ? bo : bv, m0 + a * 128 + (2 * i + wr) * 16 + r, n0 + wc * 64 + b * 32 + g * 8, m_end, cbase, R);
        } else
        {
            // ONE copy of the epilogue body for four m-tiles (code size: it is inlined with every activation / dropout / residual branch).  The
            // quadrants (a, b) = (0,0) (0,1) (1,0) (1,1) pass through the registers of quadrant (0,0) - dead once stored - and the two bias
            // windows swap, so the rolled loop costs no registers
#pragma unroll 1
            for (int qd = 0; qd < 4; ++qd) {
                const int a = qd >> 1, b = qd & 1;
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    v7_chunk<VO>(p, fl, acc[0][0][i][0], acc[0][0][i][1], bv, m0 + a * 128 + (2 * i + wr) * 16 + r, n0 + wc * 64 + b * 32 + g * 8, m_end, cbase, R);
                if (qd == 0) {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j) acc[0][0][i][j] = acc[0][1][i][j];
                } else if (qd == 1) {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j) acc[0][0][i][j] = acc[1][0][i][j];
                } else if (qd == 2) {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j) acc[0][0][i][j] = acc[1][1][i][j];
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float t = bv[e]; bv[e] = bo[e]; bo[e] = t; }
            }
        }
        AV_STAMP7(seq, 3);
        ++seq;
        li = nli; m0 = nm0; n0 = nn0; mb = nmb;
    }

    // the short last round: quadrant jobs (v4's scheme, image epilogue), spread over the same persistent workgroups
    const int nq = 4 * (ntile - nfull);
    if (nq > 0) {
        int qs, qn_;
        { const int q = nq >> 3, rem = nq & 7; qn_ = q + (xcd < rem ? 1 : 0); qs = xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q; }
        for (int lj = slot; lj < qn_; lj += S) {
            const int j = qs + lj;
            int tm0, tn0, tmb;
            coords(nfull + (j >> 2), tm0, tn0, tmb);
            const int quad = j & 3;
            const int rows_here = p.M - tm0 < bm_eff ? p.M - tm0 : bm_eff;
            const int qm = tm0 + (quad >> 1) * 128, qn = tn0 + (quad & 1) * 128;
            __syncthreads();                                 // (also retires this workgroup's outstanding stores: __syncthreads waits vmcnt(0))
            if (qm < tm0 + rows_here && qn < p.N)            // block-uniform
                v4_quadrant_job(p, fl, smem, A, B, qm, qn, tm0 + rows_here, cbase, R, bias);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// v6 (EXPERIMENTAL, off by default: AVAMD_GEMM_V6=1; measured in DESIGN.md section 7): 256 x 256 x 64 tile on FOUR wavefronts (one per SIMD,
// 512-register budget), each owning 128 x 128 of the tile (8 x 8 MFMA tiles,
// 256 accumulator registers).  Against the 8-phase kernel: LDS fragment traffic per K-tile 128 KB instead of 196 KB, ONE workgroup
// barrier per K-tile instead of eight, and every fragment / DMA instruction has a fixed slot in a stream of 16 groups of 8 MFMAs.
// Two 64 KB stages (A 256 x 64 | B 256 x 64, 128-B rows, 16-B chunks XOR-swizzled with row & 7 on the source side).  Per K-tile t:
//   groups 0-7   (k-step 0): MFMAs with B set `fb`; each group reads the next A fragment and one B fragment of k-step 1 (`fbn`), and issues
//                one LDS-DMA instruction of tile t+1's second half;
//   groups 8-13  (k-step 1): MFMAs with `fbn`; reads of the next A fragment;
//   barrier:     every wavefront has consumed stage t & 1 (its last two A fragments are in registers) and tile t+1 has landed (vmcnt(0));
//   groups 14-15: MFMAs of tile t beside the first reads of tile t+1 (B set of k-step 0, first A fragment) and the first half of tile
//                t+2's LDS-DMA into the stage just freed.
constexpr int V6_NT = 256, V6_STAGE = 65536, V6_CLD = 260, V6_EPI = 128 * V6_CLD * 4, V6_LDS = V6_EPI;

__global__ __launch_bounds__(V6_NT, 1) void gemm_nt_bf16_v6_kernel(const av_gemm_args p, const int nbM, const int nbN, const FastFlags fl) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    AV_STAMP(0);
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = w >> 1, wc = w & 1, r = lane & 15, g = lane >> 4;
    const int ntile = nbM * nbN;
    int bid = blockIdx.x;
    {
        const int q = ntile >> 3, rem = ntile & 7, xcd = bid & 7, slot = bid >> 3;
        bid = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + slot;
    }
    constexpr int GM = 4;
    const int per_group = GM * nbN;
    const int grp = bid / per_group, in_grp = bid - grp * per_group;
    const int first_m = grp * GM;
    const int gsz = nbM - first_m < GM ? nbM - first_m : GM;
    const int mb = first_m + in_grp % gsz, nb = in_grp / gsz;
    const int m0 = mb * 256, n0 = nb * 256;
    const int z = blockIdx.z;
    const int zo = p.batch_inner > 0 ? z / p.batch_inner : 0;
    const int zi = p.batch_inner > 0 ? z % p.batch_inner : z;
    const bf16_t* A = (const bf16_t*)p.A + (long long)zo * p.oA + (long long)zi * p.sA;
    const bf16_t* B = (const bf16_t*)p.B + (long long)zo * p.oB + (long long)zi * p.sB;

    // LDS-DMA sources of my 8 + 8 wave-instructions (8 rows x 128 B each): rows beyond the matrix are clamped (computed, never stored)
    const bf16_t* pa[8];
    const bf16_t* pb[8];
    {
        const int sub = lane >> 3, choff = ((lane & 7) ^ sub) << 3;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = (w * 8 + i) * 8 + sub;
            int ga = m0 + row, gb = n0 + row;
            ga = ga < p.M ? ga : p.M - 1;
            gb = gb < p.N ? gb : p.N - 1;
            pa[i] = A + (long long)ga * p.lda + choff;
            pb[i] = B + (long long)gb * p.ldb + choff;
        }
    }
    const unsigned dst_a = __builtin_amdgcn_readfirstlane((unsigned)(w * 8192)), dst_b = dst_a + 32768u;
#define V6_DMA(T, Q)                                                                                               \
    do {                                                                                                           \
        char* st_ = smem + ((T) & 1) * V6_STAGE;                                                                   \
        if ((Q) < 8) __builtin_amdgcn_global_load_lds((gptr_t)(pa[(Q) & 7] + (T) * BK), (lptr_t)(st_ + dst_a + ((Q) & 7) * 1024), 16, 0, 0); \
        else __builtin_amdgcn_global_load_lds((gptr_t)(pb[(Q) & 7] + (T) * BK), (lptr_t)(st_ + dst_b + ((Q) & 7) * 1024), 16, 0, 0);       \
    } while (0)

    f32x4 acc[8][8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = p.K / BK;
    // prologue: tile 0 and the first half (A rows) of tile 1
#pragma unroll
    for (int q = 0; q < 16; ++q) V6_DMA(0, q);
    if (nk > 1) {
#pragma unroll
        for (int q = 0; q < 8; ++q) V6_DMA(1, q);
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    AV_STAMP(1);

    const int sw = r & 7;
    const int ch0 = (g ^ sw) << 4, ch1 = ((4 + g) ^ sw) << 4;
    const int a_off = (wr * 128 + r) * 128, b_off = 32768 + (wc * 128 + r) * 128;
#define V6_A(ST, I, CH) (*(const bf16x8*)((ST) + a_off + (I) * 2048 + (CH)))
#define V6_B(ST, J, CH) (*(const bf16x8*)((ST) + b_off + (J) * 2048 + (CH)))
#define V6_MMA8(I, FA, FB)                                                                                         \
    _Pragma("unroll") for (int j = 0; j < 8; ++j) acc[I][j] = AV_MFMA_F32_16X16X32_LP(FA, FB[j], acc[I][j], 0, 0, 0);

    bf16x8 fb[8], fbn[8], fa0, fa1, fa2;
    {
        const char* st = smem;
#pragma unroll
        for (int j = 0; j < 8; ++j) fb[j] = V6_B(st, j, ch0);
        fa0 = V6_A(st, 0, ch0);
    }
    for (int t = 0; t < nk; ++t) {
        const char* st = smem + (t & 1) * V6_STAGE;
        const char* stn = smem + ((t + 1) & 1) * V6_STAGE;
        const bool more1 = t + 1 < nk, more2 = t + 2 < nk;
        // ---- k-step 0
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            __builtin_amdgcn_sched_barrier(0);
            fa1 = i < 7 ? V6_A(st, i + 1, ch0) : V6_A(st, 0, ch1);
            fbn[i] = V6_B(st, i, ch1);
            if (more1) V6_DMA(t + 1, 8 + i);
            __builtin_amdgcn_sched_barrier(0);
            V6_MMA8(i, fa0, fb)
            fa0 = fa1;
        }
        // ---- k-step 1, groups 0-5
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            __builtin_amdgcn_sched_barrier(0);
            fa1 = V6_A(st, i + 1, ch1);
            if (i == 5) fa2 = V6_A(st, 7, ch1);
            __builtin_amdgcn_sched_barrier(0);
            V6_MMA8(i, fa0, fbn)
            fa0 = fa1;
        }
        // ---- hand-over: stage t & 1 is consumed (fa0 = A(6, 1), fa2 = A(7, 1) are in registers), tile t+1 has landed
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        if (more1) {
#pragma unroll
            for (int j = 0; j < 4; ++j) fb[j] = V6_B(stn, j, ch0);
        }
        if (more2) {
#pragma unroll
            for (int q = 0; q < 4; ++q) V6_DMA(t + 2, q);
        }
        __builtin_amdgcn_sched_barrier(0);
        V6_MMA8(6, fa0, fbn)
        __builtin_amdgcn_sched_barrier(0);
        if (more1) {
#pragma unroll
            for (int j = 4; j < 8; ++j) fb[j] = V6_B(stn, j, ch0);
            fa0 = V6_A(stn, 0, ch0);
        }
        if (more2) {
#pragma unroll
            for (int q = 4; q < 8; ++q) V6_DMA(t + 2, q);
        }
        __builtin_amdgcn_sched_barrier(0);
        V6_MMA8(7, fa2, fbn)
    }
#undef V6_A
#undef V6_B
#undef V6_MMA8
#undef V6_DMA
    AV_STAMP(2);

    // ---- epilogue: two images of 128 rows x 256 columns (m-tiles 4 h .. 4 h + 3 of every wavefront) through LDS, then 16-B rows
    float* cs = (float*)smem;
    const long long cbase = (long long)zo * p.oC + (long long)zi * p.sC;
    const float* R = p.R ? p.R + (long long)zi * p.sR : nullptr;
    const float* bias = p.bias ? p.bias + (long long)zi * p.sBias : nullptr;
    constexpr int CPR = 32;
    float bv[8];
    {
        const int gnt = n0 + (tid % CPR) * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) bv[e] = (bias && gnt + e < p.N) ? bias[gnt + e] : 0.f;
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    cs[(wr * 64 + i * 16 + 4 * g + e) * V6_CLD + wc * 128 + j * 16 + r] = acc[4 * h + i][j][e] * p.alpha;
        __syncthreads();
        // one wavefront per SIMD: nothing else hides the LDS round trip of a row chunk, so the next chunk's reads are issued before this one
        // is processed (rolled loop: the shared store tail is instantiated once)
        const int cc = (tid % CPR) * 8, gn = n0 + cc, row0 = tid / CPR;            // rows row0 + 8 it
        const bool full = gn + 8 <= p.N;
        f32x4 n0v = *(const f32x4*)(cs + row0 * V6_CLD + cc), n1v = *(const f32x4*)(cs + row0 * V6_CLD + cc + 4);
        for (int it = 0; it < 128 * CPR / V6_NT; ++it) {
            const int row = row0 + it * (V6_NT / CPR);
            const f32x4 v0 = n0v, v1 = n1v;
            if (it + 1 < 128 * CPR / V6_NT) {
                n0v = *(const f32x4*)(cs + (row + V6_NT / CPR) * V6_CLD + cc);
                n1v = *(const f32x4*)(cs + (row + V6_NT / CPR) * V6_CLD + cc + 4);
            }
            const int gm = m0 + (row >> 6) * 128 + h * 64 + (row & 63);
            if (gm >= p.M || gn >= p.N) continue;
            float v[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] = v0[e] + bv[e]; v[4 + e] = v1[e] + bv[4 + e]; }
            const long long off = cbase + (long long)gm * p.ldc + gn;
            epilogue_store(p, fl, v, off, gm, gn, full, R);
        }
    }
#ifdef AV_GEMM_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    AV_STAMP(3);
}

template <int BNT, bool CONV, bool AKM = false, bool BKM = false>
int launch_fast(const av_gemm_args& p, hipStream_t st, const FastFlags& fl) {
    constexpr int STAGE = TILE_A + BNT * BK * 2;
    constexpr int EPI = BM * (BNT + 4) * 4 + 2048;           // image + BatchNorm partial scratch
    constexpr int LDS = 2 * STAGE > EPI ? 2 * STAGE : EPI;
    auto kern = gemm_nt_bf16_kernel<BNT, CONV, AKM, BKM>;
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess) {
            av_set_error("av_gemm(fast): cannot raise dynamic LDS to %d", LDS);
            return AV_ERR_LAUNCH;
        }
        attr_done = true;
    }
    const int nbM = av_cdiv(p.M, BM), nbN = av_cdiv(p.N, BNT);
    dim3 grid((unsigned)(nbM * (long long)nbN), 1, (unsigned)p.batch);
    hipLaunchKernelGGL(kern, grid, dim3(NT), LDS, st, p, nbM, nbN, fl);
    AV_LAUNCH_CHECK();
    return AV_OK;
}

// returns -1 when the arguments do not qualify for the fast path
int av_gemm_fast_try(const av_gemm_args& p, hipStream_t st) {
    if (p.in_dtype != AV_BF16) return -1;
    if (p.k_total > 0 && !(p.a_mode == AV_A_TRANS && p.b_mode == AV_B_KN && p.batch_inner == 0 && (long long)(p.batch - 1) * p.K < p.k_total)) return -1;
    const bool conv = p.a_mode == AV_A_CONV2D;
    const bool akm = p.a_mode == AV_A_TRANS, bkm = p.b_mode == AV_B_KN;        // k-major operands: A [K][M], B [K][N]
    if (!conv && !akm && p.a_mode != AV_A_ROWMAJOR) return -1;
    if (!bkm && p.b_mode != AV_B_NK) return -1;
    if (p.K < BK || p.ldb % 8 || p.sA % 8 || p.sB % 8 || p.oA % 8 || p.oB % 8) return -1;
    if (p.K % BK && !(akm && bkm)) return -1;                                   // a row-major operand needs whole 64-wide K-steps
    if (conv && bkm) return -1;
    if ((akm && (p.M % 8 || p.M <= 64)) || (bkm && (p.N % 8 || p.N <= 64))) return -1;
    if (!al16(p.A) || !al16(p.B) || p.M < 1 || p.N < 1) return -1;
    if (conv) {
        if (p.cCin % 64 || p.cCtot % 8 || p.cCoff % 8 || (p.cKh * p.cKw > 32 && (p.cKw != 1 || p.cKh > 65535))) return -1;
        if (p.cPM && (p.cNF < 1 || p.cT != 1 || p.cKt != 1 || p.batch != 1 || p.cKh * p.cKw > 32 || p.M % (p.cOh * p.cOw * p.cNF))) return -1;     // position-major: plain 2-D convolutions, whole image blocks
    } else {
        if (p.lda % 8 || p.stats) return -1;
    }
    const long long oes = p.out_dtype == AV_F32 ? 4 : 2;
    FastFlags fl;
    fl.c_vec = al16(p.C) && (!p.C2 || al16(p.C2)) && (p.ldc * oes) % 16 == 0 && (p.sC * oes) % 16 == 0 && (p.oC * oes) % 16 == 0;
    fl.r_vec = p.R && al16(p.R) && (p.ldr % 4 == 0) && (p.sR % 4 == 0);
    const long long aes = p.aux_dtype == AV_F32 ? 4 : 2;
    fl.aux_vec = p.aux && al16(p.aux) && (p.ldc * aes) % 16 == 0 && (p.sC * aes) % 16 == 0 && (p.oC * aes) % 16 == 0;
    const bool narrow = p.N <= 64;
    if (akm || bkm) {
        if (p.stats || p.lda % 8) return -1;
        // dW-shaped products (both operands k-major) of at least 32 tiles of 256 x 256 (QKV, FFN; the 1024 x 1024 out-projection gradient - 16 tiles x
        // 10+ K slices of fp32 partials - stays faster on the 128 x 128 kernel): the 8-phase kernel's k-major form, one workgroup per (tile, K slice).
        // AVAMD_GEMM_KM8=0 keeps them all on the 128 x 128 kernel, 2 forces the 8-phase form (tests; ops.matmul_tn picks its slice count by the same rule).
        static const int km8 = [] { const char* e = getenv("AVAMD_GEMM_KM8"); return e ? atoi(e) : 1; }();
        if (akm && bkm && km8 && p.M >= 256 && p.N >= 256 && (km8 >= 2 || av_cdiv(p.M, V4_BM) * av_cdiv(p.N, V4_BN) >= 32) && p.batch_inner == 0 && p.lda < (1 << 24) && p.ldb < (1 << 24)) {
            static bool km_attr = false;
            if (!km_attr) {
                if (hipFuncSetAttribute((const void*)gemm_nt_bf16_v4_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, V4_LDS) != hipSuccess) {
                    av_set_error("av_gemm(fast v4 k-major): cannot raise dynamic LDS to %d", V4_LDS);
                    return AV_ERR_LAUNCH;
                }
                km_attr = true;
            }
            const int nbM = av_cdiv(p.M, V4_BM), nbN = av_cdiv(p.N, V4_BN);
            hipLaunchKernelGGL((gemm_nt_bf16_v4_kernel<false, true>), dim3((unsigned)(nbM * nbN * p.batch), 1, 1), dim3(V4_NT), V4_LDS, st, p, nbM, nbN, fl,
                               nbM * nbN * p.batch, V4_BM);
            AV_LAUNCH_CHECK();
            return AV_OK;
        }
        if (akm && bkm) return launch_fast<128, false, true, true>(p, st, fl);
        if (akm) return launch_fast<128, false, true, false>(p, st, fl);
        return launch_fast<128, false, false, true>(p, st, fl);
    }
    // Tiling choice for the plain NT products: estimated time = rounds x (K-tiles x time per K-tile + per-tile prologue / epilogue), from
    // measurements on the step's shapes (tools/gemm_variants.py): 256 x 256 8-phase kernel 1.59 us per K-tile + 5 us on 256 slots,
    // 256 x 128 three-stage kernel 0.85 + 2 on 256 slots, 128 x 128 kernel 1.0 + 2.5 on 512 slots (two workgroups per CU).
    // AVAMD_GEMM_V4: 0 never, 1 (default) by that estimate, 2 whenever it applies (tests / diagnostics).
    static const int v4_mode = [] { const char* e = getenv("AVAMD_GEMM_V4"); return e ? atoi(e) : 1; }();
    static const int v2_mode = [] { const char* e = getenv("AVAMD_GEMM_V2"); return e ? atoi(e) : 2; }();   // 0 never, 1 always, 2 (default) when K >= 2048
    const bool v2_ok = !conv && !narrow && p.M >= 512 && (v2_mode == 1 || (v2_mode == 2 && p.K >= 2048));
    static const int v4_conv = [] { const char* e = getenv("AVAMD_GEMM_V4_CONV"); return e ? atoi(e) : 1; }();
    if (conv && v4_conv && v4_mode > 0 && p.N >= 256 && p.M >= 4096 && p.K >= 512 && p.batch == 1 && p.cKh * p.cKw <= 32 && p.alpha == 1.0f) {
        // ResNet layer3 / layer4 convolutions (N = 256 / 512): implicit im2col on the 8-phase kernel
        static bool v4c_attr = false;
        if (!v4c_attr) {
            if (hipFuncSetAttribute((const void*)gemm_nt_bf16_v4_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, V4_LDS) != hipSuccess) {
                av_set_error("av_gemm(fast v4 conv): cannot raise dynamic LDS to %d", V4_LDS);
                return AV_ERR_LAUNCH;
            }
            v4c_attr = true;
        }
        const int nbM = av_cdiv(p.M, V4_BM), nbN = av_cdiv(p.N, V4_BN);
        hipLaunchKernelGGL(gemm_nt_bf16_v4_kernel<true>, dim3((unsigned)(nbM * nbN), 1, 1), dim3(V4_NT), V4_LDS, st, p, nbM, nbN, fl, nbM * nbN, V4_BM);
        AV_LAUNCH_CHECK();
        return AV_OK;
    }
    static const int v6_mode = [] { const char* e = getenv("AVAMD_GEMM_V6"); return e ? atoi(e) : 0; }();
    if (v6_mode > 0 && !conv && !narrow && p.M >= 256 && p.N >= 256 && !p.stats) {
        static bool v6_attr = false;
        if (!v6_attr) {
            if (hipFuncSetAttribute((const void*)gemm_nt_bf16_v6_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, V6_LDS) != hipSuccess) {
                av_set_error("av_gemm(fast v6): cannot raise dynamic LDS to %d", V6_LDS);
                return AV_ERR_LAUNCH;
            }
            v6_attr = true;
        }
        const int nbM = av_cdiv(p.M, 256), nbN = av_cdiv(p.N, 256);
        hipLaunchKernelGGL(gemm_nt_bf16_v6_kernel, dim3((unsigned)(nbM * nbN), 1, (unsigned)p.batch), dim3(V6_NT), V6_LDS, st, p, nbM, nbN, fl);
        AV_LAUNCH_CHECK();
        return AV_OK;
    }
    if (!conv && !narrow && v4_mode > 0 && p.M >= 256 && p.N >= 256) {
        const double nk = p.K / 64.0;
        const int nbN4 = av_cdiv(p.N, V4_BN);
        static const int v4_tail = [] { const char* e = getenv("AVAMD_GEMM_V4_TAIL"); return e ? atoi(e) : 1; }();
        static const int v4_bm = [] { const char* e = getenv("AVAMD_GEMM_V4_BM"); return e ? atoi(e) : 0; }();     // 0: choose; else force (multiple of 16)
        static const int v7_mode = [] { const char* e = getenv("AVAMD_GEMM_V7"); return e ? atoi(e) : 1; }();
        // persistent form (see the kernel's notes): specialised per epilogue class; the classes need whole 16-B chunks everywhere
        // (epilogue_store_t<true>) - anything else stays on the v4 kernel
        const bool v7_vo = p.N % 8 == 0 && fl.c_vec && (!p.R || fl.r_vec) && (!p.aux || fl.aux_vec);
        const bool v7_cls = (p.out_dtype == AV_BF16 && (p.act == AV_ACT_NONE || p.act == AV_ACT_GELU || p.act == AV_ACT_GELU_GF || p.act == AV_ACT_MUL_AUX)) ||
                            (p.out_dtype == AV_F32 && p.act == AV_ACT_NONE);
        const bool v7_ok = v7_mode > 0 && v7_vo && v7_cls;
        // row-tile height bm (a multiple of 16, rows of the 256-row window a tile owns): it sets the tile COUNT (248 tiles of 208 rows fill 256 CUs
        // where 200 of 256 rows leave 56 idle), and - v7 only - tiles of <= 224 rows run the instantiation whose second A half multiplies three
        // m-tiles per wavefront instead of four (NM1 = 3: 7/8 of the MFMAs; measured ~0.9 of the tile time)
        int best_bm = V4_BM; double e4 = 1e30; int best_full = 0;
        for (int bm = V4_BM; bm >= 160; bm -= 16) {
            if (v4_bm && bm != v4_bm) continue;
            const long long t4 = (long long)av_cdiv(p.M, bm) * nbN4 * p.batch;
            const double per_tile = nk * 1.55 * ((v7_ok && bm <= 224) ? 0.9 : 1.0) + 4.0;      // us
            const long long r4 = t4 % 256;
            const bool tail = p.batch == 1 && v4_tail && t4 > 256 && r4 > 0 && r4 <= 128;
            const double e = tail ? (double)(t4 / 256) * per_tile + (double)((4 * r4 + 255) / 256) * (nk * 0.8 + 4.0) : (double)((t4 + 255) / 256) * per_tile;   // a quadrant job: 16-18 us at K = 1024
            if (e < e4 * 0.98) { e4 = e; best_bm = bm; best_full = tail ? (int)(t4 - r4) : (int)t4; }
        }
        const long long t2 = (long long)av_cdiv(p.M, V2_BM) * av_cdiv(p.N, V2_BN) * p.batch;
        const long long t1 = (long long)av_cdiv(p.M, BM) * av_cdiv(p.N, 128) * p.batch;
        const double e2 = (double)((t2 + 255) / 256) * (nk * 0.85 + 2.0);
        const double e1 = (double)((t1 + 511) / 512) * (nk * 1.0 + 2.5);
        if (v7_ok && (v7_mode >= 2 || v4_mode >= 2 || e4 < (v2_ok ? (e2 < e1 ? e2 : e1) : e1))) {
            static int ncu = 0;
            if (!ncu) {
                int dev = 0;
                if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu <= 0) ncu = 256;
            }
            static const int v7_g = [] { const char* e = getenv("AVAMD_GEMM_V7_G"); return e ? atoi(e) : 0; }();     // diagnostic: persistent workgroups per launch (0 = one per CU)
            const int ng = v7_g > 0 && v7_g < ncu ? v7_g : ncu;
            const int nbM = av_cdiv(p.M, best_bm);
            const int ntile = nbM * nbN4;
            int nfull = ntile;
            if (p.batch == 1 && v4_tail && ntile > ng) { const int r = ntile % ng; if (r > 0 && r <= ng / 2) nfull = ntile - r; }
            const int G = ntile < ng ? ntile : ng;
            int rc = AV_OK;
            auto go = [&](auto kern) {
                static bool attr = false;                    // one flag per instantiation (the lambda's call operator is a template)
                if (!attr) {
                    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, V4_LDS) != hipSuccess) {
                        av_set_error("av_gemm(fast v7): cannot raise dynamic LDS to %d", V4_LDS);
                        rc = AV_ERR_LAUNCH;
                        return;
                    }
                    attr = true;
                }
                hipLaunchKernelGGL(kern, dim3((unsigned)G, 1, (unsigned)p.batch), dim3(V4_NT), V4_LDS, st, p, nbM, nbN4, fl, nfull, best_bm);
            };
            const bool nm3 = best_bm <= 224;                 // tiles of <= 224 rows: three m-tiles per wavefront in the second A half
            if (p.out_dtype == AV_F32) { if (nm3) go(gemm_nt_bf16_v7_kernel<AV_ACT_NONE, AV_F32, true, 3>); else go(gemm_nt_bf16_v7_kernel<AV_ACT_NONE, AV_F32, true, 4>); }
            else if (p.act == AV_ACT_NONE) { if (nm3) go(gemm_nt_bf16_v7_kernel<AV_ACT_NONE, AV_BF16, true, 3>); else go(gemm_nt_bf16_v7_kernel<AV_ACT_NONE, AV_BF16, true, 4>); }
            else if (p.act == AV_ACT_GELU) { if (nm3) go(gemm_nt_bf16_v7_kernel<AV_ACT_GELU, AV_BF16, true, 3>); else go(gemm_nt_bf16_v7_kernel<AV_ACT_GELU, AV_BF16, true, 4>); }
            else if (p.act == AV_ACT_GELU_GF) { if (nm3) go(gemm_nt_bf16_v7_kernel<AV_ACT_GELU_GF, AV_BF16, true, 3>); else go(gemm_nt_bf16_v7_kernel<AV_ACT_GELU_GF, AV_BF16, true, 4>); }
            else { if (nm3) go(gemm_nt_bf16_v7_kernel<AV_ACT_MUL_AUX, AV_BF16, true, 3>); else go(gemm_nt_bf16_v7_kernel<AV_ACT_MUL_AUX, AV_BF16, true, 4>); }
            if (rc != AV_OK) return rc;
            AV_LAUNCH_CHECK();
            return AV_OK;
        }
        if (v4_mode >= 2 || e4 < (v2_ok ? (e2 < e1 ? e2 : e1) : e1)) {
            static bool v4_attr = false;
            if (!v4_attr) {
                if (hipFuncSetAttribute((const void*)gemm_nt_bf16_v4_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, V4_LDS) != hipSuccess) {
                    av_set_error("av_gemm(fast v4): cannot raise dynamic LDS to %d", V4_LDS);
                    return AV_ERR_LAUNCH;
                }
                v4_attr = true;
            }
            const int nbM = av_cdiv(p.M, best_bm);
            const int ntile = nbM * nbN4;
            const int nfull = p.batch == 1 ? best_full : ntile;
            const unsigned nblocks = (unsigned)(nfull + 4 * (ntile - nfull));
            hipLaunchKernelGGL(gemm_nt_bf16_v4_kernel<false>, dim3(nblocks, 1, (unsigned)p.batch), dim3(V4_NT), V4_LDS, st, p, nbM, nbN4, fl, nfull, best_bm);
            AV_LAUNCH_CHECK();
            return AV_OK;
        }
    }
    if (v2_ok) {
        static bool v2_attr = false;
        if (!v2_attr) {
            if (hipFuncSetAttribute((const void*)gemm_nt_bf16_v2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, V2_LDS) != hipSuccess) {
                av_set_error("av_gemm(fast v2): cannot raise dynamic LDS to %d", V2_LDS);
                return AV_ERR_LAUNCH;
            }
            v2_attr = true;
        }
        const int nbM = av_cdiv(p.M, V2_BM), nbN = av_cdiv(p.N, V2_BN);
        hipLaunchKernelGGL(gemm_nt_bf16_v2_kernel, dim3((unsigned)(nbM * (long long)nbN), 1, (unsigned)p.batch), dim3(V2_NT), V2_LDS, st, p, nbM, nbN, fl);
        AV_LAUNCH_CHECK();
        return AV_OK;
    }
    if (conv) return narrow ? launch_fast<64, true>(p, st, fl) : launch_fast<128, true>(p, st, fl);
    return narrow ? launch_fast<64, false>(p, st, fl) : launch_fast<128, false>(p, st, fl);
}

#ifdef AV_GEMM_STAMPS
extern "C" int av_gemm_stamps_read(unsigned long long* host_out, int n_blocks) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_gemm_stamps), (size_t)n_blocks * 4 * sizeof(unsigned long long)) == hipSuccess ? 0 : 1;
}
extern "C" int av_gemm_stamps7_read(unsigned long long* host_out, int clear) {
    if (hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_v7_stamps), sizeof(unsigned long long) * 256 * 16 * 4) != hipSuccess) return 1;
    if (clear) { static unsigned long long z[256 * 16 * 4]; if (hipMemcpyToSymbol(HIP_SYMBOL(g_v7_stamps), z, sizeof(z)) != hipSuccess) return 1; }
    return 0;
}
#endif

extern "C" int av_transpose(const void* in, int idt, void* out, int odt, int R, int C, long long ldi, int Rpad, void* stream) {
    AV_CHECK(in && out && R > 0 && C > 0 && ldi >= C && Rpad >= R, "av_transpose: bad args R=%d C=%d ldi=%lld Rpad=%d", R, C, ldi, Rpad);
    dim3 grid((unsigned)((C + 63) / 64), (unsigned)((Rpad + 63) / 64));
    hipStream_t st = (hipStream_t)stream;
    if (idt == AV_BF16 && odt == AV_BF16) hipLaunchKernelGGL((transpose_kernel<bf16_t, bf16_t>), grid, dim3(256), 0, st, (const bf16_t*)in, (bf16_t*)out, R, C, ldi, Rpad);
    else if (idt == AV_F32 && odt == AV_BF16) hipLaunchKernelGGL((transpose_kernel<float, bf16_t>), grid, dim3(256), 0, st, (const float*)in, (bf16_t*)out, R, C, ldi, Rpad);
    else if (idt == AV_F32 && odt == AV_F32) hipLaunchKernelGGL((transpose_kernel<float, float>), grid, dim3(256), 0, st, (const float*)in, (float*)out, R, C, ldi, Rpad);
    else if (idt == AV_BF16 && odt == AV_F32) hipLaunchKernelGGL((transpose_kernel<bf16_t, float>), grid, dim3(256), 0, st, (const bf16_t*)in, (float*)out, R, C, ldi, Rpad);
    else { av_set_error("av_transpose: bad dtypes %d -> %d", idt, odt); return AV_ERR_ARG; }
    AV_LAUNCH_CHECK();
    return AV_OK;
}
