// MFMA GEMM family for gfx950: C = epilogue(alpha * A x B), fp32 accumulation.
//   T = float  -> v_mfma_f32_16x16x4_f32   (exact fp32, parity mode; 1/16 of the bf16 rate)
//   T = bf16   -> v_mfma_f32_16x16x32_bf16 (perf mode)
// Block tile 128 x BN x 32 (BN = 128 or 64), 256 threads = 4 wavefronts as 2(M) x 2(N), each wave owns
// 64 x BN/2 = 4 x (BN/32) MFMA tiles.  Global -> registers -> LDS staging, two LDS buffers, one barrier per
// K step; the global loads of step k+1 are in flight while step k runs on the matrix cores.
// Operand addressing is a template mode so that nn.Linear, strided conv1d, implicit-im2col conv2d/conv3d and
// the transposed products of the backward pass share one main loop (include/av_hip.h).
#include "av_common.h"

namespace {

constexpr int BM = 128;
constexpr int BK = 32;
constexpr int NT = 256;

template <typename T> struct Cfg;
template <> struct Cfg<float> { static constexpr int LD = 36, VEC = 4; };   // LDS row = 144 B
template <> struct Cfg<bf16_t> { static constexpr int LD = 40, VEC = 8; };  // LDS row = 80 B

template <typename T> struct Vec16 { uint4 v; };

template <typename T> __device__ __forceinline__ T elem_of(const uint4& v, int i);
template <> __device__ __forceinline__ float elem_of<float>(const uint4& v, int i) {
    const unsigned u = i == 0 ? v.x : i == 1 ? v.y : i == 2 ? v.z : v.w;
    return __uint_as_float(u);
}
template <> __device__ __forceinline__ bf16_t elem_of<bf16_t>(const uint4& v, int i) {
    const unsigned u = (i >> 1) == 0 ? v.x : (i >> 1) == 1 ? v.y : (i >> 1) == 2 ? v.z : v.w;
    const unsigned short s = (i & 1) ? (unsigned short)(u >> 16) : (unsigned short)(u & 0xffff);
    return __builtin_bit_cast(bf16_t, s);
}
template <typename T> __device__ __forceinline__ void set_elem(uint4& v, int i, T x);
template <> __device__ __forceinline__ void set_elem<float>(uint4& v, int i, float x) {
    const unsigned u = __float_as_uint(x);
    if (i == 0) v.x = u; else if (i == 1) v.y = u; else if (i == 2) v.z = u; else v.w = u;
}
template <> __device__ __forceinline__ void set_elem<bf16_t>(uint4& v, int i, bf16_t x) {
    const unsigned s = __builtin_bit_cast(unsigned short, x);
    unsigned* w = (i >> 1) == 0 ? &v.x : (i >> 1) == 1 ? &v.y : (i >> 1) == 2 ? &v.z : &v.w;
    *w = (i & 1) ? ((*w & 0x0000ffffu) | (s << 16)) : ((*w & 0xffff0000u) | s);
}

// guarded element-wise fill of one 16-byte chunk from `n_valid` contiguous elements at p
template <typename T> __device__ __forceinline__ uint4 load_partial(const T* p, int n_valid) {
    uint4 v = make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int i = 0; i < Cfg<T>::VEC; ++i)
        if (i < n_valid) set_elem<T>(v, i, p[i]);
    return v;
}

template <typename T, int WN_T>
__device__ __forceinline__ void mma_step(const T* __restrict__ sa, const T* __restrict__ sb, f32x4 (&acc)[4][WN_T], int lane);

template <int WN_T>
__device__ __forceinline__ void mma_step_bf16(const bf16_t* sa, const bf16_t* sb, f32x4 (&acc)[4][WN_T], int lane) {
    constexpr int LD = Cfg<bf16_t>::LD;
    const int r = lane & 15, g = lane >> 4;
    bf16x8 a[4], b[WN_T];
#pragma unroll
    for (int i = 0; i < 4; ++i) a[i] = *(const bf16x8*)(sa + (i * 16 + r) * LD + 8 * g);
#pragma unroll
    for (int j = 0; j < WN_T; ++j) b[j] = *(const bf16x8*)(sb + (j * 16 + r) * LD + 8 * g);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < WN_T; ++j)
            acc[i][j] = AV_MFMA_F32_16X16X32_LP(a[i], b[j], acc[i][j], 0, 0, 0);
}

template <int WN_T>
__device__ __forceinline__ void mma_step_f32(const float* sa, const float* sb, f32x4 (&acc)[4][WN_T], int lane) {
    constexpr int LD = Cfg<float>::LD;
    const int r = lane & 15, g = lane >> 4;
    // 16x16x4 takes ONE f32 per lane with k = lane>>4.  Each lane reads a float4 (k = 16h+4g .. +3) and feeds
    // element jj to MFMA jj, i.e. MFMA jj contracts over k = {16h + 4g' + jj : g' = 0..3}; A and B use the same
    // permutation, so the K sum is complete and exact fp32.
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        f32x4 a[4], b[WN_T];
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = *(const f32x4*)(sa + (i * 16 + r) * LD + 16 * h + 4 * g);
#pragma unroll
        for (int j = 0; j < WN_T; ++j) b[j] = *(const f32x4*)(sb + (j * 16 + r) * LD + 16 * h + 4 * g);
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < WN_T; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][jj], b[j][jj], acc[i][j], 0, 0, 0);
    }
}

template <> __device__ __forceinline__ void mma_step<bf16_t, 4>(const bf16_t* a, const bf16_t* b, f32x4 (&acc)[4][4], int l) { mma_step_bf16<4>(a, b, acc, l); }
template <> __device__ __forceinline__ void mma_step<bf16_t, 2>(const bf16_t* a, const bf16_t* b, f32x4 (&acc)[4][2], int l) { mma_step_bf16<2>(a, b, acc, l); }
template <> __device__ __forceinline__ void mma_step<float, 4>(const float* a, const float* b, f32x4 (&acc)[4][4], int l) { mma_step_f32<4>(a, b, acc, l); }
template <> __device__ __forceinline__ void mma_step<float, 2>(const float* a, const float* b, f32x4 (&acc)[4][2], int l) { mma_step_f32<2>(a, b, acc, l); }

// ---------------------------------------------------------------------------------------------------------
// operand stagers.  ROWS = tile rows (BM or BN); "K-contiguous" modes produce chunks [row][kc*VEC..],
// "transposed" modes produce chunks [k][m0..m0+VEC) that are scattered into the [row][k] LDS image.
// ---------------------------------------------------------------------------------------------------------
template <typename T, int ROWS> struct RowInfo {          // per-thread, per-chunk row bookkeeping for conv modes
    long long base[ROWS * (BK / Cfg<T>::VEC) / NT];
    int iy0[ROWS * (BK / Cfg<T>::VEC) / NT];
    int ix0[ROWS * (BK / Cfg<T>::VEC) / NT];
    int it0[ROWS * (BK / Cfg<T>::VEC) / NT];
};

template <typename T, int ROWS, int MODE>
__device__ __forceinline__ void stage_load(uint4 (&reg)[ROWS * (BK / Cfg<T>::VEC) / NT], const T* __restrict__ base,
                                           long long ld, int row0, int nrows, int k0, int K, bool vec_ok,
                                           const av_gemm_args& p, const RowInfo<T, ROWS>& ri, int tid) {
    constexpr int VEC = Cfg<T>::VEC;
    constexpr int CPR = BK / VEC;
    constexpr int NCH = ROWS * CPR / NT;
#pragma unroll
    for (int ci = 0; ci < NCH; ++ci) {
        const int c = tid + ci * NT;
        if constexpr (MODE == AV_A_ROWMAJOR) {
            const int row = c / CPR, kc = c % CPR;
            const int gr = row0 + row, gk = k0 + kc * VEC;
            if (gr < nrows && gk < K) {
                const T* ptr = base + (long long)gr * ld + gk;
                if (vec_ok && gk + VEC <= K) reg[ci] = *(const uint4*)ptr;
                else reg[ci] = load_partial<T>(ptr, K - gk);
            } else reg[ci] = make_uint4(0, 0, 0, 0);
        } else if constexpr (MODE == AV_A_TRANS) {
            constexpr int MPR = ROWS / VEC;
            const int kk = c / MPR, mm = (c % MPR) * VEC;
            const int gk = k0 + kk, gr = row0 + mm;
            if (gk < K && gr < nrows) {
                const T* ptr = base + (long long)gk * ld + gr;
                if (vec_ok && gr + VEC <= nrows) reg[ci] = *(const uint4*)ptr;
                else reg[ci] = load_partial<T>(ptr, nrows - gr);
            } else reg[ci] = make_uint4(0, 0, 0, 0);
        } else if constexpr (MODE == AV_A_CONV2D) {
            const int kc = c % CPR;
            const int gk = k0 + kc * VEC;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (ri.base[ci] >= 0 && gk < K) {
                const int tap = gk / p.cCin, cc = gk - tap * p.cCin;
                const int ky = tap / p.cKw, kx = tap - ky * p.cKw;
                const int iy = ri.iy0[ci] + ky, ix = ri.ix0[ci] + kx;
                if (iy >= 0 && iy < p.cH && ix >= 0 && ix < p.cW)
                    v = *(const uint4*)(base + ((ri.base[ci] + (long long)iy * p.cW + ix) * p.cCtot + p.cCoff + cc));
            }
            reg[ci] = v;
        } else {  // AV_A_CONV3D1: single-channel volume, element gather
            const int kc = c % CPR;
            const int gk = k0 + kc * VEC;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (ri.base[ci] >= 0) {
                const int khw = p.cKh * p.cKw;
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    const int k = gk + e;
                    if (k < K) {
                        const int kt = k / khw, rem = k - kt * khw;
                        const int ky = rem / p.cKw, kx = rem - ky * p.cKw;
                        const int it = ri.it0[ci] + kt, iy = ri.iy0[ci] + ky, ix = ri.ix0[ci] + kx;
                        if (it >= 0 && it < p.cT && iy >= 0 && iy < p.cH && ix >= 0 && ix < p.cW)
                            set_elem<T>(v, e, base[ri.base[ci] + ((long long)it * p.cH + iy) * p.cW + ix]);
                    }
                }
            }
            reg[ci] = v;
        }
    }
}

template <typename T, int ROWS, int MODE>
__device__ __forceinline__ void stage_store(T* __restrict__ s, const uint4 (&reg)[ROWS * (BK / Cfg<T>::VEC) / NT], int tid) {
    constexpr int VEC = Cfg<T>::VEC, LD = Cfg<T>::LD;
    constexpr int CPR = BK / VEC;
    constexpr int NCH = ROWS * CPR / NT;
#pragma unroll
    for (int ci = 0; ci < NCH; ++ci) {
        const int c = tid + ci * NT;
        if constexpr (MODE == AV_A_TRANS) {
            constexpr int MPR = ROWS / VEC;
            const int kk = c / MPR, mm = (c % MPR) * VEC;
#pragma unroll
            for (int e = 0; e < VEC; ++e) s[(mm + e) * LD + kk] = elem_of<T>(reg[ci], e);
        } else {
            const int row = c / CPR, kc = c % CPR;
            *(uint4*)(s + row * LD + kc * VEC) = reg[ci];
        }
    }
}

template <typename T, int ROWS, int MODE>
__device__ __forceinline__ void init_rows(RowInfo<T, ROWS>& ri, const av_gemm_args& p, int row0, int nrows, int tid) {
    constexpr int VEC = Cfg<T>::VEC;
    constexpr int CPR = BK / VEC;
    constexpr int NCH = ROWS * CPR / NT;
    if constexpr (MODE == AV_A_CONV2D || MODE == AV_A_CONV3D1) {
#pragma unroll
        for (int ci = 0; ci < NCH; ++ci) {
            const int c = tid + ci * NT;
            const int m = row0 + c / CPR;
            if (m < nrows) {
                const int ox = m % p.cOw;
                int q = m / p.cOw;
                const int oy = q % p.cOh;
                q /= p.cOh;                                 // q = img (2-D) or img*T + t (3-D)
                ri.iy0[ci] = oy * p.cSh - p.cPh;
                ri.ix0[ci] = ox * p.cSw - p.cPw;
                if constexpr (MODE == AV_A_CONV2D) {
                    ri.base[ci] = (long long)q * p.cH * p.cW;
                    ri.it0[ci] = 0;
                } else {
                    const int t = q % p.cT, img = q / p.cT;
                    ri.base[ci] = (long long)img * p.cT * p.cH * p.cW;
                    ri.it0[ci] = t - p.cPt;
                }
            } else {
                ri.base[ci] = -1; ri.iy0[ci] = 0; ri.ix0[ci] = 0; ri.it0[ci] = 0;
            }
        }
    }
}

template <typename T, int BN, int AMODE, int BMODE>
__global__ __launch_bounds__(NT) void gemm_kernel(const av_gemm_args p, const int nbN, const bool a_vec, const bool b_vec) {
    constexpr int LD = Cfg<T>::LD, VEC = Cfg<T>::VEC;
    constexpr int WN_T = BN / 32;
    constexpr int A_CH = BM * (BK / VEC) / NT, B_CH = BN * (BK / VEC) / NT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* sA = (T*)smem;                  // [2][BM][LD]
    T* sB = sA + 2 * BM * LD;          // [2][BN][LD]

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    const int mb = blockIdx.x / nbN, nb = blockIdx.x % nbN;
    const int m0 = mb * BM, n0 = nb * BN;
    const int z = blockIdx.z;
    const int zo = p.batch_inner > 0 ? z / p.batch_inner : 0;
    const int zi = p.batch_inner > 0 ? z % p.batch_inner : z;
    const T* A = (const T*)p.A + (long long)zo * p.oA + (long long)zi * p.sA;
    const T* B = (const T*)p.B + (long long)zo * p.oB + (long long)zi * p.sB;

    RowInfo<T, BM> ri;
    init_rows<T, BM, AMODE>(ri, p, m0, p.M, tid);
    RowInfo<T, BN> rib;   // unused for B (plain modes only)

    f32x4 acc[4][WN_T];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < WN_T; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    uint4 ra[A_CH], rb[B_CH];
    const int nk = (p.K + BK - 1) / BK;
    stage_load<T, BM, AMODE>(ra, A, p.lda, m0, p.M, 0, p.K, a_vec, p, ri, tid);
    stage_load<T, BN, BMODE == AV_B_NK ? AV_A_ROWMAJOR : AV_A_TRANS>(rb, B, p.ldb, n0, p.N, 0, p.K, b_vec, p, rib, tid);
    stage_store<T, BM, AMODE>(sA, ra, tid);
    stage_store<T, BN, BMODE == AV_B_NK ? AV_A_ROWMAJOR : AV_A_TRANS>(sB, rb, tid);
    __syncthreads();

    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) {
            stage_load<T, BM, AMODE>(ra, A, p.lda, m0, p.M, (kt + 1) * BK, p.K, a_vec, p, ri, tid);
            stage_load<T, BN, BMODE == AV_B_NK ? AV_A_ROWMAJOR : AV_A_TRANS>(rb, B, p.ldb, n0, p.N, (kt + 1) * BK, p.K, b_vec, p, rib, tid);
        }
        mma_step<T, WN_T>(sA + cur * BM * LD + wm * 64 * LD, sB + cur * BN * LD + wn * (BN / 2) * LD, acc, lane);
        if (kt + 1 < nk) {
            stage_store<T, BM, AMODE>(sA + (cur ^ 1) * BM * LD, ra, tid);
            stage_store<T, BN, BMODE == AV_B_NK ? AV_A_ROWMAJOR : AV_A_TRANS>(sB + (cur ^ 1) * BN * LD, rb, tid);
        }
        __syncthreads();
    }

    // ---------------- epilogue ----------------
    const int r = lane & 15, g = lane >> 4;
    const long long cbase = (long long)zo * p.oC + (long long)zi * p.sC;
    const float* R = p.R ? p.R + (long long)zi * p.sR : nullptr;
    float csum[WN_T], csq[WN_T];
#pragma unroll
    for (int j = 0; j < WN_T; ++j) { csum[j] = 0.f; csq[j] = 0.f; }
#pragma unroll
    for (int j = 0; j < WN_T; ++j) {
        const int n = n0 + wn * (BN / 2) + j * 16 + r;
        const bool nok = n < p.N;
        const float bias = (p.bias && nok) ? p.bias[(long long)zi * p.sBias + n] : 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int m = m0 + wm * 64 + i * 16 + 4 * g + e;
                if (nok && m < p.M) {
                    float v = acc[i][j][e] * p.alpha + bias;
                    const long long off = cbase + (long long)m * p.ldc + n;
                    if (p.act == AV_ACT_GELU_GF) {               // C2 = gelu'(v) * m, v = gelu(v) * m
                        const float mlt = p.drop_p > 0.f ? drop_mult_call(p.drop_seed, p.drop_stream, (unsigned long long)off, p.drop_p, drop_inv_keep(p.drop_p)) : 1.f;
                        if (p.C2) st_any(p.C2, off, p.out_dtype, gelu_grad_f(v) * mlt);
                        v = gelu_f(v) * mlt;
                    } else {
                        if (p.C2) st_any(p.C2, off, p.out_dtype, v);
                        if (p.act == AV_ACT_GELU) v = gelu_f(v);
                        else if (p.act == AV_ACT_MUL_GELU_GRAD) v *= gelu_grad_f(ld_any(p.aux, off, p.aux_dtype));
                        else if (p.act == AV_ACT_MUL_AUX) v *= ld_any(p.aux, off, p.aux_dtype);
                        if (p.drop_p > 0.f) v *= drop_mult_call(p.drop_seed, p.drop_stream, (unsigned long long)off, p.drop_p, drop_inv_keep(p.drop_p));
                    }
                    if (R) v += R[(long long)m * p.ldr + n];
                    st_any(p.C, off, p.out_dtype, v);
                    csum[j] += v; csq[j] += v * v;
                }
            }
        }
    }
    if (p.stats) {   // per-column partial sums over this block's rows (train-mode BatchNorm statistics)
        __syncthreads();                 // all MFMA reads of LDS are done; reuse it
        float* red = (float*)smem;       // [2 wm][2][BN]
#pragma unroll
        for (int j = 0; j < WN_T; ++j) {
            float s = csum[j], q = csq[j];
            s += __shfl_xor(s, 16, 64); s += __shfl_xor(s, 32, 64);
            q += __shfl_xor(q, 16, 64); q += __shfl_xor(q, 32, 64);
            if (g == 0) {
                const int col = wn * (BN / 2) + j * 16 + r;
                red[(wm * 2 + 0) * BN + col] = s;
                red[(wm * 2 + 1) * BN + col] = q;
            }
        }
        __syncthreads();
        if (tid < BN && n0 + tid < p.N) {
            float* out = p.stats + (long long)mb * 2 * p.N;
            out[n0 + tid] = red[0 * BN + tid] + red[2 * BN + tid];
            out[p.N + n0 + tid] = red[1 * BN + tid] + red[3 * BN + tid];
        }
    }
}

template <typename T, int BN, int AMODE, int BMODE>
int launch(const av_gemm_args& p, hipStream_t st, bool a_vec, bool b_vec) {
    const size_t lds = (size_t)(2 * BM + 2 * BN) * Cfg<T>::LD * sizeof(T);
    static bool attr_done = false;
    auto kern = gemm_kernel<T, BN, AMODE, BMODE>;
    if (!attr_done && lds > 48 * 1024) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
            av_set_error("av_gemm: cannot raise dynamic LDS to %zu", lds);
            return AV_ERR_LAUNCH;
        }
        attr_done = true;
    }
    const int nbM = av_cdiv(p.M, BM), nbN = av_cdiv(p.N, BN);
    dim3 grid((unsigned)(nbM * (long long)nbN), 1, (unsigned)p.batch);
    hipLaunchKernelGGL(kern, grid, dim3(NT), lds, st, p, nbN, a_vec, b_vec);
    AV_LAUNCH_CHECK();
    return AV_OK;
}

template <typename T, int BN>
int dispatch_modes(const av_gemm_args& p, hipStream_t st, bool a_vec, bool b_vec) {
    if (p.a_mode == AV_A_ROWMAJOR && p.b_mode == AV_B_NK) return launch<T, BN, AV_A_ROWMAJOR, AV_B_NK>(p, st, a_vec, b_vec);
    if (p.a_mode == AV_A_ROWMAJOR && p.b_mode == AV_B_KN) return launch<T, BN, AV_A_ROWMAJOR, AV_B_KN>(p, st, a_vec, b_vec);
    if (p.a_mode == AV_A_TRANS && p.b_mode == AV_B_KN) return launch<T, BN, AV_A_TRANS, AV_B_KN>(p, st, a_vec, b_vec);
    if (p.a_mode == AV_A_CONV2D && p.b_mode == AV_B_NK) return launch<T, BN, AV_A_CONV2D, AV_B_NK>(p, st, a_vec, b_vec);
    if (p.a_mode == AV_A_CONV3D1 && p.b_mode == AV_B_NK) return launch<T, BN, AV_A_CONV3D1, AV_B_NK>(p, st, a_vec, b_vec);
    av_set_error("av_gemm: unsupported operand mode pair a_mode=%d b_mode=%d", p.a_mode, p.b_mode);
    return AV_ERR_ARG;
}

template <typename T>
bool vec_ok(const void* base, long long ld, long long stride) {
    const long long es = sizeof(T);
    return ((uintptr_t)base % 16 == 0) && ((ld * es) % 16 == 0) && ((stride * es) % 16 == 0);
}
template <typename T> bool vec_ok2(const void* base, long long ld, long long s1, long long s2) {
    return vec_ok<T>(base, ld, s1) && ((s2 * (long long)sizeof(T)) % 16 == 0);
}

}  // namespace

int av_gemm_fast_try(const av_gemm_args& p, hipStream_t st);   // gemm_fast.hip; -1 = not eligible

extern "C" int av_gemm(const av_gemm_args* a, void* stream) {
    AV_CHECK(a != nullptr, "av_gemm: null args");
    const av_gemm_args& p = *a;
    AV_CHECK(p.A && p.B && p.C, "av_gemm: null operand pointer");
    AV_CHECK(p.M >= 0 && p.N >= 0 && p.K > 0 && p.batch >= 1, "av_gemm: bad shape M=%d N=%d K=%d batch=%d", p.M, p.N, p.K, p.batch);
    AV_CHECK(p.in_dtype == AV_F32 || p.in_dtype == AV_BF16, "av_gemm: bad in_dtype %d", p.in_dtype);
    AV_CHECK(p.out_dtype == AV_F32 || p.out_dtype == AV_BF16, "av_gemm: bad out_dtype %d", p.out_dtype);
    AV_CHECK(!(p.stats && p.batch != 1), "av_gemm: stats need batch == 1");
    AV_CHECK(!((p.act == AV_ACT_MUL_GELU_GRAD || p.act == AV_ACT_MUL_AUX) && !p.aux), "av_gemm: MUL_GELU_GRAD / MUL_AUX need aux");
    AV_CHECK(p.act >= AV_ACT_NONE && p.act <= AV_ACT_MUL_AUX, "av_gemm: unknown activation %d", p.act);
    AV_CHECK(p.drop_p >= 0.f && p.drop_p < 1.f, "av_gemm: drop_p=%f out of [0,1)", p.drop_p);
    if (p.M == 0 || p.N == 0) return AV_OK;
    const long long es = p.in_dtype == AV_F32 ? 4 : 2;
    const int vec = (int)(16 / es);
    if (p.a_mode == AV_A_CONV2D) {
        AV_CHECK(p.cCin % vec == 0 && p.cCoff % vec == 0 && p.cCtot % vec == 0 && (uintptr_t)p.A % 16 == 0 && (p.sA * es) % 16 == 0,
                 "av_gemm conv2d: channels (Cin=%d, Coff=%d, Ctot=%d) must be multiples of %d and the base 16-byte aligned", p.cCin, p.cCoff, p.cCtot, vec);
        AV_CHECK(p.K == p.cKh * p.cKw * p.cCin, "av_gemm conv2d: K=%d != Kh*Kw*Cin", p.K);
        AV_CHECK(p.cOh > 0 && p.cOw > 0 && p.M % (p.cOh * p.cOw) == 0, "av_gemm conv2d: M=%d not a multiple of Oh*Ow", p.M);
    }
    if (p.a_mode == AV_A_CONV3D1) {
        AV_CHECK(p.K == p.cKt * p.cKh * p.cKw, "av_gemm conv3d: K=%d != Kt*Kh*Kw", p.K);
        AV_CHECK(p.cOh > 0 && p.cOw > 0 && p.cT > 0 && p.M % (p.cOh * p.cOw * p.cT) == 0, "av_gemm conv3d: bad M=%d", p.M);
    }
    hipStream_t st = (hipStream_t)stream;
    {
        const int fr = av_gemm_fast_try(p, st);
        if (fr >= 0) return fr;
    }
    AV_CHECK(p.cPM == 0, "av_gemm: position-major pixel order (cPM) needs the bf16 fast path (2-D convolution, Cin a multiple of 64, one batch)");
    AV_CHECK(p.k_total == 0, "av_gemm: split-K (k_total) needs the bf16 fast path with both operands k-major (a_mode 1, b_mode 1, M, N multiples of 8 and > 64)");
    const bool wide = p.N > 64;
    if (p.in_dtype == AV_F32) {
        const bool av = vec_ok2<float>(p.A, p.lda, p.sA, p.oA), bv = vec_ok2<float>(p.B, p.ldb, p.sB, p.oB);
        return wide ? dispatch_modes<float, 128>(p, st, av, bv) : dispatch_modes<float, 64>(p, st, av, bv);
    } else {
        const bool av = vec_ok2<bf16_t>(p.A, p.lda, p.sA, p.oA), bv = vec_ok2<bf16_t>(p.B, p.ldb, p.sB, p.oB);
        return wide ? dispatch_modes<bf16_t, 128>(p, st, av, bv) : dispatch_modes<bf16_t, 64>(p, st, av, bv);
    }
}
