// Row-wise kernels: one 64-lane wavefront per row, shuffle reductions (no LDS for the row reduce).
// LayerNorm fwd/bwd (+GELU), log_softmax fwd/bwd, column sums (bias gradients), casts, axpby, row masking.
// All HBM-bound: each element is read once (rows are cached in registers between the passes).
#include "av_common.h"

namespace {

constexpr int MAXIT = 32;  // 64 lanes x 32 = rows up to 2048 columns

template <int ACT, int NIT>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const void* __restrict__ x, int xdt, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, void* __restrict__ y, int ydt,
                                                     float* __restrict__ mean_o, float* __restrict__ rstd_o,
                                                     long long rows, int cols, float eps) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const long long base = row * cols;
    float v[NIT];
    float s = 0.f;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int c = lane + 64 * it;
        v[it] = c < cols ? ld_any(x, base + c, xdt) : 0.f;
        s += v[it];
    }
    const float mean = wave_sum(s) / cols;
    float q = 0.f;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int c = lane + 64 * it;
        const float d = c < cols ? v[it] - mean : 0.f;
        q += d * d;
    }
    const float rstd = rsqrtf(wave_sum(q) / cols + eps);
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int c = lane + 64 * it;
        if (c < cols) {
            float o = (v[it] - mean) * rstd * gamma[c] + beta[c];
            if (ACT == AV_ACT_GELU) o = gelu_f(o);
            st_any(y, base + c, ydt, o);
        }
    }
    if (lane == 0) {
        if (mean_o) mean_o[row] = mean;
        if (rstd_o) rstd_o[row] = rstd;
    }
}

// dx = dres + rstd * (g - mean(g) - xhat * mean(g*xhat)),  g = dy*gamma;  partial dgamma/dbeta per block
template <int NIT, bool PG>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const void* __restrict__ x, int xdt, const void* __restrict__ dy, int dydt,
                                                     const float* __restrict__ gamma, const float* __restrict__ mean,
                                                     const float* __restrict__ rstd, const float* __restrict__ dres,
                                                     float* __restrict__ dx, float* __restrict__ dgb, long long rows, int cols,
                                                     long long rows_per_block) {
    __shared__ float red[4][2][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const long long r_begin = (long long)blockIdx.x * rows_per_block;
    long long r_end = r_begin + rows_per_block;
    if (r_end > rows) r_end = rows;
    float dg[PG ? NIT : 1], db[PG ? NIT : 1];
#pragma unroll
    for (int it = 0; it < (PG ? NIT : 1); ++it) { dg[it] = 0.f; db[it] = 0.f; }
    for (long long row = r_begin + w; row < r_end; row += 4) {
        const long long base = row * cols;
        const float mu = mean[row], rs = rstd[row];
        float xh[NIT], g[NIT];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int c = lane + 64 * it;
            if (c < cols) {
                const float d = ld_any(dy, base + c, dydt);
                xh[it] = (ld_any(x, base + c, xdt) - mu) * rs;
                g[it] = d * gamma[c];
                if (PG) { dg[it] += d * xh[it]; db[it] += d; }
                s1 += g[it];
                s2 += g[it] * xh[it];
            } else { xh[it] = 0.f; g[it] = 0.f; }
        }
        s1 = wave_sum(s1) / cols;
        s2 = wave_sum(s2) / cols;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int c = lane + 64 * it;
            if (c < cols) {
                float o = rs * (g[it] - s1 - xh[it] * s2);
                if (dres) o += dres[base + c];
                dx[base + c] = o;
            }
        }
    }
    if (PG && dgb) {   // combine the 4 waves, write [blk][2][cols]
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            if (64 * it >= cols) break;
            red[w][0][lane] = dg[PG ? it : 0];
            red[w][1][lane] = db[PG ? it : 0];
            __syncthreads();
            if (w == 0) {
                const int c = lane + 64 * it;
                if (c < cols) {
                    float* o = dgb + (long long)blockIdx.x * 2 * cols;
                    o[c] = red[0][0][lane] + red[1][0][lane] + red[2][0][lane] + red[3][0][lane];
                    o[cols + c] = red[0][1][lane] + red[1][1][lane] + red[2][1][lane] + red[3][1][lane];
                }
            }
            __syncthreads();
        }
    }
}


// ---- vector forms for cols % 256 == 0 (the 512 / 1024-wide rows of wav2vec2): lane owns 4 consecutive elements per
// 256-column slab -> 16-B fp32 / 8-B bf16 accesses ------------------------------------------------------------------
__device__ __forceinline__ f32x4 ld4(const void* p, long long i, int dtype) {
    if (dtype == AV_F32) return *(const f32x4*)((const float*)p + i);
    const bf16x4 v = *(const bf16x4*)((const bf16_t*)p + i);
    return f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
}
__device__ __forceinline__ void st4(void* p, long long i, int dtype, const f32x4& v) {
    if (dtype == AV_F32) *(f32x4*)((float*)p + i) = v;
    else {
        bf16x4 o; o[0] = (bf16_t)v[0]; o[1] = (bf16_t)v[1]; o[2] = (bf16_t)v[2]; o[3] = (bf16_t)v[3];
        *(bf16x4*)((bf16_t*)p + i) = o;
    }
}

template <int ACT, int NV>
__global__ __launch_bounds__(256) void ln_fwd_vec_kernel(const void* __restrict__ x, int xdt, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, void* __restrict__ y, int ydt,
                                                         float* __restrict__ mean_o, float* __restrict__ rstd_o, long long rows, float eps) {
    constexpr int cols = NV * 256;
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const long long base = row * cols;
    f32x4 v[NV];
    float s = 0.f;
#pragma unroll
    for (int it = 0; it < NV; ++it) { v[it] = ld4(x, base + it * 256 + lane * 4, xdt); s += v[it][0] + v[it][1] + v[it][2] + v[it][3]; }
    const float mean = wave_sum(s) / cols;
    float q = 0.f;
#pragma unroll
    for (int it = 0; it < NV; ++it)
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float d = v[it][e] - mean; q += d * d; }
    const float rstd = rsqrtf(wave_sum(q) / cols + eps);
#pragma unroll
    for (int it = 0; it < NV; ++it) {
        const int c = it * 256 + lane * 4;
        const f32x4 g = *(const f32x4*)(gamma + c), b = *(const f32x4*)(beta + c);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float t = (v[it][e] - mean) * rstd * g[e] + b[e];
            if (ACT == AV_ACT_GELU) t = ydt == AV_BF16 ? gelu_fast(t) : gelu_f(t);
            o[e] = t;
        }
        st4(y, base + c, ydt, o);
    }
    if (lane == 0) {
        if (mean_o) mean_o[row] = mean;
        if (rstd_o) rstd_o[row] = rstd;
    }
}

template <int NV, bool PG>
__global__ __launch_bounds__(256) void ln_bwd_vec_kernel(const void* __restrict__ x, int xdt, const void* __restrict__ dy, int dydt,
                                                         const float* __restrict__ gamma, const float* __restrict__ mean,
                                                         const float* __restrict__ rstd, const float* __restrict__ dres,
                                                         float* __restrict__ dx, float* __restrict__ dgb, long long rows,
                                                         long long rows_per_block, bf16_t* __restrict__ dxl, float drop_p,
                                                         unsigned long long drop_seed, unsigned drop_stream) {
    constexpr int cols = NV * 256;
    const float drop_ik = drop_p > 0.f ? drop_inv_keep(drop_p) : 1.0f;     // the forward site scaled its survivors by the same factor
    __shared__ float red[4][2][256];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const long long r_begin = (long long)blockIdx.x * rows_per_block;
    long long r_end = r_begin + rows_per_block;
    if (r_end > rows) r_end = rows;
    f32x4 gam[NV];
#pragma unroll
    for (int it = 0; it < NV; ++it) gam[it] = *(const f32x4*)(gamma + it * 256 + lane * 4);
    f32x4 dg[PG ? NV : 1], db[PG ? NV : 1];
#pragma unroll
    for (int it = 0; it < (PG ? NV : 1); ++it) { dg[it] = f32x4{0.f, 0.f, 0.f, 0.f}; db[it] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    for (long long row = r_begin + w; row < r_end; row += 4) {
        const long long base = row * cols;
        const float mu = mean[row], rs = rstd[row];
        f32x4 xh[NV], g[NV];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int it = 0; it < NV; ++it) {
            const f32x4 d = ld4(dy, base + it * 256 + lane * 4, dydt);
            const f32x4 xv = ld4(x, base + it * 256 + lane * 4, xdt);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                xh[it][e] = (xv[e] - mu) * rs;
                g[it][e] = d[e] * gam[it][e];
                if (PG) { dg[it][e] += d[e] * xh[it][e]; db[it][e] += d[e]; }
                s1 += g[it][e];
                s2 += g[it][e] * xh[it][e];
            }
        }
        s1 = wave_sum(s1) / cols;
        s2 = wave_sum(s2) / cols;
#pragma unroll
        for (int it = 0; it < NV; ++it) {
            const long long off = base + it * 256 + lane * 4;
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = rs * (g[it][e] - s1 - xh[it][e] * s2);
            if (dres) {
                const f32x4 rr = *(const f32x4*)(dres + off);
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] += rr[e];
            }
            *(f32x4*)(dx + off) = o;
            if (dxl) {                                       // bf16 copy for the GEMM that consumes dx next (saves a cast pass); with
                bf16x4 ol;                                   // drop_p > 0 the copy is dx o mask / (1 - p): the dropout backward of that GEMM's input
                if (drop_p > 0.f) {
                    float m4[4];
                    drop_mult4(drop_seed, drop_stream, (unsigned long long)off, drop_p, drop_ik, m4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) ol[e] = (bf16_t)(o[e] * m4[e]);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) ol[e] = (bf16_t)o[e];
                }
                *(bf16x4*)(dxl + off) = ol;
            }
        }
    }
    if (PG && dgb) {
#pragma unroll
        for (int it = 0; it < NV; ++it) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { red[w][0][lane * 4 + e] = dg[PG ? it : 0][e]; red[w][1][lane * 4 + e] = db[PG ? it : 0][e]; }
            __syncthreads();
            {
                const int c = threadIdx.x;       // 256 threads <-> the 256 columns of this slab
                float* o = dgb + (long long)blockIdx.x * 2 * cols;
                o[it * 256 + c] = red[0][0][c] + red[1][0][c] + red[2][0][c] + red[3][0][c];
                o[cols + it * 256 + c] = red[0][1][c] + red[1][1][c] + red[2][1][c] + red[3][1][c];
            }
            __syncthreads();
        }
    }
}

__global__ __launch_bounds__(256) void lsm_fwd_kernel(const void* __restrict__ x, int xdt, float* __restrict__ y, long long rows, int cols) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const long long base = row * cols;
    float v[MAXIT];
    float mx = -INFINITY;
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
        const int c = lane + 64 * it;
        v[it] = c < cols ? ld_any(x, base + c, xdt) : -INFINITY;
        mx = fmaxf(mx, v[it]);
    }
    mx = wave_max(mx);
    float s = 0.f;
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
        const int c = lane + 64 * it;
        if (c < cols) s += expf(v[it] - mx);
    }
    const float lse = mx + logf(wave_sum(s));
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
        const int c = lane + 64 * it;
        if (c < cols) y[base + c] = v[it] - lse;
    }
}

__global__ __launch_bounds__(256) void lsm_bwd_kernel(const float* __restrict__ y, const float* __restrict__ dy, void* __restrict__ dx,
                                                      int dxdt, long long rows, int cols, int ldx) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const long long base = row * cols, obase = row * ldx;
    float d[MAXIT];
    float s = 0.f;
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
        const int c = lane + 64 * it;
        d[it] = c < cols ? dy[base + c] : 0.f;
        s += d[it];
    }
    s = wave_sum(s);
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
        const int c = lane + 64 * it;
        if (c < cols) st_any(dx, obase + c, dxdt, d[it] - expf(y[base + c]) * s);
        else if (c < ldx) st_any(dx, obase + c, dxdt, 0.f);          // padding columns of a K-padded GEMM operand: exact zeros
    }
}

constexpr int CS_ROWS = 512;
__global__ __launch_bounds__(256) void colsum_kernel(const void* __restrict__ x, int xdt, float* __restrict__ out, long long rows,
                                                     int cols, long long ld) {
    __shared__ float red[4][64];
    const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cx;
    const long long r0 = (long long)blockIdx.y * CS_ROWS;
    long long r1 = r0 + CS_ROWS;
    if (r1 > rows) r1 = rows;
    float s = 0.f;
    if (c < cols)
        for (long long r = r0 + ry; r < r1; r += 4) s += ld_any(x, r * ld + c, xdt);
    red[ry][cx] = s;
    __syncthreads();
    if (ry == 0 && c < cols) atomicAdd(out + c, red[0][cx] + red[1][cx] + red[2][cx] + red[3][cx]);
}

// bf16 rows, 16-B loads.  The kernel is bound by its atomics, not by its reads (measured: halving the rows per block doubled the atomic
// rounds per output element and the time, 24 -> 42 us on [12736 x 1024]): a block therefore covers only 128 columns (a 16-lane group
// reads 256 contiguous bytes of one row, a wavefront 4 rows, the 4 wavefronts 16 rows per step) but 256 rows, so an output element sees
// rows / 256 atomic adds while [12736 x 1024] still spreads over 8 x 50 = 400 blocks with 8 loads in flight per lane.
constexpr int CSV_ROWS = 256;
constexpr int CSVB_COLS = 128;
__global__ __launch_bounds__(256) void colsum_bf16_vec_kernel(const bf16_t* __restrict__ x, float* __restrict__ out, long long rows, int cols, long long ld) {
    __shared__ float red[3][16][8];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, cg = lane & 15, rs = lane >> 4;
    const int c = blockIdx.x * CSVB_COLS + cg * 8;
    const long long r0 = (long long)blockIdx.y * CSV_ROWS;
    long long r1 = r0 + CSV_ROWS;
    if (r1 > rows) r1 = rows;
    float s[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s[e] = 0.f;
    if (c < cols) {
        const bf16_t* px = x + c;
#pragma unroll 8
        for (long long r = r0 + w * 4 + rs; r < r1; r += 16) {
            const bf16x8 v = *(const bf16x8*)(px + r * ld);
#pragma unroll
            for (int e = 0; e < 8; ++e) s[e] += (float)v[e];
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {                              // the four row sub-lanes of a wavefront
        s[e] += __shfl_xor(s[e], 16, 64);
        s[e] += __shfl_xor(s[e], 32, 64);
    }
    if (w > 0 && rs == 0) {
#pragma unroll
        for (int e = 0; e < 8; ++e) red[w - 1][cg][e] = s[e];
    }
    __syncthreads();
    if (w == 0 && rs == 0 && c < cols) {
#pragma unroll
        for (int e = 0; e < 8; ++e) atomicAdd(out + c + e, s[e] + red[0][cg][e] + red[1][cg][e] + red[2][cg][e]);
    }
}

// fp32 rows, 16-B loads: a wavefront covers 256 consecutive columns of one row
constexpr int CSVF_ROWS = 128;
__global__ __launch_bounds__(256) void colsum_f32_vec_kernel(const float* __restrict__ x, float* __restrict__ out, long long rows, int cols, long long ld) {
    __shared__ float red[3][64][4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c = blockIdx.x * 256 + lane * 4;
    const long long r0 = (long long)blockIdx.y * CSVF_ROWS;
    long long r1 = r0 + CSVF_ROWS;
    if (r1 > rows) r1 = rows;
    f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
    if (c < cols) {
        const float* px = x + c;
#pragma unroll 4
        for (long long r = r0 + w; r < r1; r += 4) s += *(const f32x4*)(px + r * ld);
    }
    if (w > 0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) red[w - 1][lane][e] = s[e];
    }
    __syncthreads();
    if (w == 0 && c < cols) {
#pragma unroll
        for (int e = 0; e < 4; ++e) atomicAdd(out + c + e, s[e] + red[0][lane][e] + red[1][lane][e] + red[2][lane][e]);
    }
}

__global__ void sum_slices_kernel(const float* __restrict__ parts, int S, long long n, long long stride, float alpha, float* __restrict__ out,
                                  int accumulate, int vec) {
    if (vec) {
        const long long n4 = n / 4;
        for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
            f32x4 s = *(const f32x4*)(parts + 4 * i);
            for (int k = 1; k < S; ++k) s += *(const f32x4*)(parts + (long long)k * stride + 4 * i);     // fixed order: reproducible
            s *= alpha;
            if (accumulate) s += *(const f32x4*)(out + 4 * i);
            *(f32x4*)(out + 4 * i) = s;
        }
    } else {
        for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
            float s = parts[i];
            for (int k = 1; k < S; ++k) s += parts[(long long)k * stride + i];
            s *= alpha;
            out[i] = accumulate ? out[i] + s : s;
        }
    }
}

__global__ void cast_kernel(const void* __restrict__ x, int xdt, void* __restrict__ y, int ydt, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        st_any(y, i, ydt, ld_any(x, i, xdt));
}
__global__ void cast_f32_bf16_vec(const float4* __restrict__ x, bf16x4* __restrict__ y, long long n4) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        const float4 v = x[i];
        bf16x4 o; o[0] = (bf16_t)v.x; o[1] = (bf16_t)v.y; o[2] = (bf16_t)v.z; o[3] = (bf16_t)v.w;
        y[i] = o;
    }
}
__global__ void axpby_kernel(float a, const void* __restrict__ x, int xdt, float b, float* __restrict__ y, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        y[i] = a * ld_any(x, i, xdt) + (b == 0.f ? 0.f : b * y[i]);
}
__global__ void mask_rows_kernel(void* __restrict__ x, int xdt, const unsigned char* __restrict__ keep, long long rows, int cols) {
    const long long n = rows * cols;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        if (!keep[i / cols]) st_any(x, i, xdt, 0.f);
}

__global__ void mul_scalar_dev_kernel(const float* __restrict__ x, const float* __restrict__ sc, float* __restrict__ y, long long n) {
    const float s = sc[0];
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) y[i] = s * x[i];
}

__global__ void cast_dropout_kernel(const void* __restrict__ x, int xdt, void* __restrict__ y, int ydt, long long n, float p, float inv_keep,
                                    unsigned long long seed, unsigned stream_id) {
    const long long n4 = (n + 3) / 4;
    for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x; q < n4; q += (long long)gridDim.x * blockDim.x) {
        float m[4];
        drop_mult4(seed, stream_id, (unsigned long long)q * 4, p, inv_keep, m);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const long long i = q * 4 + e;
            if (i < n) st_any(y, i, ydt, ld_any(x, i, xdt) * m[e]);
        }
    }
}
__global__ void dropout_uniform_kernel(float* __restrict__ u, long long n, unsigned long long seed, unsigned stream_id) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        u[i] = drop_uniform(seed, stream_id, (unsigned long long)i);
}
__global__ void overwrite_rows_kernel(void* __restrict__ x, int xdt, const unsigned char* __restrict__ mask, const float* __restrict__ embed,
                                      long long rows, int cols) {
    const long long n = rows * cols;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        if (mask[i / cols]) st_any(x, i, xdt, embed[i % cols]);
}

// SpecAugment along the feature axis (hf:1298-1316): x[b, t, c] = 0 for every t where mask[b, c] is set
__global__ void zero_feature_cols_kernel(void* __restrict__ x, int xdt, const unsigned char* __restrict__ mask, long long n, int T, int H) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const long long b = i / ((long long)T * H);
        if (mask[b * H + (i % H)]) st_any(x, i, xdt, 0.f);
    }
}

inline int ew_grid(long long n) {
    long long b = (n + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 2048 ? 2048 : b));
}

}  // namespace

extern "C" int av_layernorm_fwd(const void* x, int xdt, const float* gamma, const float* beta, void* y, int ydt, float* mean,
                                float* rstd, long long rows, int cols, float eps, int act, void* stream) {
    AV_CHECK(x && gamma && beta && y, "av_layernorm_fwd: null pointer");
    AV_CHECK(cols > 0 && cols <= 64 * MAXIT, "av_layernorm_fwd: cols=%d out of range (1..%d)", cols, 64 * MAXIT);
    AV_CHECK(act == AV_ACT_NONE || act == AV_ACT_GELU, "av_layernorm_fwd: bad act %d", act);
    if (rows == 0) return AV_OK;
    dim3 grid((unsigned)((rows + 3) / 4));
    const bool xa = xdt == AV_F32 ? ((uintptr_t)x % 16 == 0) : ((uintptr_t)x % 8 == 0);
    const bool ya = ydt == AV_F32 ? ((uintptr_t)y % 16 == 0) : ((uintptr_t)y % 8 == 0);
    if (cols % 256 == 0 && cols <= 2048 && xa && ya && (uintptr_t)gamma % 16 == 0 && (uintptr_t)beta % 16 == 0) {
#define LNV(A, N) hipLaunchKernelGGL((ln_fwd_vec_kernel<A, N>), grid, dim3(256), 0, (hipStream_t)stream, x, xdt, gamma, beta, y, ydt, mean, rstd, rows, eps)
#define LNV_N(A) do { switch (cols / 256) { case 1: LNV(A, 1); break; case 2: LNV(A, 2); break; case 3: LNV(A, 3); break; case 4: LNV(A, 4); break; \
                                           case 5: LNV(A, 5); break; case 6: LNV(A, 6); break; case 7: LNV(A, 7); break; default: LNV(A, 8); } } while (0)
        if (act == AV_ACT_GELU) LNV_N(AV_ACT_GELU); else LNV_N(AV_ACT_NONE);
#undef LNV_N
#undef LNV
        AV_LAUNCH_CHECK();
        return AV_OK;
    }
#define LN_FWD(A, N) hipLaunchKernelGGL((ln_fwd_kernel<A, N>), grid, dim3(256), 0, (hipStream_t)stream, x, xdt, gamma, beta, y, ydt, mean, rstd, rows, cols, eps)
#define LN_FWD_N(A) do { if (cols <= 64) LN_FWD(A, 1); else if (cols <= 128) LN_FWD(A, 2); else if (cols <= 512) LN_FWD(A, 8); \
                         else if (cols <= 1024) LN_FWD(A, 16); else LN_FWD(A, 32); } while (0)
    if (act == AV_ACT_GELU) LN_FWD_N(AV_ACT_GELU); else LN_FWD_N(AV_ACT_NONE);
#undef LN_FWD_N
#undef LN_FWD
    AV_LAUNCH_CHECK();
    return AV_OK;
}

extern "C" int av_layernorm_bwd_drop(const void* x, int xdt, const void* dy, int dydt, const float* gamma, const float* mean,
                                     const float* rstd, const float* dres, float* dx, float* dgb_partial, int nblk, long long rows,
                                     int cols, void* dx_bf16, float drop_p, unsigned long long drop_seed, unsigned int drop_stream, void* stream);
extern "C" int av_layernorm_bwd(const void* x, int xdt, const void* dy, int dydt, const float* gamma, const float* mean,
                                const float* rstd, const float* dres, float* dx, float* dgb_partial, int nblk, long long rows,
                                int cols, void* dx_bf16, void* stream) {
    return av_layernorm_bwd_drop(x, xdt, dy, dydt, gamma, mean, rstd, dres, dx, dgb_partial, nblk, rows, cols, dx_bf16, 0.f, 0ull, 0u, stream);
}
extern "C" int av_layernorm_bwd_drop(const void* x, int xdt, const void* dy, int dydt, const float* gamma, const float* mean,
                                     const float* rstd, const float* dres, float* dx, float* dgb_partial, int nblk, long long rows,
                                     int cols, void* dx_bf16, float drop_p, unsigned long long drop_seed, unsigned int drop_stream, void* stream) {
    AV_CHECK(x && dy && gamma && mean && rstd && dx, "av_layernorm_bwd: null pointer");
    AV_CHECK(drop_p >= 0.f && drop_p < 1.f && (drop_p == 0.f || dx_bf16), "av_layernorm_bwd: drop_p=%f needs dx_bf16 and [0,1)", drop_p);
    AV_CHECK(!dx_bf16 || (uintptr_t)dx_bf16 % 8 == 0, "av_layernorm_bwd: dx_bf16 must be 8-byte aligned");
    AV_CHECK(cols > 0 && cols <= 64 * MAXIT, "av_layernorm_bwd: cols=%d out of range", cols);
    AV_CHECK(nblk >= 1, "av_layernorm_bwd: nblk=%d", nblk);
    if (rows == 0) return AV_OK;
    if (!dgb_partial) nblk = (int)((rows + 3) / 4 > 65535 * 16 ? 65535 * 16 : (rows + 3) / 4);   // no partials: one row per wave
    const long long rpb = (rows + nblk - 1) / nblk;
    {
        const bool xa = xdt == AV_F32 ? ((uintptr_t)x % 16 == 0) : ((uintptr_t)x % 8 == 0);
        const bool da = dydt == AV_F32 ? ((uintptr_t)dy % 16 == 0) : ((uintptr_t)dy % 8 == 0);
        if ((cols == 256 || cols == 512 || cols == 1024 || cols == 2048) && xa && da && (uintptr_t)gamma % 16 == 0 && (uintptr_t)dx % 16 == 0 &&
            (!dres || (uintptr_t)dres % 16 == 0)) {
#define LBV(N, P) hipLaunchKernelGGL((ln_bwd_vec_kernel<N, P>), dim3(nblk), dim3(256), 0, (hipStream_t)stream, x, xdt, dy, dydt, gamma, mean, rstd, dres, dx, dgb_partial, rows, rpb, (bf16_t*)dx_bf16, drop_p, drop_seed, drop_stream)
#define LBV_N(P) do { if (cols == 256) LBV(1, P); else if (cols == 512) LBV(2, P); else if (cols == 1024) LBV(4, P); else LBV(8, P); } while (0)
            if (dgb_partial) LBV_N(true); else LBV_N(false);
#undef LBV_N
#undef LBV
            AV_LAUNCH_CHECK();
            return AV_OK;
        }
    }
#define LN_BWD(N, P) hipLaunchKernelGGL((ln_bwd_kernel<N, P>), dim3(nblk), dim3(256), 0, (hipStream_t)stream, x, xdt, dy, dydt, gamma, mean, rstd, dres, dx, dgb_partial, rows, cols, rpb)
#define LN_BWD_N(P) do { if (cols <= 64) LN_BWD(1, P); else if (cols <= 128) LN_BWD(2, P); else if (cols <= 512) LN_BWD(8, P); \
                         else if (cols <= 1024) LN_BWD(16, P); else LN_BWD(32, P); } while (0)
    if (dgb_partial) LN_BWD_N(true); else LN_BWD_N(false);
#undef LN_BWD_N
#undef LN_BWD
    AV_LAUNCH_CHECK();
    if (dx_bf16) {                                                                                  // generic shapes: separate pass
        if (drop_p > 0.f) return av_cast_dropout(dx, AV_F32, dx_bf16, AV_BF16, rows * (long long)cols, drop_p, drop_seed, drop_stream, stream);
        return av_cast(dx, AV_F32, dx_bf16, AV_BF16, rows * (long long)cols, stream);
    }
    return AV_OK;
}

extern "C" int av_log_softmax_fwd(const void* x, int xdt, float* y, long long rows, int cols, void* stream) {
    AV_CHECK(x && y, "av_log_softmax_fwd: null pointer");
    AV_CHECK(cols > 0 && cols <= 64 * MAXIT, "av_log_softmax_fwd: cols=%d out of range", cols);
    if (rows == 0) return AV_OK;
    hipLaunchKernelGGL(lsm_fwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, xdt, y, rows, cols);
    AV_LAUNCH_CHECK();
    return AV_OK;
}

extern "C" int av_log_softmax_bwd_ld(const float* y, const float* dy, void* dx, int dxdt, long long rows, int cols, int ldx, void* stream) {
    AV_CHECK(y && dy && dx, "av_log_softmax_bwd: null pointer");
    AV_CHECK(cols > 0 && cols <= 64 * MAXIT, "av_log_softmax_bwd: cols=%d out of range", cols);
    AV_CHECK(ldx >= cols && ldx <= 64 * MAXIT, "av_log_softmax_bwd: ldx=%d must lie in [cols=%d, %d]", ldx, cols, 64 * MAXIT);
    if (rows == 0) return AV_OK;
    hipLaunchKernelGGL(lsm_bwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, y, dy, dx, dxdt, rows, cols, ldx);
    AV_LAUNCH_CHECK();
    return AV_OK;
}
extern "C" int av_log_softmax_bwd(const float* y, const float* dy, void* dx, int dxdt, long long rows, int cols, void* stream) {
    return av_log_softmax_bwd_ld(y, dy, dx, dxdt, rows, cols, cols, stream);
}

extern "C" int av_colsum(const void* x, int xdt, float* out, long long rows, int cols, long long ld, int accumulate, void* stream) {
    AV_CHECK(x && out, "av_colsum: null pointer");
    AV_CHECK(cols > 0 && ld >= cols, "av_colsum: cols=%d ld=%lld", cols, ld);
    if (!accumulate) {
        if (hipMemsetAsync(out, 0, sizeof(float) * cols, (hipStream_t)stream) != hipSuccess) { av_set_error("av_colsum: memset failed"); return AV_ERR_LAUNCH; }
    }
    if (rows == 0) return AV_OK;
    if (xdt == AV_BF16 && cols % 8 == 0 && ld % 8 == 0 && (uintptr_t)x % 16 == 0) {
        dim3 gv((unsigned)((cols + CSVB_COLS - 1) / CSVB_COLS), (unsigned)((rows + CSV_ROWS - 1) / CSV_ROWS));
        hipLaunchKernelGGL(colsum_bf16_vec_kernel, gv, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, out, rows, cols, ld);
        AV_LAUNCH_CHECK();
        return AV_OK;
    }
    if (xdt == AV_F32 && cols % 4 == 0 && ld % 4 == 0 && (uintptr_t)x % 16 == 0) {
        dim3 gv((unsigned)((cols + 255) / 256), (unsigned)((rows + CSVF_ROWS - 1) / CSVF_ROWS));
        hipLaunchKernelGGL(colsum_f32_vec_kernel, gv, dim3(256), 0, (hipStream_t)stream, (const float*)x, out, rows, cols, ld);
        AV_LAUNCH_CHECK();
        return AV_OK;
    }
    dim3 grid((unsigned)((cols + 63) / 64), (unsigned)((rows + CS_ROWS - 1) / CS_ROWS));
    hipLaunchKernelGGL(colsum_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, xdt, out, rows, cols, ld);
    AV_LAUNCH_CHECK();
    return AV_OK;
}

extern "C" int av_sum_slices(const float* parts, int n_slices, long long n, long long stride, float alpha, float* out, int accumulate,
                             void* stream) {
    AV_CHECK(parts && out && n_slices >= 1 && n >= 0 && stride >= n, "av_sum_slices: bad args (slices=%d n=%lld stride=%lld)", n_slices, n, stride);
    if (n == 0) return AV_OK;
    const int vec = n % 4 == 0 && stride % 4 == 0 && (uintptr_t)parts % 16 == 0 && (uintptr_t)out % 16 == 0;
    hipLaunchKernelGGL(sum_slices_kernel, dim3(ew_grid(vec ? n / 4 : n)), dim3(256), 0, (hipStream_t)stream, parts, n_slices, n, stride, alpha, out,
                       accumulate, vec);
    AV_LAUNCH_CHECK();
    return AV_OK;
}

extern "C" int av_cast(const void* x, int xdt, void* y, int ydt, long long n, void* stream) {
    AV_CHECK(x && y, "av_cast: null pointer");
    if (n == 0) return AV_OK;
    if (xdt == AV_F32 && ydt == AV_BF16 && n % 4 == 0 && (uintptr_t)x % 16 == 0 && (uintptr_t)y % 8 == 0)
        hipLaunchKernelGGL(cast_f32_bf16_vec, dim3(ew_grid(n / 4)), dim3(256), 0, (hipStream_t)stream, (const float4*)x, (bf16x4*)y, n / 4);
    else
        hipLaunchKernelGGL(cast_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, x, xdt, y, ydt, n);
    AV_LAUNCH_CHECK();
    return AV_OK;
}

extern "C" int av_axpby(float a, const void* x, int xdt, float b, float* y, long long n, void* stream) {
    AV_CHECK(x && y, "av_axpby: null pointer");
    if (n == 0) return AV_OK;
    hipLaunchKernelGGL(axpby_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, a, x, xdt, b, y, n);
    AV_LAUNCH_CHECK();
    return AV_OK;
}

extern "C" int av_mask_rows(void* x, int xdt, const unsigned char* keep, long long rows, int cols, void* stream) {
    AV_CHECK(x && keep, "av_mask_rows: null pointer");
    if (rows == 0) return AV_OK;
    hipLaunchKernelGGL(mask_rows_kernel, dim3(ew_grid(rows * cols)), dim3(256), 0, (hipStream_t)stream, x, xdt, keep, rows, (int)cols);
    AV_LAUNCH_CHECK();
    return AV_OK;
}

extern "C" int av_mul_scalar_dev(const float* x, const float* scalar, float* y, long long n, void* stream) {
    AV_CHECK(x && scalar && y, "av_mul_scalar_dev: null pointer");
    if (n == 0) return AV_OK;
    hipLaunchKernelGGL(mul_scalar_dev_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, x, scalar, y, n);
    AV_LAUNCH_CHECK();
    return AV_OK;
}

extern "C" int av_cast_dropout(const void* x, int xdt, void* y, int ydt, long long n, float p, unsigned long long seed, unsigned int stream_id,
                               void* stream) {
    AV_CHECK(x && y, "av_cast_dropout: null pointer");
    AV_CHECK(p >= 0.f && p < 1.f, "av_cast_dropout: p=%f out of [0,1)", p);
    if (n == 0) return AV_OK;
    if (p == 0.f) return av_cast(x, xdt, y, ydt, n, stream);
    hipLaunchKernelGGL(cast_dropout_kernel, dim3(ew_grid((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, xdt, y, ydt, n, p, drop_inv_keep(p), seed,
                       stream_id);
    AV_LAUNCH_CHECK();
    return AV_OK;
}
extern "C" int av_dropout_uniform(float* u, long long n, unsigned long long seed, unsigned int stream_id, void* stream) {
    AV_CHECK(u != nullptr, "av_dropout_uniform: null pointer");
    if (n == 0) return AV_OK;
    hipLaunchKernelGGL(dropout_uniform_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, u, n, seed, stream_id);
    AV_LAUNCH_CHECK();
    return AV_OK;
}
extern "C" int av_zero_feature_cols(void* x, int xdt, const unsigned char* mask, int B, int T, int H, void* stream) {
    AV_CHECK(x && mask && B > 0 && T > 0 && H > 0, "av_zero_feature_cols: bad args");
    const long long n = (long long)B * T * H;
    hipLaunchKernelGGL(zero_feature_cols_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, x, xdt, mask, n, T, H);
    AV_LAUNCH_CHECK();
    return AV_OK;
}
extern "C" int av_overwrite_rows(void* x, int xdt, const unsigned char* mask, const float* embed, long long rows, int cols, void* stream) {
    AV_CHECK(x && mask && embed && cols > 0, "av_overwrite_rows: bad args");
    if (rows == 0) return AV_OK;
    hipLaunchKernelGGL(overwrite_rows_kernel, dim3(ew_grid(rows * cols)), dim3(256), 0, (hipStream_t)stream, x, xdt, mask, embed, rows, cols);
    AV_LAUNCH_CHECK();
    return AV_OK;
}
