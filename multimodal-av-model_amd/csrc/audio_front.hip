// wav2vec2 feature-encoder layer 0, fully fused: Conv1d(1 -> C, k, stride) + bias -> LayerNorm over C -> GELU
// (hf:291-299 with in_conv_dim = 1).  C_in = 1 makes this layer bandwidth-bound: the 64 000-sample
// waveform expands to a [12 799 x 512] activation, so conv, LN and GELU are done in one pass and the result is
// written once, channel-last ([B, L_out, C]) in the compute dtype, which is the layout the strided-GEMM conv
// layers 1-6 consume.  One wavefront per output frame (C/64 channels per lane), weights in LDS.
#include "av_common.h"

namespace {

constexpr int MAXC_PER_LANE = 16;  // C <= 1024
constexpr int MAXK = 16;

template <typename TO>
__global__ __launch_bounds__(256) void conv0_ln_gelu_kernel(const float* __restrict__ wav, const float* __restrict__ w,
                                                            const float* __restrict__ bias, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, TO* __restrict__ out, int T_in, int L_out,
                                                            int C, int k, int stride, float eps, int frames_per_block) {
    extern __shared__ float sw[];   // [k][C] (tap-major so that lanes read consecutive channels)
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int b = blockIdx.y;
    for (int i = tid; i < C * k; i += 256) {
        const int c = i / k, j = i - c * k;
        sw[j * C + c] = w[i];
    }
    __syncthreads();
    const float* x = wav + (long long)b * T_in;
    const int f0 = blockIdx.x * frames_per_block;
    // the lane's channels c = lane + 64*it are the same for every frame: bias / gamma / beta live in registers (a predicated load
    // inside the frame loop compiles to a branch + vmcnt(0) per access)
    float bl[MAXC_PER_LANE], gl[MAXC_PER_LANE], btl[MAXC_PER_LANE];
#pragma unroll
    for (int it = 0; it < MAXC_PER_LANE; ++it) {
        const int c = lane + 64 * it;
        const int cc = c < C ? c : C - 1;
        const float b0 = bias ? bias[cc] : 0.f, g0 = gamma[cc], e0 = beta[cc];
        bl[it] = c < C ? b0 : 0.f; gl[it] = c < C ? g0 : 0.f; btl[it] = c < C ? e0 : 0.f;
    }
    for (int f = f0 + wv; f < f0 + frames_per_block && f < L_out; f += 4) {
        float xv[MAXK];
#pragma unroll
        for (int j = 0; j < MAXK; ++j) {
            const int jj = j < k ? j : k - 1;                      // unconditional load from a valid address, then select
            const float t = x[f * stride + jj];
            xv[j] = j < k ? t : 0.f;
        }
        float v[MAXC_PER_LANE];
        float s = 0.f;
#pragma unroll
        for (int it = 0; it < MAXC_PER_LANE; ++it) {
            const int c = lane + 64 * it;
            float a = 0.f;
            if (c < C) {
                a = bl[it];
#pragma unroll
                for (int j = 0; j < MAXK; ++j)
                    if (j < k) a += sw[j * C + c] * xv[j];
            }
            v[it] = a;
            s += a;
        }
        const float mean = wave_sum(s) / C;
        float q = 0.f;
#pragma unroll
        for (int it = 0; it < MAXC_PER_LANE; ++it) {
            const int c = lane + 64 * it;
            const float d = (c < C) ? v[it] - mean : 0.f;
            q += d * d;
        }
        const float rstd = rsqrtf(wave_sum(q) / C + eps);
        TO* o = out + ((long long)b * L_out + f) * C;
#pragma unroll
        for (int it = 0; it < MAXC_PER_LANE; ++it) {
            const int c = lane + 64 * it;
            if (c < C) {
                const float t = (v[it] - mean) * rstd * gl[it] + btl[it];
                o[c] = from_f32<TO>(sizeof(TO) == 2 ? gelu_fast(t) : gelu_f(t));       // bf16 output: 1.5e-7-accurate cheap erf
            }
        }
    }
}


// ---- C = 512 fast form: a lane owns 8 CONSECUTIVE channels (16-B weight reads from LDS, one 16-B store per frame), wave
// reductions on DPP (no LDS traffic).  Same arithmetic as the generic kernel above.
__device__ __forceinline__ float dpp_wave_sum(float v) {
#define AV_DPP_ADD(ctrl, rmask) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, rmask, 0xf, true))
    AV_DPP_ADD(0x111, 0xf); AV_DPP_ADD(0x112, 0xf); AV_DPP_ADD(0x114, 0xf); AV_DPP_ADD(0x118, 0xf);   // row_shr 1, 2, 4, 8: row sums in lanes 15, 31, 47, 63
    AV_DPP_ADD(0x142, 0xa);                                                                          // row_bcast15 into rows 1 and 3
    AV_DPP_ADD(0x143, 0xc);                                                                          // row_bcast31 into rows 2 and 3
#undef AV_DPP_ADD
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

template <typename TO, int KK>
__global__ __launch_bounds__(256) void conv0_ln_gelu_c512_kernel(const float* __restrict__ wav, const float* __restrict__ w,
                                                                 const float* __restrict__ bias, const float* __restrict__ gamma,
                                                                 const float* __restrict__ beta, TO* __restrict__ out, int T_in, int L_out,
                                                                 int stride, float eps, int frames_per_block) {
    constexpr int C = 512, k = KK;
    extern __shared__ float sw[];   // [k][64 lanes][8]: tap-major, the lane's 8 channels contiguous
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int b = blockIdx.y;
    for (int i = tid; i < C * k; i += 256) {
        const int c = i / k, j = i - c * k;
        sw[(j * 64 + (c >> 3)) * 8 + (c & 7)] = w[i];
    }
    float bl[8], gl[8], btl[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { bl[e] = bias ? bias[lane * 8 + e] : 0.f; gl[e] = gamma[lane * 8 + e]; btl[e] = beta[lane * 8 + e]; }
    __syncthreads();
    const float* x = wav + (long long)b * T_in;
    const int f0 = blockIdx.x * frames_per_block;
    const int fend = f0 + frames_per_block < L_out ? f0 + frames_per_block : L_out;
    float xn[KK];                                                        // taps of the NEXT frame of this wave: loaded one frame ahead
    {
        const int fq = f0 + wv < fend ? f0 + wv : fend - 1;
#pragma unroll
        for (int j = 0; j < KK; ++j) xn[j] = x[fq * stride + j];
    }
    for (int f = f0 + wv; f < fend; f += 4) {
        float a[8], xv[KK];
#pragma unroll
        for (int j = 0; j < KK; ++j) xv[j] = xn[j];
        {
            const int fq = f + 4 < fend ? f + 4 : fend - 1;
#pragma unroll
            for (int j = 0; j < KK; ++j) xn[j] = x[fq * stride + j];
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) a[e] = bl[e];
#pragma unroll
        for (int j = 0; j < KK; ++j) {
            const float xj = xv[j];
            const f32x4 w0 = *(const f32x4*)(sw + (j * 64 + lane) * 8), w1 = *(const f32x4*)(sw + (j * 64 + lane) * 8 + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                // the kernel is bound by vector issue (45 instructions per output element at 1.8 TB/s of stores).  bf16 output: fused
                // multiply-adds (the library is built with -ffp-contract=off; packed fp32 issues at half rate and bought nothing);
                // fp32 output (parity mode): multiply then add, as the reference's convolution rounds
                if constexpr (sizeof(TO) == 2) { a[e] = __builtin_fmaf(w0[e], xj, a[e]); a[4 + e] = __builtin_fmaf(w1[e], xj, a[4 + e]); }
                else { a[e] += w0[e] * xj; a[4 + e] += w1[e] * xj; }
            }
        }
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) s += a[e];
        const float mean = dpp_wave_sum(s) / C;
        float q = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float d = a[e] - mean; if constexpr (sizeof(TO) == 2) q = __builtin_fmaf(d, d, q); else q += d * d; }
        const float rstd = rsqrtf(dpp_wave_sum(q) / C + eps);
        TO* o = out + ((long long)b * L_out + f) * C + lane * 8;
        if (sizeof(TO) == 2) {
            bf16x8 ov;
#pragma unroll
            for (int e = 0; e < 8; ++e) ov[e] = (bf16_t)gelu_fast(__builtin_fmaf(a[e] - mean, rstd * gl[e], btl[e]));
            *(bf16x8*)o = ov;
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = from_f32<TO>(gelu_f((a[e] - mean) * rstd * gl[e] + btl[e]));
        }
    }
}

}  // namespace

extern "C" int av_conv0_ln_gelu(const float* wav, const float* w, const float* bias, const float* gamma, const float* beta, void* out,
                                int out_dtype, int B, int T_in, int L_out, int C, int k, int stride, float eps, void* stream) {
    AV_CHECK(wav && w && gamma && beta && out, "av_conv0_ln_gelu: null pointer");
    AV_CHECK(C > 0 && C <= 64 * MAXC_PER_LANE && k > 0 && k <= MAXK && stride > 0, "av_conv0_ln_gelu: C=%d k=%d stride=%d out of range", C, k, stride);
    AV_CHECK(L_out >= 0 && (long long)(L_out - 1) * stride + k <= T_in, "av_conv0_ln_gelu: L_out=%d does not fit T_in=%d", L_out, T_in);
    if (B == 0 || L_out == 0) return AV_OK;
    const int fpb = 64;
    dim3 grid((unsigned)((L_out + fpb - 1) / fpb), (unsigned)B);
    const size_t lds = sizeof(float) * (size_t)C * k;
    if (C == 512 && k == 10 && out_dtype == AV_BF16 && (uintptr_t)out % 16 == 0) {
        hipLaunchKernelGGL((conv0_ln_gelu_c512_kernel<bf16_t, 10>), grid, dim3(256), lds, (hipStream_t)stream, wav, w, bias, gamma, beta, (bf16_t*)out, T_in, L_out,
                           stride, eps, fpb);
        AV_LAUNCH_CHECK();
        return AV_OK;
    }
    if (out_dtype == AV_F32)
        hipLaunchKernelGGL(conv0_ln_gelu_kernel<float>, grid, dim3(256), lds, (hipStream_t)stream, wav, w, bias, gamma, beta, (float*)out, T_in, L_out, C, k, stride, eps, fpb);
    else
        hipLaunchKernelGGL(conv0_ln_gelu_kernel<bf16_t>, grid, dim3(256), lds, (hipStream_t)stream, wav, w, bias, gamma, beta, (bf16_t*)out, T_in, L_out, C, k, stride, eps, fpb);
    AV_LAUNCH_CHECK();
    return AV_OK;
}
