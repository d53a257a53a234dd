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
    const int cpl = (C + 63) / 64;
    for (int f = f0 + wv; f < f0 + frames_per_block && f < L_out; f += 4) {
        float xv[MAXK];
#pragma unroll
        for (int j = 0; j < MAXK; ++j) xv[j] = j < k ? x[f * stride + j] : 0.f;
        float v[MAXC_PER_LANE];
        float s = 0.f;
#pragma unroll
        for (int it = 0; it < MAXC_PER_LANE; ++it) {
            const int c = lane + 64 * it;
            float a = 0.f;
            if (it < cpl && c < C) {
                a = bias ? bias[c] : 0.f;
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
                const float t = (v[it] - mean) * rstd * gamma[c] + beta[c];
                o[c] = from_f32<TO>(sizeof(TO) == 2 ? gelu_fast(t) : gelu_f(t));       // bf16 output: 1.5e-7-accurate cheap erf
            }
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
    if (out_dtype == AV_F32)
        hipLaunchKernelGGL(conv0_ln_gelu_kernel<float>, grid, dim3(256), lds, (hipStream_t)stream, wav, w, bias, gamma, beta, (float*)out, T_in, L_out, C, k, stride, eps, fpb);
    else
        hipLaunchKernelGGL(conv0_ln_gelu_kernel<bf16_t>, grid, dim3(256), lds, (hipStream_t)stream, wav, w, bias, gamma, beta, (bf16_t*)out, T_in, L_out, C, k, stride, eps, fpb);
    AV_LAUNCH_CHECK();
    return AV_OK;
}
