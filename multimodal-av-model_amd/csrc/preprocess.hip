// Device side of the input pipeline (dataset/multi_speaker_dataset.py:13-59 of the reference): what `load_pair` does
// per sample on the host with numpy / cv2 after decoding -
//   * lips:  frames.astype(float32).mean(-1) -> cv2.resize(96, 96) (bilinear) -> / 255          (:49-58)
//   * audio: zero-pad both clips to the longer one, add, divide by (max|x| + 1e-6)              (:21-32)
//   * masks: 1 where both speakers talk, 2 where only this speaker still talks, 0 elsewhere     (:35-45)
// as two bandwidth-bound kernels over data that is already on the device (decoding the wav / npy files stays on the host).
// The arithmetic follows the float32 operation order of the reference (build flag -ffp-contract=off: no fused multiply-adds), so
// the results equal the numpy restatement in oracle/pipeline_oracle.py bit for bit.  cv2 is not installed in this image: the
// resize follows the published INTER_LINEAR law (pixel centres, src = (dst + 0.5) * scale - 0.5 evaluated in double, clamped
// taps, horizontal pass then vertical pass in float32) - parity with cv2 itself is unpinned.
#include "av_common.h"

namespace {

template <typename TS>
__device__ __forceinline__ float gray_at(const TS* __restrict__ f, int y, int x, int Ws, int C) {
    const TS* p = f + ((long long)y * Ws + x) * C;
    float s = (float)p[0];
    for (int c = 1; c < C; ++c) s = s + (float)p[c];              // numpy add.reduce over the last axis, float32 accumulator
    return s / (float)C;
}

template <typename TS>
__global__ __launch_bounds__(256) void lip_gray_resize_kernel(const TS* __restrict__ src, float* __restrict__ dst, int T, int Hs, int Ws, int C,
                                                              int Hd, int Wd, double scale_y, double scale_x, float divisor) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)T * Hd * Wd) return;
    const int x = (int)(i % Wd), y = (int)((i / Wd) % Hd), t = (int)(i / ((long long)Wd * Hd));
    float fx = (float)((x + 0.5) * scale_x - 0.5), fy = (float)((y + 0.5) * scale_y - 0.5);
    int sx = (int)floorf(fx), sy = (int)floorf(fy);
    fx -= (float)sx; fy -= (float)sy;
    if (sx < 0) { sx = 0; fx = 0.f; }
    if (sx >= Ws - 1) { sx = Ws - 1; fx = 0.f; }
    if (sy < 0) { sy = 0; fy = 0.f; }
    if (sy >= Hs - 1) { sy = Hs - 1; fy = 0.f; }
    const int sx1 = sx + 1 < Ws ? sx + 1 : Ws - 1, sy1 = sy + 1 < Hs ? sy + 1 : Hs - 1;
    const TS* f = src + (long long)t * Hs * Ws * C;
    const float a0 = 1.f - fx, a1 = fx, b0 = 1.f - fy, b1 = fy;
    const float h0 = gray_at(f, sy, sx, Ws, C) * a0 + gray_at(f, sy, sx1, Ws, C) * a1;
    const float h1 = gray_at(f, sy1, sx, Ws, C) * a0 + gray_at(f, sy1, sx1, Ws, C) * a1;
    dst[i] = (h0 * b0 + h1 * b1) / divisor;
}

__device__ __forceinline__ float mixed_at(const float* __restrict__ a1, long long len1, const float* __restrict__ a2, long long len2, long long i) {
    return (i < len1 ? a1[i] : 0.f) + (i < len2 ? a2[i] : 0.f);
}

__global__ __launch_bounds__(256) void mix_absmax_kernel(const float* __restrict__ a1, long long len1, const float* __restrict__ a2, long long len2,
                                                         long long n, unsigned* __restrict__ peak_bits) {
    float m = 0.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) m = fmaxf(m, fabsf(mixed_at(a1, len1, a2, len2, i)));
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) atomicMax(peak_bits, __float_as_uint(m));       // non-negative floats order like their bit patterns
}

__global__ __launch_bounds__(256) void mix_finish_kernel(const float* __restrict__ a1, long long len1, const float* __restrict__ a2, long long len2, long long n,
                                                         const unsigned* __restrict__ peak_bits, float* __restrict__ mixed, long long* __restrict__ mask1,
                                                         long long* __restrict__ mask2) {
    const float denom = __uint_as_float(*peak_bits) + 1e-6f;
    const long long lo = len1 < len2 ? len1 : len2;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        mixed[i] = mixed_at(a1, len1, a2, len2, i) / denom;
        mask1[i] = i < lo ? 1 : (i < len1 ? 2 : 0);
        mask2[i] = i < lo ? 1 : (i < len2 ? 2 : 0);
    }
}


// ---------------------------------------------------------------------------------------------------- band-limited resampling
// y[t] = sum over both wings of a Kaiser-windowed sinc evaluated at the fractional position of output sample t (table with linear
// interpolation between entries: `win` / `delta`, `num_table` entries per zero crossing) - the published "windowed-sinc with
// interpolated table" law (J. O. Smith, "Digital Audio Resampling", as implemented by resampy): for down-sampling the filter is
// stretched by `scale` = sr_out / sr_in and its gain scaled by the same factor (done on the table by the caller).
__global__ __launch_bounds__(256) void resample_sinc_kernel(const float* __restrict__ x, long long n_in, float* __restrict__ y, long long n_out,
                                                            const float* __restrict__ win, const float* __restrict__ delta, int nwin, int num_table,
                                                            double time_increment, double scale) {
    const int index_step = (int)(scale * num_table);
    for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < n_out; t += (long long)gridDim.x * 256) {
        const double time_register = (double)t * time_increment;
        const long long n = (long long)time_register;
        double frac = scale * (time_register - (double)n);
        double index_frac = frac * num_table;
        int offset = (int)index_frac;
        double eta = index_frac - offset;
        double acc = 0.0;
        long long i_max = (nwin - offset) / index_step;
        if (i_max > n + 1) i_max = n + 1;
        for (long long i = 0; i < i_max; ++i) {                  // left wing: x[n], x[n - 1], ...
            const int k = offset + (int)i * index_step;
            acc += ((double)win[k] + eta * (double)delta[k]) * (double)x[n - i];
        }
        frac = scale - frac;                                       // right wing: x[n + 1], x[n + 2], ...
        index_frac = frac * num_table;
        offset = (int)index_frac;
        eta = index_frac - offset;
        long long k_max = (nwin - offset) / index_step;
        if (k_max > n_in - n - 1) k_max = n_in - n - 1;
        for (long long k = 0; k < k_max; ++k) {
            const int j = offset + (int)k * index_step;
            acc += ((double)win[j] + eta * (double)delta[j]) * (double)x[n + k + 1];
        }
        y[t] = (float)acc;
    }
}

}  // namespace

extern "C" int av_lip_gray_resize(const void* src, int src_is_u8, float* dst, int T, int Hs, int Ws, int C, int Hd, int Wd, float divisor, void* stream) {
    AV_CHECK(src && dst, "av_lip_gray_resize: null pointer");
    AV_CHECK(T >= 0 && Hs > 0 && Ws > 0 && C > 0 && C <= 4 && Hd > 0 && Wd > 0 && divisor != 0.f,
             "av_lip_gray_resize: bad shape T=%d src=%dx%dx%d dst=%dx%d", T, Hs, Ws, C, Hd, Wd);
    const long long total = (long long)T * Hd * Wd;
    if (total == 0) return AV_OK;
    AV_CHECK(total < (1LL << 31) * 256, "av_lip_gray_resize: too many output pixels");
    const double sy = (double)Hs / Hd, sx = (double)Ws / Wd;
    const dim3 grid((unsigned)((total + 255) / 256));
    if (src_is_u8) hipLaunchKernelGGL(lip_gray_resize_kernel<unsigned char>, grid, dim3(256), 0, (hipStream_t)stream, (const unsigned char*)src, dst, T, Hs, Ws, C, Hd, Wd, sy, sx, divisor);
    else hipLaunchKernelGGL(lip_gray_resize_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)src, dst, T, Hs, Ws, C, Hd, Wd, sy, sx, divisor);
    AV_LAUNCH_CHECK();
    return AV_OK;
}

extern "C" int av_mix_pair(const float* a1, long long len1, const float* a2, long long len2, float* mixed, long long* mask1, long long* mask2,
                           unsigned* peak_ws, void* stream) {
    AV_CHECK(len1 >= 0 && len2 >= 0, "av_mix_pair: negative length");
    const long long n = len1 > len2 ? len1 : len2;
    if (n == 0) return AV_OK;
    AV_CHECK((a1 || len1 == 0) && (a2 || len2 == 0) && mixed && mask1 && mask2 && peak_ws, "av_mix_pair: null pointer");
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(peak_ws, 0, sizeof(unsigned), st) != hipSuccess) { av_set_error("av_mix_pair: memset failed"); return AV_ERR_LAUNCH; }
    long long blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(mix_absmax_kernel, dim3((unsigned)blocks), dim3(256), 0, st, a1, len1, a2, len2, n, peak_ws);
    hipLaunchKernelGGL(mix_finish_kernel, dim3((unsigned)blocks), dim3(256), 0, st, a1, len1, a2, len2, n, peak_ws, mixed, mask1, mask2);
    AV_LAUNCH_CHECK();
    return AV_OK;
}

extern "C" int av_resample_sinc(const float* x, long long n_in, float* y, long long n_out, const float* win, const float* delta, int nwin, int num_table,
                                int sr_in, int sr_out, void* stream) {
    AV_CHECK(n_in >= 0 && n_out >= 0 && nwin > 0 && num_table > 0 && sr_in > 0 && sr_out > 0, "av_resample_sinc: bad args");
    if (n_out == 0) return AV_OK;
    AV_CHECK(x && y && win && delta && n_in > 0, "av_resample_sinc: null pointer");
    const double ratio = (double)sr_out / (double)sr_in;
    const double scale = ratio < 1.0 ? ratio : 1.0;
    AV_CHECK((int)(scale * num_table) >= 1, "av_resample_sinc: rate ratio %d -> %d too small for a table of %d entries per zero crossing", sr_in, sr_out, num_table);
    long long blocks = (n_out + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(resample_sinc_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, n_in, y, n_out, win, delta, nwin, num_table,
                       1.0 / ratio, scale);
    AV_LAUNCH_CHECK();
    return AV_OK;
}
