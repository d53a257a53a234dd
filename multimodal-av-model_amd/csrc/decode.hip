// Greedy CTC decoding on the device (beam_search.py:2-48 of the reference: its "beam search" scores raw frame paths additively
// without prefix merging, so the best beam is the per-frame argmax path - SURVEY 0.3): per-frame argmax over the vocabulary,
// then repeats collapsed and blanks removed.  One workgroup per utterance; only the collapsed ids travel to the host.
#include "av_common.h"

namespace {

constexpr int MAXT = 4096;

__global__ __launch_bounds__(256) void ctc_greedy_kernel(const float* __restrict__ lp, const long long* __restrict__ lengths, int* __restrict__ out_ids,
                                                         int* __restrict__ out_len, int T, int V, int blank) {
    __shared__ int ids[MAXT];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    int Tb = lengths ? (int)lengths[b] : T;
    if (Tb > T) Tb = T;
    if (Tb < 0) Tb = 0;
    const float* base = lp + (long long)b * T * V;
    for (int t = w; t < Tb; t += 4) {
        const float* row = base + (long long)t * V;
        float best = -INFINITY;
        int bi = 0x7fffffff;
        for (int v = lane; v < V; v += 64) {
            const float x = row[v];
            if (x > best || (x == best && v < bi)) { best = x; bi = v; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ob = __shfl_xor(best, o, 64);
            const int oi = __shfl_xor(bi, o, 64);
            if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }      // first maximal index, as torch.argmax
        }
        if (lane == 0) ids[t] = bi == 0x7fffffff ? 0 : bi;
    }
    __syncthreads();
    if (tid == 0) {
        int n = 0, prev = -1;
        int* o = out_ids + (long long)b * T;
        for (int t = 0; t < Tb; ++t) {
            const int i = ids[t];
            if (i != prev && i != blank) o[n++] = i;
            prev = i;
        }
        out_len[b] = n;
        for (int t = n; t < T; ++t) o[t] = -1;
    }
}

}  // namespace

extern "C" int av_ctc_greedy(const float* log_probs, const long long* lengths, int* out_ids, int* out_len, int B, int T, int V, int blank,
                             void* stream) {
    AV_CHECK(log_probs && out_ids && out_len, "av_ctc_greedy: null pointer");
    AV_CHECK(B >= 0 && T > 0 && T <= MAXT && V > 0, "av_ctc_greedy: bad shape B=%d T=%d V=%d (T <= %d)", B, T, V, MAXT);
    if (B == 0) return AV_OK;
    hipLaunchKernelGGL(ctc_greedy_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, log_probs, lengths, out_ids, out_len, T, V, blank);
    AV_LAUNCH_CHECK();
    return AV_OK;
}
