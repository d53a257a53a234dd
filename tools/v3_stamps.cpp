// Diagnostic harness for the experimental 256 x 256 GEMM: builds gemm_fast.hip with core-clock stamps around the phases of its
// K loop and prints the average clocks per K-tile (wavefronts 0 and 5 of every workgroup).  Not part of the product.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/v3_stamps.cpp -o tools/_bin/v3_stamps && AVAMD_GEMM_V3=2 tools/_bin/v3_stamps
#define AV_V3_STAMPS 1
#include <cstdarg>
#include <cstdio>
#include <vector>
#include "../multimodal-av-model_amd/csrc/gemm_fast.hip"
void av_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); }
int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 8192, N = argc > 2 ? atoi(argv[2]) : 8192, K = argc > 3 ? atoi(argv[3]) : 8192;
    std::vector<unsigned short> ha((size_t)M * K), hb((size_t)N * K);
    unsigned x = 12345;
    for (auto& v : ha) { x = x * 1664525u + 1013904223u; v = (unsigned short)(0x3c00 + ((x >> 9) & 0x3ff) * 0 + ((x >> 16) & 0x7f)) ^ ((x >> 3) & 0x8000); }
    for (auto& v : hb) { x = x * 1664525u + 1013904223u; v = (unsigned short)(0x3c00 + ((x >> 16) & 0x7f)) ^ ((x >> 3) & 0x8000); }
    void *A, *B, *C; unsigned long long* dbg;
    hipMalloc(&A, ha.size() * 2); hipMalloc(&B, hb.size() * 2); hipMalloc(&C, (size_t)M * N * 2); hipMalloc(&dbg, 80);
    hipMemcpy(A, ha.data(), ha.size() * 2, hipMemcpyHostToDevice); hipMemcpy(B, hb.data(), hb.size() * 2, hipMemcpyHostToDevice);
    av_gemm_args p = {};
    p.A = A; p.B = B; p.C = C; p.M = M; p.N = N; p.K = K; p.batch = 1; p.lda = K; p.ldb = K; p.ldc = N;
    p.a_mode = AV_A_ROWMAJOR; p.b_mode = AV_B_NK; p.in_dtype = AV_BF16; p.out_dtype = AV_BF16; p.aux_dtype = AV_BF16; p.act = AV_ACT_NONE; p.alpha = 1.f;
    p.aux = dbg;
    for (int it = 0; it < 3; ++it) {
        hipMemset(dbg, 0, 80);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0, 0);
        const int rc = av_gemm_fast_try(p, 0);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long h[10]; hipMemcpy(h, dbg, 80, hipMemcpyDeviceToHost);
        printf("rc=%d %.1f us %.0f TF/s\n", rc, ms * 1e3, 2.0 * M * N * K / ms / 1e9);
        for (int wv = 0; wv < 2; ++wv) {
            const double n = (double)h[wv * 5 + 4];
            if (n > 0) printf("  wave %d: per K-tile clocks: wait %.0f  barrier %.0f  dma-issue %.0f  reads+mfma-issue %.0f  (sum %.0f)\n", wv ? 5 : 0,
                              h[wv * 5] / n, h[wv * 5 + 1] / n, h[wv * 5 + 2] / n, h[wv * 5 + 3] / n, (h[wv * 5] + h[wv * 5 + 1] + h[wv * 5 + 2] + h[wv * 5 + 3]) / n);
        }
    }
    return 0;
}
