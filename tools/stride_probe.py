"""Does the row stride of the operands (K-contiguous rows, 2 KB apart at K = 1024) pace the LDS-DMA stream?  Same products with padded lda / ldb."""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("multimodal-av-model_amd.ops")
M = int(os.environ.get("GEMM_M", "12736"))


def run(N, K, pa, pb, pc=0):
    A = (torch.rand(M, K + pa, device="cuda") * 2 - 1).to(torch.bfloat16); W = (torch.rand(N, K + pb, device="cuda") * 2 - 1).to(torch.bfloat16)
    out = torch.empty(M, N + pc, device="cuda", dtype=torch.bfloat16)
    f = lambda: ops.gemm(A, W, out, M=M, N=N, K=K, lda=K + pa, ldb=K + pb, ldc=N + pc)
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 50
    print(f"N={N:5d} K={K:5d} lda=K+{pa:<4d} ldb=K+{pb:<4d} ldc=N+{pc:<4d} {us:8.1f} us {2.0*M*N*K/us/1e6:7.1f} TF/s", flush=True)


for (N, K) in ((4096, 1024), (3072, 1024), (1024, 4096), (1024, 1024)):
    for (pa, pb, pc) in ((0, 0, 0), (64, 64, 0), (128, 128, 0), (32, 32, 0), (64, 0, 0), (0, 64, 0), (192, 192, 0), (0, 0, 64), (64, 64, 64)):
        run(N, K, pa, pb, pc)
