"""Reference ceiling for the step's GEMM shapes: torch.matmul (hipBLASLt / rocBLAS) on the same operands, for comparison with
tools/gemm_shapes.py.  Measurement only: the product never calls a BLAS library."""
import os, sys, torch
M = int(os.environ.get("GEMM_M", "6368"))
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / n
for (N, K) in [(3072, 1024), (1024, 1024), (4096, 1024), (1024, 4096), (1024, 3072)]:
    a = (torch.rand(M, K, device="cuda") * 2 - 1).to(torch.bfloat16)
    w = (torch.rand(N, K, device="cuda") * 2 - 1).to(torch.bfloat16)
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    us = timeit(lambda: torch.matmul(a, w.t(), out=out))
    print(f"NT torch M={M} N={N} K={K}: {us:7.1f} us {2.0 * M * N * K / us / 1e6:7.1f} TF/s", flush=True)
for (Mo, No) in [(3072, 1024), (1024, 1024), (4096, 1024), (1024, 4096)]:
    dy = (torch.rand(M, Mo, device="cuda") * 2 - 1).to(torch.bfloat16)
    x = (torch.rand(M, No, device="cuda") * 2 - 1).to(torch.bfloat16)
    us = timeit(lambda: torch.matmul(dy.t(), x))
    print(f"TN torch out={Mo}x{No} K={M}: {us:7.1f} us {2.0 * M * No * Mo / us / 1e6:7.1f} TF/s", flush=True)
for n in (4096, 8192):
    a = (torch.rand(n, n, device="cuda") * 2 - 1).to(torch.bfloat16); w = (torch.rand(n, n, device="cuda") * 2 - 1).to(torch.bfloat16)
    us = timeit(lambda: torch.matmul(a, w.t()))
    print(f"NT torch {n}^3: {us:7.1f} us {2.0 * n ** 3 / us / 1e6:7.1f} TF/s", flush=True)
