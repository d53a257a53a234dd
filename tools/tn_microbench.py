"""dW = dY^T X and dX = dY W timing (bf16): k-major fast kernel (default) vs transpose + NT kernel (AVAMD_GEMM_KMAJOR=0)."""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("multimodal-av-model_amd.ops")


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / n


T = int(os.environ.get("TN_T", "12736"))
for (Nout, Kin) in [(1024, 1024), (3072, 1024), (4096, 1024), (1024, 4096)]:
    dy = (torch.rand(T, Nout, device="cuda") - 0.5).to(torch.bfloat16)
    x = (torch.rand(T, Kin, device="cuda") - 0.5).to(torch.bfloat16)
    w = (torch.rand(Nout, Kin, device="cuda") - 0.5).to(torch.bfloat16)
    fl = 2.0 * T * Nout * Kin
    us = timeit(lambda: ops.matmul_tn(dy, x))
    print(f"dW  tokens={T} out={Nout} in={Kin}: {us:8.1f} us {fl / us / 1e6:7.1f} TF/s", flush=True)
    us = timeit(lambda: ops.matmul_nn(dy, w, out_dtype=torch.bfloat16))
    print(f"dX  tokens={T} out={Nout} in={Kin}: {us:8.1f} us {fl / us / 1e6:7.1f} TF/s  (weight transposed per call when KMAJOR=0)", flush=True)
