"""Quick A/B of av_gemm variants on three in-step shapes (plain bf16 output)."""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("multimodal-av-model_amd.ops")
for (M, N, K) in [(6368, 4096, 1024), (6368, 1024, 1024), (6368, 1024, 4096), (8192, 8192, 8192)]:
    a = (torch.rand(M, K, device="cuda") * 2 - 1).to(torch.bfloat16)
    w = (torch.rand(N, K, device="cuda") * 2 - 1).to(torch.bfloat16)
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    for _ in range(3):
        ops.linear(a, w, None, out=out)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.linear(a, w, None, out=out)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1000 / 20
    print(f"M={M} N={N} K={K}: {us:8.1f} us {2.0 * M * N * K / us / 1e6:7.1f} TF/s", flush=True)
