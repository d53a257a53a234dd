"""av_colsum on the bias-gradient shapes of the step (bf16 [64 x 199 tokens, N])."""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("multimodal-av-model_amd.ops")
M = 12736
for N in (1024, 3072, 4096):
    x = torch.randn(M, N, device="cuda").to(torch.bfloat16)
    ref = x.float().sum(0)
    out = ops.colsum(x)
    assert (out - ref).abs().max() < 2e-2 * ref.abs().max(), (out - ref).abs().max()
    for _ in range(3): ops.colsum(x)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): ops.colsum(x)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 20
    print(f"colsum [{M} x {N}] bf16: {us:7.1f} us  {M * N * 2 / us / 1e6:6.2f} TB/s", flush=True)
