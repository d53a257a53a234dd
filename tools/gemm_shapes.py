"""Per-shape timing of the step's GEMMs in isolation (wav2vec2-large layer shapes at M = 32 x 199 frames): NT (fwd / dgrad) and TN (dW)."""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("multimodal-av-model_amd.ops")
M = int(os.environ.get("GEMM_M", "6368"))
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / n
tot = 0.0
for (N, K, cnt, what) in [(3072, 1024, 24, "qkv fwd"), (1024, 1024, 24, "out fwd"), (4096, 1024, 24, "ffn1 fwd"), (1024, 4096, 24, "ffn2 fwd"),
                          (1024, 3072, 24, "qkv dgrad"), (1024, 1024, 24, "out dgrad"), (1024, 4096, 24, "ffn1 dgrad"), (4096, 1024, 24, "ffn2 dgrad")]:
    a = (torch.rand(M, K, device="cuda") * 2 - 1).to(torch.bfloat16)
    w = (torch.rand(N, K, device="cuda") * 2 - 1).to(torch.bfloat16)
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    us = timeit(lambda: ops.linear(a, w, None, out=out))
    tot += us * cnt
    print(f"NT {what:10s} M={M} N={N} K={K}: {us:7.1f} us {2.0 * M * N * K / us / 1e6:7.1f} TF/s  x{cnt} = {us * cnt / 1000:6.2f} ms", flush=True)
for (Mo, No, cnt, what) in [(3072, 1024, 24, "qkv dW"), (1024, 1024, 24, "out dW"), (4096, 1024, 24, "ffn1 dW"), (1024, 4096, 24, "ffn2 dW")]:
    dy = (torch.rand(M, Mo, device="cuda") * 2 - 1).to(torch.bfloat16)
    x = (torch.rand(M, No, device="cuda") * 2 - 1).to(torch.bfloat16)
    us = timeit(lambda: ops.matmul_tn(dy, x))
    tot += us * cnt
    print(f"TN {what:10s} out={Mo}x{No} K={M}: {us:7.1f} us {2.0 * M * No * Mo / us / 1e6:7.1f} TF/s  x{cnt} = {us * cnt / 1000:6.2f} ms", flush=True)
print(f"total {tot / 1000:.2f} ms per step")
