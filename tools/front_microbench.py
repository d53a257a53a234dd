"""conv0 + LN + GELU (wav2vec2 feature-encoder layer 0) timing at B=32 x 4 s."""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("multimodal-av-model_amd.ops"); L = importlib.import_module("multimodal-av-model_amd._lib")
B, T, C, k, st = 32, 64000, 512, 10, 5
Lo = (T - k) // st + 1
wav = torch.randn(B, T, device="cuda"); w = torch.randn(C, k, device="cuda") * 0.3; b = torch.randn(C, device="cuda")
g = torch.randn(C, device="cuda"); be = torch.randn(C, device="cuda")
out = torch.empty(B, Lo, C, device="cuda", dtype=torch.bfloat16)
def run():
    L.check(L.lib().av_conv0_ln_gelu(ops.ptr(wav), ops.ptr(w), ops.ptr(b), ops.ptr(g), ops.ptr(be), ops.ptr(out), L.AV_BF16, B, T, Lo, C, k, st, 1e-5, ops.stream()), "conv0")
for _ in range(3): run()
torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): run()
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 100
print(f"conv0_ln_gelu: {us:.1f} us, output {out.numel() * 2 / us / 1e6:.2f} TB/s")
