"""Fusion cross-attention (K17) at the benchmark size (2 x 64 items, T_v = 100, embed 512 = 4 x 128): the fused block kernel against the
separate launches it replaces (q-projection GEMM, kv-projection GEMM, attention core).  Events on the launch stream."""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("multimodal-av-model_amd.ops")
B, T, E, nh = int(os.environ.get("K17_B", "128")), 100, 512, 4
hd = E // nh
a = torch.randn(B, T, E, device="cuda").to(torch.bfloat16); v = torch.randn(B, T, E, device="cuda").to(torch.bfloat16)
w = (torch.randn(3 * E, E, device="cuda") / E ** 0.5).to(torch.bfloat16); bias = torch.randn(3 * E, device="cuda") * 0.1
bq, bkv = bias[:E].contiguous(), bias[E:].contiguous()


def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / n


def unfused():
    q = ops.linear(a, w[:E], bq).view(B, T, nh, hd)
    kv = ops.linear(v, w[E:], bkv).view(B, T, 2, nh, hd)
    return ops.attention_fwd(q, kv[:, :, 0], kv[:, :, 1], None, hd ** -0.5, need_lse=True)


fl = B * (6.0 * T * E * E + 4.0 * T * T * E)
for name, fn in (("fused block (save q/kv/lse)", lambda: ops.fusion_xattn_fwd(a, v, w, bias, nh, hd ** -0.5, True)),
                 ("fused block (inference)", lambda: ops.fusion_xattn_fwd(a, v, w, bias, nh, hd ** -0.5, False)),
                 ("3 launches: q GEMM + kv GEMM + attention", unfused)):
    us = timeit(fn)
    print(f"B={B} T={T}: {name:42s} {us:8.1f} us  {fl / us / 1e6:7.1f} TF/s = {fl / us / 1e6 / 2500:.3f} of the bf16 MFMA peak", flush=True)

q = torch.randn(B, T, nh, hd, device="cuda").to(torch.bfloat16); kv = torch.randn(B, T, 2, nh, hd, device="cuda").to(torch.bfloat16)
do = torch.randn(B, T, nh, hd, device="cuda").to(torch.bfloat16)
o, lse = ops.attention_fwd(q, kv[:, :, 0], kv[:, :, 1], None, hd ** -0.5, need_lse=True)
dq = torch.empty_like(q); dkv = torch.empty_like(kv)
flb = 10.0 * B * T * T * E
for name, fn in (("whole-sequence core backward", lambda: ops.fusion_xattn_bwd(q, kv, o, do, lse, hd ** -0.5)),
                 ("tiled backward kernels (delta + kv + q)", lambda: ops.attention_bwd(q, kv[:, :, 0], kv[:, :, 1], do, dq, dkv[:, :, 0], dkv[:, :, 1], None, hd ** -0.5, o=o, lse=lse))):
    us = timeit(fn)
    print(f"B={B} T={T}: {name:42s} {us:8.1f} us  {flb / us / 1e6:7.1f} TF/s = {flb / us / 1e6 / 2500:.3f} of the bf16 MFMA peak", flush=True)
