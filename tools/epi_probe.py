"""Isolated timing of the step's epilogue variants of the dominant GEMM shapes (M from GEMM_M, default 64 x 199 tokens).
Run twice in one gpurun call to compare tilings on the same box: AVAMD_GEMM_V4=0 (128x128 / 256x128) vs default (256x256 8-phase)."""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("multimodal-av-model_amd.ops"); L = importlib.import_module("multimodal-av-model_amd._lib")
M = int(os.environ.get("GEMM_M", "12736"))
drop = (0.1, 1234, 5)


def run(name, N, K, **kw):
    a = (torch.rand(M, K, device="cuda") * 2 - 1).to(torch.bfloat16); w = (torch.rand(N, K, device="cuda") * 2 - 1).to(torch.bfloat16)
    out = torch.empty(M, N, device="cuda", dtype=kw.pop("odt", torch.bfloat16))
    if kw.pop("bias", False): kw["bias"] = torch.randn(N, device="cuda")
    if kw.pop("c2", False): kw["C2"] = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    if kw.pop("aux", False): kw["aux"] = torch.randn(M, N, device="cuda").to(torch.bfloat16)
    if kw.pop("res", False): kw["R"] = torch.randn(M, N, device="cuda")
    for _ in range(3): ops.linear(a, w, out=out, **kw)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ops.linear(a, w, out=out, **kw)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 50
    print(f"{name:34s} N={N:5d} K={K:5d} {us:8.1f} us {2.0*M*N*K/us/1e6:7.1f} TF/s", flush=True)


run("plain", 4096, 1024)
run("bias (QKV)", 3072, 1024, bias=True)
run("bias+gelu+drop (frozen FFN up)", 4096, 1024, bias=True, act=L.ACT_GELU, drop=drop)
run("bias+gelu_gf+C2+drop (FFN up)", 4096, 1024, bias=True, act=L.ACT_GELU_GF, c2=True, drop=drop)
run("bias+gelu_gf+C2 (no dropout)", 4096, 1024, bias=True, act=L.ACT_GELU_GF, c2=True)
run("mul_aux (FFN down dX)", 4096, 1024, act=L.ACT_MUL_AUX, aux=True)
run("plain", 1024, 4096)
run("bias+res+drop f32 (FFN down)", 1024, 4096, bias=True, res=True, drop=drop, odt=torch.float32)
run("bias+res f32 (no dropout)", 1024, 4096, bias=True, res=True, odt=torch.float32)
run("plain", 1024, 1024)
run("bias+res+drop f32 (out-proj)", 1024, 1024, bias=True, res=True, drop=drop, odt=torch.float32)
run("bias+res f32 (no dropout)", 1024, 1024, bias=True, res=True, odt=torch.float32)
run("plain (QKV dX)", 1024, 3072)
