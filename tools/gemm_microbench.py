"""Per-shape timing of av_gemm (bf16 fast path) on one MI355X: TFLOP/s with events, random data."""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("multimodal-av-model_amd.ops")
L = importlib.import_module("multimodal-av-model_amd._lib")

shapes = [(6368, 3072, 1024), (6368, 1024, 1024), (6368, 4096, 1024), (6368, 1024, 4096), (4096, 4096, 4096), (8192, 8192, 8192),
          (12736, 4096, 1024), (3200, 4096, 1024), (1024, 4096, 6400)]
for (M, N, K) in shapes:
    a = (torch.rand(M, K, device="cuda") * 2 - 1).to(torch.bfloat16)
    w = (torch.rand(N, K, device="cuda") * 2 - 1).to(torch.bfloat16)
    for name, kw in (("plain->bf16", dict()), ("bias+gelu+C2", dict(bias=torch.randn(N, device="cuda"), act=L.ACT_GELU, c2=True)),
                     ("resid->f32", dict(res=True))):
        out_dtype = torch.float32 if kw.get("res") else torch.bfloat16
        out = torch.empty(M, N, device="cuda", dtype=out_dtype)
        c2 = torch.empty(M, N, device="cuda", dtype=out_dtype) if kw.get("c2") else None
        r = torch.randn(M, N, device="cuda") if kw.get("res") else None
        for _ in range(3):
            ops.linear(a, w, kw.get("bias"), out=out, act=kw.get("act", 0), C2=c2, R=r)
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        n = 20
        e0.record()
        for _ in range(n):
            ops.linear(a, w, kw.get("bias"), out=out, act=kw.get("act", 0), C2=c2, R=r)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1000 / n
        print(f"M={M:6d} N={N:5d} K={K:5d} {name:14s} {us:9.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TF/s", flush=True)
