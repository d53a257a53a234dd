"""Pace of the 256 x 256 main loop against K depth and operand row stride, configurations interleaved over several rounds (same process)."""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("multimodal-av-model_amd.ops")
cfgs = [(12288, 4096, 1024, 0), (12288, 4096, 1024, 64), (12288, 4096, 2048, 0), (12288, 4096, 4096, 0), (12288, 4096, 4096, 64), (8192, 8192, 8192, 0), (12288, 4096, 512, 0),
        (12288, 4096, 1024, 8), (12288, 4096, 1024, 192)]
T = {}
for (M, N, K, pad) in cfgs:
    A = (torch.rand(M, K + pad, device="cuda") * 2 - 1).to(torch.bfloat16); W = (torch.rand(N, K + pad, device="cuda") * 2 - 1).to(torch.bfloat16)
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    T[(M, N, K, pad)] = (A, W, out)
res = {c: [] for c in cfgs}
for rnd in range(4):
    for c in cfgs:
        M, N, K, pad = c
        A, W, out = T[c]
        f = lambda: ops.gemm(A, W, out, M=M, N=N, K=K, lda=K + pad, ldb=K + pad, ldc=N)
        for _ in range(2): f()
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): f()
        e1.record(); torch.cuda.synchronize()
        res[c].append(e0.elapsed_time(e1) * 100)
for c in cfgs:
    M, N, K, pad = c
    tiles = ((M + 255) // 256) * ((N + 255) // 256); rounds = (tiles + 255) // 256; nk = K // 64
    us = sorted(res[c])[len(res[c]) // 2]
    print(f"M={M} N={N} K={K} ld=K+{pad:<3d}: " + " ".join(f"{x:7.1f}" for x in res[c]) + f"  median {us:7.1f} us {2.0*M*N*K/us/1e6:7.1f} TF/s  tiles {tiles} rounds {rounds}"
          f"  per round {us/rounds:6.1f}  minus ~5 us epilogue -> {(us/rounds-5)/nk:5.2f} us per K-tile", flush=True)
