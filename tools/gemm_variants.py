"""A/B of the bf16 NT GEMM tilings on the step's shapes (M = 64 x 199 tokens) and on square problems: one child process per variant
(the variant switches are read once per process), random operands, events on the launch stream.
  python tools/gemm_variants.py            -> table;   child mode: python tools/gemm_variants.py --child"""
import importlib, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHAPES = [(12736, 1024, 1024, "plain"), (12736, 1024, 1024, "res"), (12736, 1024, 4096, "plain"), (12736, 1024, 3072, "plain"),
          (12736, 3072, 1024, "plain"), (12736, 4096, 1024, "plain"), (12736, 4096, 1024, "gelu+c2+drop"), (6368, 1024, 1024, "plain"),
          (6368, 4096, 1024, "plain")]
VARIANTS = [("BM=256", {"AVAMD_GEMM_V4_BM": "256"}), ("BM auto", {}), ("BM=208", {"AVAMD_GEMM_V4_BM": "208"})]


def child():
    import torch
    sys.path.insert(0, ROOT)
    ops = importlib.import_module("multimodal-av-model_amd.ops")
    res = []
    L = importlib.import_module("multimodal-av-model_amd._lib")
    for (M, N, K, epi) in SHAPES:
        a = (torch.rand(M, K, device="cuda") * 2 - 1).to(torch.bfloat16)
        w = (torch.rand(N, K, device="cuda") * 2 - 1).to(torch.bfloat16)
        kw = {}
        odt = torch.bfloat16
        if "res" in epi:
            kw.update(R=torch.randn(M, N, device="cuda"), bias=torch.randn(N, device="cuda")); odt = torch.float32
        if "gelu+c2" in epi:
            kw.update(bias=torch.randn(N, device="cuda"), act=L.ACT_GELU, C2=torch.empty(M, N, device="cuda", dtype=torch.bfloat16))
        if epi == "bias":
            kw.update(bias=torch.randn(N, device="cuda"))
        if epi == "geluonly":
            kw.update(bias=torch.randn(N, device="cuda"), act=L.ACT_GELU)
        if epi == "c2only":
            kw.update(bias=torch.randn(N, device="cuda"), C2=torch.empty(M, N, device="cuda", dtype=torch.bfloat16))
        if "gelugrad" in epi:
            kw.update(act=L.ACT_MUL_GELU_GRAD, aux=torch.randn(M, N, device="cuda").to(torch.bfloat16))
        if "drop" in epi:
            kw.update(drop=(0.1, 1234, 7))
        out = torch.empty(M, N, device="cuda", dtype=odt)
        ref = (a[:64].float() @ w.float().t())
        err = err2 = 0.0
        if epi == "plain":
            ops.linear(a, w, None, out=out)
            err = float((out[:64].float() - ref).abs().max() / ref.abs().max())
            err2 = float((out[-64:].float() - a[-64:].float() @ w.float().t()).abs().max() / ref.abs().max())
        run = lambda: ops.linear(a, w, out=out, **kw)
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        best = 1e9
        for rep in range(3):
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            n = 20
            e0.record()
            for _ in range(n):
                run()
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1000 / n)
        res.append((M, N, K, epi, best, 2.0 * M * N * K / best / 1e6, max(err, err2)))
    for r in res:
        print("RES %d %d %d %s %.2f %.1f %.2e" % r, flush=True)


if __name__ == "__main__":
    if "--child" in sys.argv:
        child()
    else:
        table = {}
        for name, env in VARIANTS:
            e = dict(os.environ); e.update(env)
            out = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=e, capture_output=True, text=True)
            if out.returncode != 0:
                print(name, "FAILED", out.stderr[-2000:])
                continue
            for l in out.stdout.splitlines():
                if l.startswith("RES"):
                    _, M, N, K, epi, us, tf, err = l.split()
                    table.setdefault((int(M), int(N), int(K), epi), {})[name] = (float(us), float(tf), float(err))
        for shp, d in table.items():
            print(f"M={shp[0]:6d} N={shp[1]:5d} K={shp[2]:5d} {shp[3]:13s} " + "  |  ".join(f"{n}: {v[0]:8.1f} us {v[1]:7.1f} TF/s err {v[2]:.1e}" for n, v in d.items()), flush=True)
