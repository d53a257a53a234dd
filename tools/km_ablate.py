"""What paces a K-tile of the 8-phase kernel's k-major form (dW = dY^T X)?  DIAGNOSTIC builds of gemm_fast.hip (never the product library): the full
kernel, one without MFMAs, one without LDS-DMA, one without fragment reads; each timed on the FFN dW product (64 tiles x 4 K slices = 256
workgroups of 50 K-tiles) next to the row-major (NT) form of the same kernel on a 256-workgroup product of the same K-tile count.
usage (GPU box): python tools/km_ablate.py"""
import importlib, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "multimodal-av-model_amd")
out_dir = os.path.join(ROOT, "tools", "_bin"); os.makedirs(out_dir, exist_ok=True)
variants = {"full": [], "no_mfma": ["-DAV_ABL_NOMFMA"], "no_dma": ["-DAV_ABL_NODMA"], "no_read": ["-DAV_ABL_NOREAD"],
            "dma_only": ["-DAV_ABL_NOMFMA", "-DAV_ABL_NOREAD"], "dma_only_kscramble": ["-DAV_ABL_NOMFMA", "-DAV_ABL_NOREAD", "-DAV_ABL_KSCRAMBLE"], "full_kscramble": ["-DAV_ABL_KSCRAMBLE"], "dma_only_piece8x128": ["-DAV_ABL_NOMFMA", "-DAV_ABL_NOREAD", "-DAV_ABL_KMPIECE"], "read_only": ["-DAV_ABL_NOMFMA", "-DAV_ABL_NODMA"]}
if os.environ.get("KM_ABL_ONLY"):
    variants = {k: v for k, v in variants.items() if k in os.environ["KM_ABL_ONLY"].split(",")}
if len(sys.argv) > 1:
    name = sys.argv[1]
    import torch
    sys.path.insert(0, ROOT)
    os.environ["AVAMD_LIB"] = os.path.join(out_dir, f"libavhip_kmabl_{name}.so")
    os.environ["AVAMD_GEMM_V4"] = "2"; os.environ["AVAMD_GEMM_V7"] = "0"
    L = importlib.import_module("multimodal-av-model_amd._lib"); ops = importlib.import_module("multimodal-av-model_amd.ops")

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

    T, Mo, No = 12736, 4096, 1024
    dy = (torch.rand(T, Mo, device="cuda") - 0.5).to(torch.bfloat16); x = (torch.rand(T, No, device="cuda") - 0.5).to(torch.bfloat16)
    S, chunk = 4, 3200
    parts = torch.empty((S, Mo, No), dtype=torch.float32, device="cuda")
    us = timeit(lambda: ops.gemm(dy, x, parts, M=Mo, N=No, K=chunk, lda=Mo, ldb=No, ldc=No, a_mode=L.A_TRANS, b_mode=L.B_KN, batch=S,
                                 sA=chunk * Mo, sB=chunk * No, sC=Mo * No, k_total=T))
    print(f"{name:8s} k-major  dW 4096 x 1024, 4 slices of 50 K-tiles (256 workgroups): {us:7.1f} us = {us / 50:5.2f} us per K-tile", flush=True)
    dy2 = (torch.rand(3200, 4096, device="cuda") - 0.5).to(torch.bfloat16); x2 = (torch.rand(3200, 4096, device="cuda") - 0.5).to(torch.bfloat16)
    out2 = torch.empty((4096, 4096), dtype=torch.float32, device="cuda")
    us = timeit(lambda: ops.gemm(dy2, x2, out2, M=4096, N=4096, K=3200, lda=4096, ldb=4096, ldc=4096, a_mode=L.A_TRANS, b_mode=L.B_KN))
    print(f"{name:8s} k-major  4096 x 4096 x 3200, one slice (256 workgroups of 50 K-tiles, 52 MB of operands): {us:7.1f} us = {us / 50:5.2f} us per K-tile", flush=True)
    for pad in (64, 8, 512):
        dyp = (torch.rand(3200, 4096 + pad, device="cuda") - 0.5).to(torch.bfloat16); xp = (torch.rand(3200, 4096 + pad, device="cuda") - 0.5).to(torch.bfloat16)
        us = timeit(lambda: ops.gemm(dyp, xp, out2, M=4096, N=4096, K=3200, lda=4096 + pad, ldb=4096 + pad, ldc=4096, a_mode=L.A_TRANS, b_mode=L.B_KN))
        print(f"{name:8s} k-major  4096 x 4096 x 3200, rows padded by {pad} elements (stride {2 * (4096 + pad)} B): {us:7.1f} us = {us / 50:5.2f} us per K-tile", flush=True)
    dy4 = (torch.rand(16, 3200, 1024, device="cuda") - 0.5).to(torch.bfloat16); x4 = (torch.rand(16, 3200, 1024, device="cuda") - 0.5).to(torch.bfloat16)
    out4 = torch.empty((16, 1024, 1024), dtype=torch.float32, device="cuda")
    us = timeit(lambda: ops.gemm(dy4, x4, out4, M=1024, N=1024, K=3200, lda=1024, ldb=1024, ldc=1024, a_mode=L.A_TRANS, b_mode=L.B_KN, batch=16,
                                 sA=3200 * 1024, sB=3200 * 1024, sC=1024 * 1024))
    print(f"{name:8s} k-major  16 problems 1024 x 1024 x 3200 (2-KB rows: a K-tile spans 128 KB per operand instead of 512 KB): {us:7.1f} us = {us / 50:5.2f} us per K-tile", flush=True)
    dy3 = (torch.rand(256, 3200, 256, device="cuda") - 0.5).to(torch.bfloat16); x3 = (torch.rand(256, 3200, 256, device="cuda") - 0.5).to(torch.bfloat16)
    out3 = torch.empty((256, 256, 256), dtype=torch.float32, device="cuda")
    us = timeit(lambda: ops.gemm(dy3, x3, out3, M=256, N=256, K=3200, lda=256, ldb=256, ldc=256, a_mode=L.A_TRANS, b_mode=L.B_KN, batch=256,
                                 sA=3200 * 256, sB=3200 * 256, sC=256 * 256))
    print(f"{name:8s} k-major  256 dense problems 256 x 256 x 3200 (512-B rows back to back: one workgroup streams 1.6 MB per operand): {us:7.1f} us = {us / 50:5.2f} us per K-tile", flush=True)
    a = (torch.rand(4096, 3200, device="cuda") - 0.5).to(torch.bfloat16); w = (torch.rand(4096, 3200, device="cuda") - 0.5).to(torch.bfloat16)
    out = torch.empty(4096, 4096, device="cuda", dtype=torch.float32)
    us = timeit(lambda: ops.gemm(a, w, out, M=4096, N=4096, K=3200, lda=3200, ldb=3200, ldc=4096))
    print(f"{name:8s} row-major NT 4096 x 4096 x 3200 (256 workgroups of 50 K-tiles), v4:   {us:7.1f} us = {us / 50:5.2f} us per K-tile", flush=True)
    sys.exit(0)
objs_common = [os.path.join(PKG, "build", s[:-4] + ".o") for s in sorted(f for f in os.listdir(os.path.join(PKG, "csrc")) if f.endswith(".hip")) if s != "gemm_fast.hip"]
procs = []
for name, flags in variants.items():
    o = os.path.join(out_dir, f"gemm_fast_kmabl_{name}.o")
    procs.append((name, o, subprocess.Popen(["hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-ffp-contract=off", "-Wno-unused-value",
                                             *flags, "-c", os.path.join(PKG, "csrc", "gemm_fast.hip"), "-o", o])))
for name, o, pr in procs:
    assert pr.wait() == 0
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(out_dir, f"libavhip_kmabl_{name}.so"), o] + objs_common)
for name in variants:
    subprocess.check_call([sys.executable, os.path.abspath(__file__), name])
