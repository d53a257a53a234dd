"""What paces a K-tile of the 256 x 256 8-phase GEMM?  Three DIAGNOSTIC builds of gemm_fast.hip (never the product library), each with the
wall-clock stamps of tools/gemm_stamps.py: the full kernel, one without MFMAs (-DAV_ABL_NOMFMA: every LDS-DMA, fragment read, wait and barrier
stays), one without LDS-DMA (-DAV_ABL_NODMA: MFMAs and fragment reads on whatever the LDS holds).  Prints the main-loop time per K-tile of each
and the implied L2 -> LDS rate per CU.  usage (GPU box): python tools/gemm_ablate.py"""
import ctypes, importlib, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "multimodal-av-model_amd")
out_dir = os.path.join(ROOT, "tools", "_bin"); os.makedirs(out_dir, exist_ok=True)
variants = {"full": [], "no_mfma": ["-DAV_ABL_NOMFMA"], "no_dma": ["-DAV_ABL_NODMA"]}
if len(sys.argv) > 1:                       # child: one variant per process (one libavhip per process)
    name = sys.argv[1]
    import torch
    sys.path.insert(0, ROOT)
    os.environ["AVAMD_LIB"] = os.path.join(out_dir, f"libavhip_abl_{name}.so")
    os.environ["AVAMD_GEMM_V4"] = "2"
    L = importlib.import_module("multimodal-av-model_amd._lib"); ops = importlib.import_module("multimodal-av-model_amd.ops")
    lib = L.lib()
    lib.av_gemm_stamps_read.argtypes = [ctypes.c_void_p, ctypes.c_int]; lib.av_gemm_stamps_read.restype = ctypes.c_int
    for (M, N, K) in ((12736, 4096, 1024), (12736, 1024, 4096), (8192, 8192, 8192)):
        a = (torch.rand(M, K, device="cuda") * 2 - 1).to(torch.bfloat16); w = (torch.rand(N, K, device="cuda") * 2 - 1).to(torch.bfloat16)
        out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        for _ in range(4):
            ops.linear(a, w, out=out)
        torch.cuda.synchronize()
        buf = np.zeros((4096, 4), dtype=np.uint64)
        assert lib.av_gemm_stamps_read(buf.ctypes.data, 4096) == 0
        st = buf[buf[:, 3] > 0].astype(np.int64)
        st = st[st[:, 0] > st[:, 3].max() - 100 * 100000]
        loop = (st[:, 2] - st[:, 1]) / 100.0
        loop = loop[loop > 0.6 * np.median(loop)]
        nk = K // 64
        per = float(np.median(loop)) / nk
        print(f"{name:8s} M={M} N={N} K={K}: main loop {np.median(loop):8.2f} us = {per:5.3f} us per K-tile of 64 KiB"
              + (f" -> {65536 / per / 1e3:6.1f} GB/s per CU of LDS-DMA" if name != "no_dma" else "")
              + (f" ; MFMA alone would need {2048 / 2.1e3:5.3f} us at 2.1 GHz" if name == "full" else ""), flush=True)
    sys.exit(0)
objs_common = [os.path.join(PKG, "build", s[:-4] + ".o") for s in sorted(f for f in os.listdir(os.path.join(PKG, "csrc")) if f.endswith(".hip")) if s != "gemm_fast.hip"]
for name, flags in variants.items():
    o = os.path.join(out_dir, f"gemm_fast_abl_{name}.o")
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-ffp-contract=off", "-Wno-unused-value", "-DAV_GEMM_STAMPS",
                           *flags, "-c", os.path.join(PKG, "csrc", "gemm_fast.hip"), "-o", o])
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(out_dir, f"libavhip_abl_{name}.so"), o] + objs_common)
for name in variants:
    subprocess.check_call([sys.executable, os.path.abspath(__file__), name])
