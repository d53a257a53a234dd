"""Per-tile anatomy of the persistent GEMM (v7).  DIAGNOSTIC build (-DAV_GEMM_STAMPS), never the product library: wall-clock stamps
(s_memrealtime, 100 MHz) of every workgroup at tile start / main loop begin / main loop end / epilogue stores issued.
usage (GPU box): python tools/v7_stamps.py"""
import ctypes, importlib, os, subprocess, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = os.path.join(ROOT, "multimodal-av-model_amd")
out_dir = os.path.join(ROOT, "tools", "_bin"); os.makedirs(out_dir, exist_ok=True)
lib_path = os.path.join(out_dir, "libavhip_stamps.so")
srcs = sorted(f for f in os.listdir(os.path.join(PKG, "csrc")) if f.endswith(".hip"))
objs = []
for s in srcs:
    o = os.path.join(PKG, "build", s[:-4] + ".o")
    if s == "gemm_fast.hip":
        o = os.path.join(out_dir, "gemm_fast_stamps.o")
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-ffp-contract=off", "-Wno-unused-value",
                               "-DAV_GEMM_STAMPS"] + os.environ.get("EXTRA_FLAGS", "").split() + ["-c", os.path.join(PKG, "csrc", s), "-o", o])
    objs.append(o)
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib_path] + objs)
os.environ["AVAMD_LIB"] = lib_path
L = importlib.import_module("multimodal-av-model_amd._lib")
ops = importlib.import_module("multimodal-av-model_amd.ops")
lib = L.lib()
lib.av_gemm_stamps7_read.argtypes = [ctypes.c_void_p, ctypes.c_int]; lib.av_gemm_stamps7_read.restype = ctypes.c_int
M = int(os.environ.get("GEMM_M", "12736"))


def run(name, N, K, **kw):
    a = (torch.rand(M, K, device="cuda") * 2 - 1).to(torch.bfloat16); w = (torch.rand(N, K, device="cuda") * 2 - 1).to(torch.bfloat16)
    out = torch.empty(M, N, device="cuda", dtype=kw.pop("odt", torch.bfloat16))
    if kw.pop("bias", False): kw["bias"] = torch.randn(N, device="cuda")
    if kw.pop("c2", False): kw["C2"] = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    if kw.pop("res", False): kw["R"] = torch.randn(M, N, device="cuda")
    for _ in range(5): ops.linear(a, w, out=out, **kw)
    torch.cuda.synchronize()
    buf = np.zeros((256, 16, 4), dtype=np.uint64)
    assert lib.av_gemm_stamps7_read(buf.ctypes.data, 1) == 0
    ops.linear(a, w, out=out, **kw)
    torch.cuda.synchronize()
    assert lib.av_gemm_stamps7_read(buf.ctypes.data, 1) == 0
    st = buf.astype(np.int64)
    t0 = st[:, 0, 0][st[:, 0, 0] > 0].min()
    us = (st - t0) / 100.0
    ntl = int((st[:, :, 3] > 0).sum(axis=1).max())
    print(f"{name:30s} N={N} K={K}: workgroups {int((st[:, 0, 0] > 0).sum())}, tiles per workgroup <= {ntl}, last stamp {us[st > 0].max():7.1f} us")
    for s in range(ntl):
        ok = st[:, s, 3] > 0
        u = us[ok, s]
        print(f"    tile {s}: {int(ok.sum()):4d} wgs  start {np.median(u[:, 0]):7.2f}  wait+barrier {np.median(u[:, 1] - u[:, 0]):5.2f}  main loop {np.median(u[:, 2] - u[:, 1]):6.2f}"
              f"  epilogue {np.median(u[:, 3] - u[:, 2]):6.2f}  (ends {np.median(u[:, 3]):7.2f}, max {u[:, 3].max():7.2f})")


run("plain", 4096, 1024)
run("bias (QKV)", 3072, 1024, bias=True)
run("bias+gelu_gf+C2+drop (FFN up)", 4096, 1024, bias=True, act=L.ACT_GELU_GF, c2=True, drop=(0.1, 1234, 5))
run("plain", 1024, 4096)
run("bias+res f32 (out-proj)", 1024, 1024, bias=True, res=True, odt=torch.float32)
