"""Where does a launch of the 8-phase GEMM spend its time?  DIAGNOSTIC build of libavhip (gemm_fast.hip with -DAV_GEMM_STAMPS: wall-clock
stamps per workgroup at entry / first MFMA phase / end of main loop / exit after its stores completed), never the product library.
usage (GPU box): python tools/gemm_stamps.py        -> per shape: prologue / main loop / epilogue per workgroup, rounds, idle gaps"""
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
for s in srcs:                      # reuse the product objects for everything but gemm_fast
    o = os.path.join(PKG, "build", s[:-4] + ".o")
    if s == "gemm_fast.hip":
        o = os.path.join(out_dir, "gemm_fast_stamps.o")
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-ffp-contract=off", "-Wno-unused-value",
                               "-DAV_GEMM_STAMPS", "-c", os.path.join(PKG, "csrc", s), "-o", o])
    objs.append(o)
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib_path] + objs)
os.environ["AVAMD_LIB"] = lib_path
L = importlib.import_module("multimodal-av-model_amd._lib")
ops = importlib.import_module("multimodal-av-model_amd.ops")
lib = L.lib()
lib.av_gemm_stamps_read.argtypes = [ctypes.c_void_p, ctypes.c_int]; lib.av_gemm_stamps_read.restype = ctypes.c_int
M = int(os.environ.get("GEMM_M", "12736"))


def run(name, N, K, **kw):
    a = (torch.rand(M, K, device="cuda") * 2 - 1).to(torch.bfloat16); w = (torch.rand(N, K, device="cuda") * 2 - 1).to(torch.bfloat16)
    out = torch.empty(M, N, device="cuda", dtype=kw.pop("odt", torch.bfloat16))
    if kw.pop("bias", False): kw["bias"] = torch.randn(N, device="cuda")
    if kw.pop("c2", False): kw["C2"] = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    if kw.pop("res", False): kw["R"] = torch.randn(M, N, device="cuda")
    for _ in range(5): ops.linear(a, w, out=out, **kw)
    torch.cuda.synchronize()
    nb = 4096
    buf = np.zeros((nb, 4), dtype=np.uint64)
    assert lib.av_gemm_stamps_read(buf.ctypes.data, nb) == 0
    st = buf[buf[:, 3] > 0].astype(np.int64)
    st = st[st[:, 0] > st[:, 3].max() - 100 * 1000]             # drop stale entries of earlier, larger launches (older than 1 ms)
    t0 = st[:, 0].min()
    us = (st - t0) / 100.0                                      # 100 MHz
    n = len(us)
    whole = us[:, 2] - us[:, 1] > 0.6 * np.median(us[:, 2] - us[:, 1])
    print(f"{name:30s} N={N} K={K}: {n} workgroups, launch span {us[:, 3].max():7.1f} us | per workgroup (median): entry->first phase "
          f"{np.median(us[:, 1] - us[:, 0]):5.2f}  main loop {np.median((us[:, 2] - us[:, 1])[whole]):6.2f}  epilogue+stores {np.median(us[:, 3] - us[:, 2]):6.2f} us")
    order = np.argsort(us[:, 0])
    starts = us[order, 0]
    rounds = [0] + [i for i in range(1, n) if starts[i] - starts[i - 1] > 1.0]
    for ri, i0 in enumerate(rounds):
        i1 = rounds[ri + 1] if ri + 1 < len(rounds) else n
        seg = us[order[i0:i1]]
        print(f"    wave {ri}: {i1 - i0:4d} workgroups start {seg[:, 0].min():7.1f}..{seg[:, 0].max():7.1f}  loop-end {np.median(seg[:, 2]):7.1f}  exit {np.median(seg[:, 3]):7.1f}..{seg[:, 3].max():7.1f}")


run("plain", 4096, 1024)
run("bias+gelu_gf+C2+drop (FFN up)", 4096, 1024, bias=True, act=L.ACT_GELU_GF, c2=True, drop=(0.1, 1234, 5))
run("plain", 1024, 4096)
run("bias+res f32 (out-proj)", 1024, 1024, bias=True, res=True, odt=torch.float32)
