"""Where does a workgroup of the fused cross-attention forward (two items per workgroup) spend its time?  DIAGNOSTIC build of fusion_attn.hip
(-DAV_K17_STAMPS), never the product library.  usage (GPU box): python tools/k17_stamps.py"""
import ctypes, importlib, os, subprocess, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = os.path.join(ROOT, "multimodal-av-model_amd")
out_dir = os.path.join(ROOT, "tools", "_bin"); os.makedirs(out_dir, exist_ok=True)
lib_path = os.path.join(out_dir, "libavhip_k17stamps.so")
objs = []
for s in sorted(f for f in os.listdir(os.path.join(PKG, "csrc")) if f.endswith(".hip")):
    o = os.path.join(PKG, "build", s[:-4] + ".o")
    if s == "fusion_attn.hip":
        o = os.path.join(out_dir, "fusion_attn_stamps.o")
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-ffp-contract=off", "-Wno-unused-value",
                               "-DAV_K17_STAMPS", "-c", os.path.join(PKG, "csrc", s), "-o", o])
    objs.append(o)
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib_path] + objs)
os.environ["AVAMD_LIB"] = lib_path
L = importlib.import_module("multimodal-av-model_amd._lib"); ops = importlib.import_module("multimodal-av-model_amd.ops")
lib = L.lib()
lib.av_k17_stamps_read.argtypes = [ctypes.c_void_p, ctypes.c_int]; lib.av_k17_stamps_read.restype = ctypes.c_int
B, T, E, nh = 128, 100, 512, 4
a = torch.randn(B, T, E, device="cuda").to(torch.bfloat16); v = torch.randn(B, T, E, device="cuda").to(torch.bfloat16)
w = (torch.randn(3 * E, E, device="cuda") / E ** 0.5).to(torch.bfloat16); bias = torch.randn(3 * E, device="cuda") * 0.1
for save in (True, False):
    for _ in range(5):
        ops.fusion_xattn_fwd(a, v, w, bias, nh, 128 ** -0.5, save)
    torch.cuda.synchronize()
    buf = np.zeros((1024, 8), dtype=np.uint64)
    assert lib.av_k17_stamps_read(buf.ctypes.data, 1024) == 0
    st = buf[buf[:, 4] > 0][:, :5].astype(np.int64)
    us = (st - st[:, 0].min()) / 100.0
    d = np.diff(us, axis=1)
    print(f"save={save}: {len(us)} workgroups, span {us[:, 4].max():.1f} us; median per workgroup: Q pass {np.median(d[:, 0]):.2f}  K/V pass {np.median(d[:, 1]):.2f}  "
          f"images {np.median(d[:, 2]):.2f}  attention {np.median(d[:, 3]):.2f} us; start spread {us[:, 0].max():.2f} us")
