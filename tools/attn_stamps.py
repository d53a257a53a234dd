"""Where does a workgroup of the K7 forward spend its time?  DIAGNOSTIC build of attention_short.hip (-DAV_ATTN_STAMPS; never the product library):
s_memrealtime (100 MHz) of thread 0 of every workgroup at entry / operands landed (after the barrier) / start of wavefront 0's second query
tile / exit (stores complete).  64 x 16 heads x 199 frames, keep-bit dropout as in the step.  usage (GPU box): python tools/attn_stamps.py"""
import ctypes, importlib, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "multimodal-av-model_amd")
out_dir = os.path.join(ROOT, "tools", "_bin"); os.makedirs(out_dir, exist_ok=True)
lib_path = os.path.join(out_dir, "libavhip_attn_stamps.so")
if len(sys.argv) == 1:
    o = os.path.join(out_dir, "attention_short_stamps.o")
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-ffp-contract=off", "-Wno-unused-value", "-DAV_ATTN_STAMPS",
                           "-I" + os.path.join(ROOT, "include"), "-c", os.path.join(PKG, "csrc", "attention_short.hip"), "-o", o])
    objs = [os.path.join(PKG, "build", f[:-4] + ".o") for f in sorted(os.listdir(os.path.join(PKG, "csrc"))) if f.endswith(".hip") and f != "attention_short.hip"]
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib_path, o] + objs)
    sys.exit(subprocess.call([sys.executable, os.path.abspath(__file__), "run"]))
import torch
sys.path.insert(0, ROOT)
os.environ["AVAMD_LIB"] = lib_path
L = importlib.import_module("multimodal-av-model_amd._lib"); ops = importlib.import_module("multimodal-av-model_amd.ops")
lib = L.lib()
lib.av_attn_stamps_read.argtypes = [ctypes.c_void_p]; lib.av_attn_stamps_read.restype = ctypes.c_int
B, H, T, D = 64, 16, 199, 64
qkv = torch.randn(B, T, 3, H, D, device="cuda").to(torch.bfloat16)
q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
klen = torch.full((B,), T, device="cuda", dtype=torch.int32)
for drop in (None, (0.1, 1234, 3)):
    mk = ops.attention_dropmask(B, H, T, T, drop, q.device) if drop else None
    for _ in range(4):
        ops.attention_fwd(q, k, v, klen, D ** -0.5, drop=drop, drop_mask=mk)
    torch.cuda.synchronize()
    buf = np.zeros((4096, 4), dtype=np.uint64)
    assert lib.av_attn_stamps_read(buf.ctypes.data) == 0
    st = buf[:B * H].astype(np.int64)
    t0 = st[:, 0].min()
    us = (st - t0) / 100.0
    order = np.argsort(us[:, 0])
    first = us[us[:, 0] < 2.0]; second = us[us[:, 0] >= 2.0]
    print(f"{'keep bits' if drop else 'no dropout'}: launch spans {us[:, 3].max():.1f} us; {len(first)} workgroups start within 2 us, {len(second)} later")
    for name, g in (("first round", first), ("later rounds", second)):
        if len(g) == 0:
            continue
        print(f"  {name:12s}: start {np.median(g[:, 0]):6.2f}  operands landed +{np.median(g[:, 1] - g[:, 0]):5.2f}  first query tile +{np.median(g[:, 2] - g[:, 1]):5.2f}"
              f"  second tile + stores +{np.median(g[:, 3] - g[:, 2]):5.2f}  (ends {np.median(g[:, 3]):6.2f}, max {g[:, 3].max():6.2f})")
