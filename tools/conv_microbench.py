"""ResNet layer1 convolution (3x3, 64->64, 24x24, 3200 frames) timing: weights-stationary kernel vs implicit-GEMM kernel."""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("multimodal-av-model_amd.ops"); L = importlib.import_module("multimodal-av-model_amd._lib")
N, H, W = 3200, 24, 24
M = N * H * W
x = torch.randn(M, 64, device="cuda").to(torch.bfloat16)
wk = (torch.randn(64, 576, device="cuda") / 24).to(torch.bfloat16)
y = torch.empty(M, 64, device="cuda", dtype=torch.bfloat16)
st1 = torch.empty((M + 255) // 256, 2, 64, device="cuda"); st2 = torch.empty((M + 127) // 128, 2, 64, device="cuda")
geo = dict(cT=1, cH=H, cW=W, cCtot=64, cCin=64, cCoff=0, cKt=1, cKh=3, cKw=3, cSh=1, cSw=1, cPt=0, cPh=1, cPw=1, cOh=H, cOw=W)


def fast():
    L.check(L.lib().av_conv3x3_c64(ops.ptr(x), ops.ptr(wk), ops.ptr(y), ops.ptr(st1), N, H, W, None, None, None, ops.stream()), "c64")


def gemm():
    ops.gemm(x, wk, y, M=M, N=64, K=576, lda=0, ldb=576, ldc=64, a_mode=L.A_CONV2D, conv=geo, stats=st2)


for name, fn in (("weights-stationary", fast), ("implicit GEMM", gemm)):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    print(f"{name:20s} {us:8.1f} us  {2.0 * M * 64 * 576 / us / 1e6:7.1f} TF/s", flush=True)
