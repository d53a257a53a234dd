"""Per-shape timing of the ResNet-18 trunk convolutions of layers 2-4 (implicit-GEMM path) at 6400 frames (B = 64 x 100), with and
without the BatchNorm-partials epilogue, against a plain GEMM of the same M x N x K."""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("multimodal-av-model_amd.ops"); L = importlib.import_module("multimodal-av-model_amd._lib")
NF = int(os.environ.get("FRAMES", "6400"))


def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / n


tot = 0.0
for name, Hin, Cin, Cout, k, stride, cnt in [("layer2.0.conv1", 24, 64, 128, 3, 2, 1), ("layer2 3x3", 12, 128, 128, 3, 1, 3), ("layer2 down 1x1", 24, 64, 128, 1, 2, 1),
                                             ("layer3.0.conv1", 12, 128, 256, 3, 2, 1), ("layer3 3x3", 6, 256, 256, 3, 1, 3), ("layer3 down 1x1", 12, 128, 256, 1, 2, 1),
                                             ("layer4.0.conv1", 6, 256, 512, 3, 2, 1), ("layer4 3x3", 3, 512, 512, 3, 1, 3), ("layer4 down 1x1", 6, 256, 512, 1, 2, 1)]:
    pad = k // 2
    Ho = (Hin + 2 * pad - k) // stride + 1
    M = NF * Ho * Ho
    K = k * k * Cin
    x = torch.randn(NF * Hin * Hin, Cin, device="cuda").to(torch.bfloat16)
    wk = (torch.randn(Cout, K, device="cuda") / K ** 0.5).to(torch.bfloat16)
    y = torch.empty(M, Cout, device="cuda", dtype=torch.bfloat16)
    st = torch.empty((M + 127) // 128, 2, Cout, device="cuda")
    geo = dict(cT=1, cH=Hin, cW=Hin, cCtot=Cin, cCin=Cin, cCoff=0, cKt=1, cKh=k, cKw=k, cSh=stride, cSw=stride, cPt=0, cPh=pad, cPw=pad, cOh=Ho, cOw=Ho)
    xa = torch.randn(M, K, device="cuda").to(torch.bfloat16) if M * K < 3e9 else None
    res = []
    res.append(timeit(lambda: ops.gemm(x, wk, y, M=M, N=Cout, K=K, lda=0, ldb=K, ldc=Cout, a_mode=L.A_CONV2D, conv=geo, stats=st)))
    res.append(timeit(lambda: ops.gemm(x, wk, y, M=M, N=Cout, K=K, lda=0, ldb=K, ldc=Cout, a_mode=L.A_CONV2D, conv=geo)))
    res.append(timeit(lambda: ops.linear(xa, wk, None, out=y)) if xa is not None else float("nan"))
    fl = 2.0 * M * Cout * K
    tot += res[0] * cnt
    print(f"{name:16s} M={M:7d} N={Cout:4d} K={K:5d} x{cnt}: conv+stats {res[0]:7.1f} us ({fl / res[0] / 1e6:6.0f} TF/s)  conv {res[1]:7.1f} us ({fl / res[1] / 1e6:6.0f})  "
          f"plain GEMM {res[2]:7.1f} us ({fl / res[2] / 1e6:6.0f})", flush=True)
print(f"total per visual-encoder call: {tot / 1000:.2f} ms")
