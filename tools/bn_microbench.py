"""BN-apply (+ residual) + PReLU pass in isolation at the lip encoder's activation sizes (3200 frames per pass): achieved HBM bytes/s."""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("multimodal-av-model_amd.ops"); L = importlib.import_module("multimodal-av-model_amd._lib")
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / n
for (hw, C) in ((576, 64), (144, 128), (36, 256), (9, 512)):
    n = 3200 * hw * C
    x = torch.randn(n, device="cuda").to(torch.bfloat16); r = torch.randn(n, device="cuda").to(torch.bfloat16); o = torch.empty_like(x)
    sc = torch.rand(C, device="cuda") + 0.5; sh = torch.randn(C, device="cuda"); sl = torch.full((C,), 0.25, device="cuda")
    for res in (False, True):
        fn = lambda: L.check(L.lib().av_bn_act(ops.ptr(x), ops.ptr(sc), ops.ptr(sh), ops.ptr(r) if res else None, None, None, ops.ptr(sl), ops.ptr(o),
                                               1, n, C, ops.stream()), "av_bn_act")
        us = timeit(fn)
        gb = n * 2 * (3 if res else 2) / 1e9
        print(f"hw={hw:4d} C={C:4d} res={int(res)}: {us:7.1f} us  {gb / us * 1e6 / 1e3:6.2f} TB/s ({gb * 1e3:.0f} MB)", flush=True)
print("-- size sweep at C=512 (no residual) against a plain torch copy of the same bytes")
C = 512
sc = torch.rand(C, device="cuda") + 0.5; sh = torch.randn(C, device="cuda"); sl = torch.full((C,), 0.25, device="cuda")
for frames in (400, 800, 1600, 3200, 6400, 12800):
    n = frames * 9 * C
    x = torch.randn(n, device="cuda").to(torch.bfloat16); o = torch.empty_like(x)
    us = timeit(lambda: L.check(L.lib().av_bn_act(ops.ptr(x), ops.ptr(sc), ops.ptr(sh), None, None, None, ops.ptr(sl), ops.ptr(o), 1, n, C, ops.stream()), "av_bn_act"))
    us_c = timeit(lambda: o.copy_(x))
    print(f"n={n * 2 / 1e6:7.1f} MB: bn_act {us:6.1f} us, copy_ {us_c:6.1f} us", flush=True)
