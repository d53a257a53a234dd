import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("multimodal-av-model_amd.ops"); L = importlib.import_module("multimodal-av-model_amd._lib")
M, N, K = 6368, 4096, 1024
a = (torch.rand(M, K, device="cuda") * 2 - 1).to(torch.bfloat16); w = (torch.rand(N, K, device="cuda") * 2 - 1).to(torch.bfloat16)
bias = torch.randn(N, device="cuda"); u = torch.randn(M, N, device="cuda").to(torch.bfloat16)
def run(name, **kw):
    out = torch.empty(M, N, device="cuda", dtype=kw.pop("odt", torch.bfloat16))
    for _ in range(3): ops.linear(a, w, out=out, **kw)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ops.linear(a, w, out=out, **kw)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 50
    print(f"{name:22s} {us:8.1f} us {2.0*M*N*K/us/1e6:7.1f} TF/s", flush=True)
run("plain")
run("bias", bias=bias)
run("gelu", act=L.ACT_GELU)
run("C2", C2=torch.empty(M, N, device="cuda", dtype=torch.bfloat16))
run("bias+gelu", bias=bias, act=L.ACT_GELU)
run("bias+gelu+C2", bias=bias, act=L.ACT_GELU, C2=torch.empty(M, N, device="cuda", dtype=torch.bfloat16))
run("mul_gelu_grad", act=L.ACT_MUL_GELU_GRAD, aux=u)
run("plain f32 out", odt=torch.float32)
