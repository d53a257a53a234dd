import importlib, os, sys, torch
sys.path.insert(0, "/root/repo" if os.path.isdir("/root/repo/multimodal-av-model_amd") else os.environ.get("GRAFT_REPO_ROOT","."))
ops = importlib.import_module("multimodal-av-model_amd.ops"); L = importlib.import_module("multimodal-av-model_amd._lib")
N,H,W=3200,24,24
x=torch.randn(N,H,W,64,device="cuda").to(torch.bfloat16); wk=(torch.randn(64,576,device="cuda")/24).to(torch.bfloat16)
y=torch.empty(N*H*W,64,device="cuda",dtype=torch.bfloat16); st=torch.empty((N*H*W+255)//256,2,64,device="cuda")
sc=torch.rand(64,device="cuda")+0.5; sh=torch.randn(64,device="cuda"); sl=torch.full((64,),0.25,device="cuda")
def run(xf):
    L.check(L.lib().av_conv3x3_c64(ops.ptr(x),ops.ptr(wk),ops.ptr(y),ops.ptr(st),N,H,W,ops.ptr(sc) if xf else None,ops.ptr(sh) if xf else None,ops.ptr(sl) if xf else None,ops.stream()),"c64")
for xf in (False,True,False,True):
    for _ in range(3): run(xf)
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): run(xf)
    e1.record(); torch.cuda.synchronize()
    print("xf" if xf else "plain", f"{e0.elapsed_time(e1)*100:.1f} us")
