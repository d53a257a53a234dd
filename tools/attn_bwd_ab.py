"""K7 backward at the step's size (64 x 16 heads x 199 x 64, packed QKV, keep-bit dropout): one-pass kernel vs the two-phase kernel (AVAMD_ATTN_BWD2=0)."""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("multimodal-av-model_amd.ops")
torch.manual_seed(0)
B, H, T, D = 64, 16, int(os.environ.get("ATTN_T", "199")), 64
qkv = torch.randn(B, T, 3, H, D, device="cuda").to(torch.bfloat16)
q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
do = torch.randn(B, T, H, D, device="cuda").to(torch.bfloat16)
scale = D ** -0.5
for drop in (None, (0.1, 77, 3)):
    mask = ops.attention_dropmask(B, H, T, T, drop, "cuda") if drop else None
    kw = dict(drop=drop, drop_mask=mask) if drop else {}
    o, lse = ops.attention_fwd(q, k, v, None, scale, **kw)
    dqkv = torch.empty_like(qkv)
    dq, dk, dv = dqkv[:, :, 0], dqkv[:, :, 1], dqkv[:, :, 2]
    f = lambda: ops.attention_bwd(q, k, v, do, dq, dk, dv, None, scale, o=o, lse=lse, **kw)
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 50
    fl = 10.0 * B * H * T * T * D
    print(f"T={T} drop={'keep bits' if drop else 'none':9s}: {us:7.1f} us  {fl / us / 1e6:6.1f} TF/s   checksum dq {float(dq.float().abs().sum()):.3f} dk {float(dk.float().abs().sum()):.3f} dv {float(dv.float().abs().sum()):.3f}", flush=True)
