"""Timing of av_attention_fwd / av_attention_bwd (bf16) on one MI355X at the wav2vec2 shapes; set AVAMD_ATTN_SHORT=0 for the tiled kernels."""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("multimodal-av-model_amd.ops")

for (B, H, T, D) in [(32, 16, 199, 64), (64, 16, 199, 64), (2, 16, 49, 64), (8, 16, 256, 64), (8, 16, 749, 64)]:
    qkv = torch.randn(B, T, 3, H, D, device="cuda").to(torch.bfloat16)
    q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
    klen = torch.full((B,), T, device="cuda", dtype=torch.int32)
    do = torch.randn(B, T, H, D, device="cuda").to(torch.bfloat16)
    dqkv = torch.empty_like(qkv)
    scale = D ** -0.5

    def fwd(drop=None):
        return ops.attention_fwd(q, k, v, klen, scale, drop=drop)

    o, lse = fwd()

    def bwd(drop=None):
        ops.attention_bwd(q, k, v, do, dqkv[:, :, 0], dqkv[:, :, 1], dqkv[:, :, 2], klen, scale, o=o, lse=lse, drop=drop)

    dr = (0.1, 1234, 3)
    mk = ops.attention_dropmask(B, H, T, T, dr, q.device) if T <= 256 else None
    for name, fn, fl in (("keep-bit kernel", lambda: ops.attention_dropmask(B, H, min(T, 256), min(T, 256), dr, q.device), 0.0),
                         ("fwd+dropout (bits)", lambda: ops.attention_fwd(q, k, v, klen, scale, drop=dr, drop_mask=mk), 4.0 * B * H * T * T * D),
                         ("bwd+dropout (bits)", lambda: ops.attention_bwd(q, k, v, do, dqkv[:, :, 0], dqkv[:, :, 1], dqkv[:, :, 2], klen, scale, o=o, lse=lse, drop=dr, drop_mask=mk), 10.0 * B * H * T * T * D),("fwd", fwd, 4.0 * B * H * T * T * D), ("bwd", bwd, 10.0 * B * H * T * T * D),
                         ("fwd+dropout", lambda: fwd(dr), 4.0 * B * H * T * T * D), ("bwd+dropout", lambda: bwd(dr), 10.0 * B * H * T * T * D)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1000 / 20
        print(f"B={B} H={H} T={T} D={D} {name:24s}: {us:8.1f} us  {fl / us / 1e6:7.1f} TF/s", flush=True)
