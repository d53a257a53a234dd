"""Device side of the reference's ``load_pair`` (dataset/multi_speaker_dataset.py:13-59).

The reference decodes a wav and an npy file per speaker and then, on the host and per sample, averages the colour channels of every
128×128 lip frame, resizes it to 96×96 with cv2, scales by 1/255, mixes the two clips, peak-normalises and builds the two speaker
masks.  Here the decoded arrays are uploaded once (uint8 frames: 49 KB per frame instead of the 37 KB fp32 result, but no host
arithmetic at all) and the arithmetic runs in two HIP kernels (csrc/preprocess.hip) with the reference's float32 operation order.
Decoding the files (librosa / np.load) stays with the caller; there is no CPU fallback."""
from __future__ import annotations

from typing import Dict

import numpy as np
import torch

from .. import _lib as L
from .. import ops


def _dev(x, dtype=None) -> torch.Tensor:
    t = torch.as_tensor(x)
    if dtype is not None and t.dtype != dtype:
        t = t.to(dtype)
    if not t.is_cuda:
        if not torch.cuda.is_available():
            raise RuntimeError("device_pipeline: needs the HIP device (no CPU fallback)")
        t = t.cuda(non_blocking=True)
    return t.contiguous()


def lips_to_device(frames, size: int = 96) -> torch.Tensor:
    """frames [T, H, W, C] uint8 or float -> fp32 CUDA [T, 1, size, size] (:49-53: mean over C, bilinear resize, / 255)."""
    t = torch.as_tensor(frames)
    if t.dim() != 4:
        raise ValueError(f"lips_to_device: expected [T, H, W, C], got {tuple(t.shape)}")
    if t.shape[0] == 0:
        raise RuntimeError("lips_to_device: empty lip clip")                        # :59-60
    u8 = t.dtype == torch.uint8
    src = _dev(t, None if u8 else torch.float32)
    T, Hs, Ws, C = src.shape
    out = torch.empty((T, 1, size, size), dtype=torch.float32, device=src.device)
    L.check(L.lib().av_lip_gray_resize(ops.ptr(src), int(u8), ops.ptr(out), T, Hs, Ws, C, size, size, 255.0, ops.stream()), "av_lip_gray_resize")
    return out


def mix_pair(a1, a2) -> Dict[str, torch.Tensor]:
    """a1, a2: 1-D float waveforms (16 kHz, already cut to the sentence) -> audio fp32 [n], mask1 / mask2 int64 [n] on the device (:21-45)."""
    x1, x2 = _dev(a1, torch.float32).view(-1), _dev(a2, torch.float32).view(-1)
    n = max(x1.numel(), x2.numel())
    dev = x1.device
    mixed = torch.empty(n, dtype=torch.float32, device=dev)
    m1 = torch.empty(n, dtype=torch.long, device=dev); m2 = torch.empty(n, dtype=torch.long, device=dev)
    ws = torch.empty(1, dtype=torch.int32, device=dev)
    L.check(L.lib().av_mix_pair(ops.ptr(x1) if x1.numel() else None, x1.numel(), ops.ptr(x2) if x2.numel() else None, x2.numel(),
                                ops.ptr(mixed), ops.ptr(m1), ops.ptr(m2), ops.ptr(ws), ops.stream()), "av_mix_pair")
    return {"audio": mixed, "mask1": m1, "mask2": m2}


def load_pair_device(a1, a2, frames1, frames2, label1, label2) -> dict:
    """The dict ``load_pair`` returns (:70-84), built from decoded arrays; tensors stay on the device (collate_fn pads them there)."""
    out = mix_pair(a1, a2)
    lip1, lip2 = lips_to_device(frames1), lips_to_device(frames2)
    out.update({"lip1": lip1, "label1": np.asarray(label1, dtype=np.int64), "lip1_len": lip1.shape[0],
                "lip2": lip2, "label2": np.asarray(label2, dtype=np.int64), "lip2_len": lip2.shape[0]})
    return out
