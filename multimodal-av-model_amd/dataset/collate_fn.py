"""Batch collation with the reference's output contract (dataset/collate_fn.py:4-63).

Zero-pads lips / labels / waveform to the batch maximum and pads the speaker masks with code 3
(= padding, dataset/collate_fn.py:40,44).  Returns the same 12 keys.
"""
from __future__ import annotations

from typing import Dict, List

import torch


def _pad_stack(seqs: List[torch.Tensor], value=0) -> torch.Tensor:
    n = max(int(s.shape[0]) for s in seqs)
    out = seqs[0].new_full((len(seqs), n) + tuple(seqs[0].shape[1:]), value)
    for i, s in enumerate(seqs):
        out[i, : s.shape[0]] = s
    return out


def collate_fn(batch: List[dict]) -> Dict[str, torch.Tensor]:
    out: Dict[str, torch.Tensor] = {}
    for spk in ("1", "2"):
        lips = [torch.as_tensor(it["lip" + spk]) for it in batch]              # [T,1,H,W]
        out["lip" + spk] = _pad_stack(lips)
        out["lip" + spk + "_lengths"] = torch.tensor([int(l.shape[0]) for l in lips])
        txt = [torch.as_tensor(it["label" + spk], dtype=torch.long) for it in batch]
        out["text" + spk] = _pad_stack(txt)
        out["text" + spk + "_lengths"] = torch.tensor([int(t.shape[0]) for t in txt])
        msk = [torch.as_tensor(it["mask" + spk], dtype=torch.long) for it in batch]
        out["mask" + spk] = _pad_stack(msk, 3)
    aud = [torch.as_tensor(it["audio"], dtype=torch.float32) for it in batch]
    out["audio"] = _pad_stack(aud)
    out["audio_lengths"] = torch.tensor([int(a.shape[0]) for a in aud])
    return out
