"""Synthetic clips with the output contract of the reference dataset (dataset/multi_speaker_dataset.py:13-84).

SURVEY §8(d): waveform N(0,1) peak-normalised (``:30-32``), lips U[0,1) ``[25*sec,1,96,96]`` (``:49-53``),
speaker-2 length fraction 0.75 => mask1 = 1 on [0,.75n), 2 on [.75n,n); mask2 = 1 on [0,.75n), 0 after
(``:35-45``); labels uniform in [4,800).  ``frac`` < 1 makes a ragged item (shorter clip => collate pads).
"""
from __future__ import annotations

from typing import Dict, List

import numpy as np
import torch

from .collate_fn import collate_fn


def make_item(gen: torch.Generator, seconds: float, frac: float = 1.0, fps: int = 25, vocab: int = 800,
              spk2_frac: float = 0.75) -> dict:
    n = int(round(16000 * seconds * frac))
    tv = max(1, int(round(fps * seconds * frac)))
    audio = torch.randn(n, generator=gen, dtype=torch.float32)
    audio = audio / (audio.abs().max() + 1e-6)
    n2 = int(spk2_frac * n)
    mask1 = np.zeros(n, dtype=np.int64); mask2 = np.zeros(n, dtype=np.int64)
    mask1[:n2] = 1; mask2[:n2] = 1; mask1[n2:] = 2
    l1 = max(1, int(5 * seconds * frac)); l2 = max(1, int(spk2_frac * 5 * seconds * frac))
    return {
        "audio": audio.numpy(), "mask1": mask1, "mask2": mask2,
        "lip1": torch.rand((tv, 1, 96, 96), generator=gen, dtype=torch.float32),
        "lip2": torch.rand((tv, 1, 96, 96), generator=gen, dtype=torch.float32),
        "label1": torch.randint(4, vocab, (l1,), generator=gen).numpy(),
        "label2": torch.randint(4, vocab, (l2,), generator=gen).numpy(),
    }


def make_batch(batch_size: int, seconds: float, seed: int = 42, ragged: bool = False, vocab: int = 800
               ) -> Dict[str, torch.Tensor]:
    gen = torch.Generator(device="cpu"); gen.manual_seed(seed)
    fr = [1.0, 0.75, 0.5]
    items: List[dict] = [make_item(gen, seconds, fr[i % 3] if ragged else 1.0, vocab=vocab) for i in range(batch_size)]
    return collate_fn(items)
