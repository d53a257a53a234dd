"""TEST INFRASTRUCTURE — numpy restatement of the per-sample arithmetic of the reference's ``load_pair``
(/root/reference/dataset/multi_speaker_dataset.py:13-59), i.e. what happens to a sample after the wav / npy files are decoded.

Parity status: **parity unpinned by the reference** for this file — ``dataset.multi_speaker_dataset`` cannot be imported here
(librosa and cv2 are not installed: ordinary ModuleNotFoundError) and the reference holds no fixtures for it.  The mix / mask law
is plain numpy and is restated operation by operation; ``cv2.resize`` is restated from the published INTER_LINEAR law
(pixel centres: src = (dst + 0.5) * scale - 0.5 evaluated in double and cast to float, taps clamped to the image, horizontal pass
then vertical pass in float32).  tests/test_oracle_cpu.py pins it with known answers (identity, linear ramps, hand-made masks).
"""
from __future__ import annotations

import numpy as np


def mix_pair(a1: np.ndarray, a2: np.ndarray):
    """:21-45 — zero-pad to the longer clip, add, peak-normalise; masks 1 = both speak, 2 = only this speaker, 0 = otherwise."""
    a1 = np.asarray(a1, dtype=np.float32); a2 = np.asarray(a2, dtype=np.float32)
    len1, len2 = len(a1), len(a2)
    max_len = max(len1, len2)
    a1 = np.pad(a1, (0, max_len - len1), mode="constant")
    a2 = np.pad(a2, (0, max_len - len2), mode="constant")
    mixed = (a1 + a2).astype(np.float32)
    if max_len:
        mixed /= np.max(np.abs(mixed)) + np.float32(1e-6)          # float32 arithmetic (numpy >= 2 scalar promotion)
    mask1 = np.zeros(max_len, dtype=np.int64); mask2 = np.zeros(max_len, dtype=np.int64)
    min_len = min(len1, len2)
    mask1[:min_len] = 1; mask2[:min_len] = 1
    if len1 > len2:
        mask1[len2:len1] = 2
    elif len2 > len1:
        mask2[len1:len2] = 2
    return mixed, mask1, mask2


def _taps(n_dst: int, n_src: int):
    scale = float(n_src) / float(n_dst)                            # double, as cv2 (1 / inv_scale)
    f = ((np.arange(n_dst, dtype=np.float64) + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    lo = s < 0
    s[lo] = 0; f[lo] = 0.0
    hi = s >= n_src - 1
    s[hi] = n_src - 1; f[hi] = 0.0
    s1 = np.minimum(s + 1, n_src - 1)
    return s, s1, (np.float32(1.0) - f).astype(np.float32), f


def resize_bilinear(img: np.ndarray, hd: int, wd: int) -> np.ndarray:
    """float32 [Hs, Ws] -> [hd, wd]; horizontal pass, then vertical pass, every product and sum rounded to float32."""
    img = np.asarray(img, dtype=np.float32)
    sy, sy1, b0, b1 = _taps(hd, img.shape[0])
    sx, sx1, a0, a1 = _taps(wd, img.shape[1])
    h = (img[:, sx] * a0[None, :]).astype(np.float32) + (img[:, sx1] * a1[None, :]).astype(np.float32)          # [Hs, wd]
    return ((h[sy] * b0[:, None]).astype(np.float32) + (h[sy1] * b1[:, None]).astype(np.float32)).astype(np.float32)


def lips(frames: np.ndarray, size: int = 96) -> np.ndarray:
    """:49-53 — frames [T, H, W, C] (uint8 or float) -> float32 [T, 1, size, size]."""
    x = np.asarray(frames).astype(np.float32)
    s = x[..., 0].copy()
    for c in range(1, x.shape[-1]):                               # np.mean(axis=-1) of float32: sequential float32 sum, then / C
        s = (s + x[..., c]).astype(np.float32)
    gray = (s / np.float32(x.shape[-1])).astype(np.float32)
    out = np.stack([resize_bilinear(f, size, size) for f in gray]) if len(gray) else np.zeros((0, size, size), np.float32)
    return (out / np.float32(255.0)).astype(np.float32)[:, None]
