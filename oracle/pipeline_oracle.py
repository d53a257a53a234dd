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


# ---- band-limited resampling (the stage librosa.load(path, sr=16000) adds when the file's rate differs, :15,18) -------------------
# librosa's default resampler is soxr_hq (not installed here, filter not published as a formula): **parity unpinned**.  What is
# restated is the published windowed-sinc-with-interpolated-table law (J. O. Smith, "Digital Audio Resampling"; resampy's
# "kaiser_best" parameters), which the device kernel implements; known answers (identity, pure tones) pin this restatement.
def sinc_table(ratio: float, num_zeros: int = 64, precision: int = 9, rolloff: float = 0.9475937167399596, beta: float = 14.769656459379492):
    num_table = 2 ** precision
    n = num_table * num_zeros
    win = np.kaiser(2 * n + 1, beta)[n:] * (rolloff * np.sinc(rolloff * np.linspace(0, num_zeros, num=n + 1, endpoint=True)))
    if ratio < 1.0:
        win = win * ratio
    delta = np.zeros_like(win)
    delta[:-1] = np.diff(win)
    return win.astype(np.float32).astype(np.float64), delta.astype(np.float32).astype(np.float64), num_table


def resample_sinc(x: np.ndarray, sr_in: int, sr_out: int) -> np.ndarray:
    """float32 [n] at sr_in -> float32 [ceil(n * sr_out / sr_in)] at sr_out (double accumulation, float32 table)."""
    x = np.asarray(x, dtype=np.float32)
    if sr_in == sr_out:
        return x
    ratio = float(sr_out) / float(sr_in)
    win, delta, num_table = sinc_table(ratio)
    nwin = len(win)
    scale = min(1.0, ratio)
    index_step = int(scale * num_table)
    n_in = len(x)
    n_out = int(np.ceil(n_in * ratio))
    t = np.arange(n_out, dtype=np.float64) * (1.0 / ratio)
    n = t.astype(np.int64)
    xd = x.astype(np.float64)
    y = np.zeros(n_out, dtype=np.float64)
    frac = scale * (t - n)
    for wing in (0, 1):
        if wing == 1:
            frac = scale - frac
        index_frac = frac * num_table
        offset = index_frac.astype(np.int64)
        eta = index_frac - offset
        cnt = (nwin - offset) // index_step
        cnt = np.minimum(cnt, n + 1 if wing == 0 else n_in - n - 1)
        for i in range(int(cnt.max()) if n_out else 0):
            m = cnt > i
            k = offset[m] + i * index_step
            src = n[m] - i if wing == 0 else n[m] + i + 1
            y[m] += (win[k] + eta[m] * delta[k]) * xd[src]
    return y.astype(np.float32)
