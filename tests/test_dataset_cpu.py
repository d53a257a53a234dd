"""SURVEY §8(f)-2 host side: native WAV decoding (csrc/wavio.hip) against Python's ``wave`` module and numpy, the oracle's
resampler against known answers, and the pairing / retry law of the dataset classes (dataset/multi_speaker_dataset.py:87-142)."""
import ctypes as C
import os
import random
import struct
import wave

import numpy as np
import pytest
import torch

from conftest import pkg
from oracle import pipeline_oracle as PO


def _write_pcm(path, data, sr, width):
    """data [n, ch] integer samples -> RIFF/WAVE through the stdlib writer (24-bit packed by hand)."""
    n, ch = data.shape
    with wave.open(path, "wb") as w:
        w.setnchannels(ch); w.setsampwidth(width); w.setframerate(sr)
        if width == 1:
            w.writeframes(data.astype(np.uint8).tobytes())
        elif width == 2:
            w.writeframes(data.astype("<i2").tobytes())
        elif width == 4:
            w.writeframes(data.astype("<i4").tobytes())
        else:
            b = data.astype("<i4").reshape(-1).view(np.uint8).reshape(-1, 4)[:, :3]
            w.writeframes(b.tobytes())


def _write_float(path, data, sr, extra_chunk=True):
    n, ch = data.shape
    payload = data.astype("<f4").tobytes()
    fmt = struct.pack("<HHIIHH", 3, ch, sr, sr * ch * 4, ch * 4, 32)
    chunks = b"fmt " + struct.pack("<I", len(fmt)) + fmt
    if extra_chunk:
        chunks += b"LIST" + struct.pack("<I", 5) + b"abcde" + b"\x00"          # odd-sized chunk + pad byte before the data
    chunks += b"data" + struct.pack("<I", len(payload)) + payload
    with open(path, "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", 4 + len(chunks)) + b"WAVE" + chunks)


def _read(path):
    ds = pkg("dataset.multi_speaker_dataset")
    buf, sr = ds.AudioStore.decode_host(path)
    return buf.numpy().copy(), sr


@pytest.mark.parametrize("width,ch", [(1, 1), (2, 1), (2, 2), (3, 2), (4, 3)])
def test_wav_pcm_decoding_is_exact(tmp_path, width, ch):
    rng = np.random.default_rng(width * 10 + ch)
    n = 70001                                                              # crosses the reader's 65536-frame chunk
    if width == 1:
        data = rng.integers(0, 256, (n, ch))
        f = (data.astype(np.float32) - 128.0) / np.float32(128.0)
    else:
        bits = 8 * width
        data = rng.integers(-(1 << (bits - 1)), 1 << (bits - 1), (n, ch))
        data[0] = -(1 << (bits - 1)); data[1] = (1 << (bits - 1)) - 1
        f = (data.astype(np.float64) / float(1 << (bits - 1))).astype(np.float32)       # libsndfile: int -> float by 2^-(bits-1)
    p = str(tmp_path / "x.wav")
    _write_pcm(p, data, 22050, width)
    got, sr = _read(p)
    ref = f[:, 0].copy()
    for c in range(1, ch):
        ref = (ref + f[:, c]).astype(np.float32)
    if ch > 1:
        ref = (ref * np.float32(1.0 / ch)).astype(np.float32)             # np.mean over channels in float32 (librosa.to_mono)
    assert sr == 22050 and got.shape == (n,)
    assert np.array_equal(got, ref)


def test_wav_float_and_chunk_walk(tmp_path):
    rng = np.random.default_rng(0)
    data = rng.standard_normal((1234, 2)).astype(np.float32)
    p = str(tmp_path / "f.wav")
    _write_float(p, data, 16000)
    got, sr = _read(p)
    assert sr == 16000
    assert np.array_equal(got, ((data[:, 0] + data[:, 1]).astype(np.float32) * np.float32(0.5)))
    L = pkg("_lib")
    sr_, ch_, fr_ = C.c_int(), C.c_int(), C.c_longlong()
    bits, isf = C.c_int(), C.c_int()
    L.check(L.lib().av_wav_info(p.encode(), C.byref(sr_), C.byref(ch_), C.byref(fr_), C.byref(bits), C.byref(isf)))
    assert (sr_.value, ch_.value, fr_.value, bits.value, isf.value) == (16000, 2, 1234, 32, 1)
    out = np.empty(10, np.float32)                                          # a window of the file
    L.check(L.lib().av_wav_read_mono_f32(p.encode(), 100, 10, out.ctypes.data))
    assert np.array_equal(out, got[100:110])


def test_wav_errors_are_reported_not_fatal(tmp_path):
    L = pkg("_lib")
    bad = tmp_path / "bad.wav"
    bad.write_bytes(b"not a wave file at all")
    sr_, ch_, fr_ = C.c_int(), C.c_int(), C.c_longlong()
    assert L.lib().av_wav_info(str(bad).encode(), C.byref(sr_), C.byref(ch_), C.byref(fr_), None, None) != 0
    assert b"RIFF" in L.lib().av_last_error()
    assert L.lib().av_wav_info(str(tmp_path / "missing.wav").encode(), C.byref(sr_), C.byref(ch_), C.byref(fr_), None, None) != 0
    p = str(tmp_path / "s.wav")
    _write_pcm(p, np.zeros((10, 1), np.int64), 16000, 2)
    out = np.empty(20, np.float32)
    assert L.lib().av_wav_read_mono_f32(p.encode(), 0, 20, out.ctypes.data) != 0      # more frames than the file holds


def test_oracle_resampler_known_answers():
    x = np.random.default_rng(1).standard_normal(4000).astype(np.float32)
    assert np.array_equal(PO.resample_sinc(x, 16000, 16000), x)                        # identity
    for sr_in in (48000, 44100, 8000):
        n = sr_in // 4
        t = np.arange(n) / sr_in
        tone = np.sin(2 * np.pi * 1000.0 * t).astype(np.float32)                       # 1 kHz: inside both pass bands
        y = PO.resample_sinc(tone, sr_in, 16000)
        assert len(y) == int(np.ceil(n * 16000 / sr_in))
        ref = np.sin(2 * np.pi * 1000.0 * np.arange(len(y)) / 16000.0)
        assert np.abs(y[200:-200] - ref[200:-200]).max() < 5e-3                               # pass-band gain of this filter: 1 +- 0.3 %
    t = np.arange(12000) / 48000.0                                                     # 10 kHz is above the new Nyquist: removed
    y = PO.resample_sinc(np.sin(2 * np.pi * 10000.0 * t).astype(np.float32), 48000, 16000)
    assert np.abs(y[200:-200]).max() < 2e-3


def test_pairing_and_retry_law():
    ds = pkg("dataset.multi_speaker_dataset")

    def sent(spk, i):
        return {"audio_path": f"a_{spk}.wav", "start_time": 0.0, "end_time": 1.0, "lip_path": f"lip_{spk}_{i}.npy",
                "text_path": f"/x/lip_J_1_M_03_{spk}_A_012_sentence_{i}.txt"}
    sents = [sent("C001", 1), sent("C001", 2), sent("C002", 3), sent("C003", 4)]
    assert ds.RandomSentencePairDataset.get_speaker_id(sents[0]["text_path"]) == "lip_J_1_M_03_C001_A"
    d = ds.RandomSentencePairDataset(sents, tokenizer=None, num_pairs_per_epoch=7, device="cpu")
    assert len(d) == 7
    seen = []
    d.load_pair = lambda s1, s2: seen.append((s1, s2)) or {"ok": True}
    random.seed(3)
    for _ in range(20):
        assert d[0] == {"ok": True}
    assert all(ds._speaker_id(a["text_path"]) != ds._speaker_id(b["text_path"]) for a, b in seen)     # same-speaker pairs are never loaded
    random.seed(3)                                                                     # the draws are the reference's: random.sample(list, 2) per attempt
    exp = []
    while len(exp) < 20:
        s1, s2 = random.sample(sents, 2)
        if ds._speaker_id(s1["text_path"]) != ds._speaker_id(s2["text_path"]):
            exp.append((s1, s2))
    assert seen == exp
    # failures are retried up to 10 times, then RuntimeError (:102-114)
    calls = []
    def boom(s1, s2):
        calls.append(1); raise ValueError("broken file")
    d.load_pair = boom
    with pytest.raises(RuntimeError):
        d[0]
    assert 1 <= len(calls) <= 10
    # fixed pairs: a same-speaker pair moves on to the next index (:128-141)
    pairs = [(sents[0], sents[1]), (sents[0], sents[2])]
    fd = ds.FixedSentencePairDataset(pairs, tokenizer=None, device="cpu")
    got = []
    fd.load_pair = lambda s1, s2: got.append((s1, s2)) or 1
    assert fd[0] == 1 and got == [pairs[1]] and len(fd) == 2
