"""SURVEY §8(f)-2 on the device: resampling kernel vs the oracle, and the dataset classes end to end on synthetic wav / npy / txt files
against the numpy restatement of ``load_pair`` (oracle/pipeline_oracle.py) - decode once, slice / mix / resize on the device."""
import os
import wave

import numpy as np
import pytest
import torch

from conftest import pkg
from oracle import pipeline_oracle as PO

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("sr_in,n", [(48000, 50001), (44100, 30000), (8000, 9000), (22050, 1)])
def test_resample_kernel_vs_oracle(sr_in, n):
    ds = pkg("dataset.multi_speaker_dataset")
    x = np.random.default_rng(sr_in).standard_normal(n).astype(np.float32)
    st = ds.AudioStore("cuda")
    y = st.resample(torch.from_numpy(x).cuda(), sr_in).cpu().numpy()
    ref = PO.resample_sinc(x, sr_in, 16000)
    assert y.shape == ref.shape
    assert np.abs(y - ref).max() <= 2e-6 * max(1.0, np.abs(ref).max())


def _wav16(path, x, sr, ch=1):
    d = np.clip(np.round(x * 32767.0), -32768, 32767).astype("<i2")
    if ch == 2:
        d = np.stack([d, d[::-1]], 1)
    with wave.open(path, "wb") as w:
        w.setnchannels(ch); w.setsampwidth(2); w.setframerate(sr)
        w.writeframes(d.tobytes())
    f = d.astype(np.float32) / np.float32(32768.0)
    return f if ch == 1 else ((f[:, 0] + f[:, 1]).astype(np.float32) * np.float32(0.5))


class _Tok:
    def encode(self, s):
        return [4 + (ord(c) % 700) for c in s]


def _make_corpus(tmp, sr_b=16000):
    rng = np.random.default_rng(7)
    wavs = {"A": _wav16(str(tmp / "rec_A.wav"), rng.uniform(-0.5, 0.5, 16000 * 6), 16000),
            "B": _wav16(str(tmp / "rec_B.wav"), rng.uniform(-0.5, 0.5, sr_b * 5), sr_b, ch=2)}
    sents = []
    for spk, i, t0, t1, T in (("A", 1, 0.5, 2.0, 37), ("A", 2, 2.25, 5.125, 50), ("B", 3, 0.0, 1.7, 42), ("B", 4, 3.0, 9.0, 25)):
        lip = rng.integers(0, 256, (T, 128, 128, 3), dtype=np.uint8)
        lp = str(tmp / f"lip_J_1_M_03_C{spk}_A_012_sentence_{i}.npy")
        np.save(lp, lip)
        tp = str(tmp / f"lip_J_1_M_03_C{spk}_A_012_sentence_{i}.txt")
        with open(tp, "w", encoding="utf-8") as f:
            f.write(f"  문장 {i} 입니다 \n")
        sents.append({"audio_path": str(tmp / f"rec_{spk}.wav"), "start_time": t0, "end_time": t1, "lip_path": lp, "text_path": tp, "_spk": spk})
    return wavs, sents


def _oracle_pair(wavs, sr_of, s1, s2, tok):
    def clip(s):
        a = PO.resample_sinc(wavs[s["_spk"]], sr_of[s["_spk"]], 16000)
        return a[int(s["start_time"] * 16000):int(s["end_time"] * 16000)]
    mixed, m1, m2 = PO.mix_pair(clip(s1), clip(s2))
    l1, l2 = PO.lips(np.load(s1["lip_path"])), PO.lips(np.load(s2["lip_path"]))
    lab = [np.array(tok.encode(open(s["text_path"], encoding="utf-8").read().strip()), dtype=np.int64) for s in (s1, s2)]
    return mixed, m1, m2, l1, l2, lab


@pytest.mark.parametrize("sr_b", [16000, 48000])
def test_load_pair_vs_oracle_and_caches(tmp_path, sr_b):
    ds = pkg("dataset.multi_speaker_dataset"); cf = pkg("dataset.collate_fn").collate_fn
    wavs, sents = _make_corpus(tmp_path, sr_b)
    sr_of = {"A": 16000, "B": sr_b}
    tok = _Tok()
    d = ds.MultiSpeakerDataset(sents, tok)
    items = []
    for s1, s2 in ((sents[0], sents[2]), (sents[1], sents[3]), (sents[3], sents[0])):
        it = d.load_pair(s1, s2)
        mixed, m1, m2, l1, l2, lab = _oracle_pair(wavs, sr_of, s1, s2, tok)
        assert set(it.keys()) == {"audio", "mask1", "mask2", "lip1", "label1", "lip1_len", "lip2", "label2", "lip2_len"}      # :70-84
        a = it["audio"].cpu().numpy()
        if sr_b == 16000:
            assert np.array_equal(a, mixed)                                 # no resampling involved: bit-exact
        else:
            assert a.shape == mixed.shape and np.abs(a - mixed).max() < 1e-5
        assert np.array_equal(it["mask1"].cpu().numpy(), m1) and np.array_equal(it["mask2"].cpu().numpy(), m2)
        assert np.array_equal(it["lip1"].cpu().numpy(), l1) and np.array_equal(it["lip2"].cpu().numpy(), l2)
        assert it["lip1"].shape[1:] == (1, 96, 96) and it["lip1_len"] == l1.shape[0] and it["lip2_len"] == l2.shape[0]
        assert np.array_equal(it["label1"], lab[0]) and np.array_equal(it["label2"], lab[1])
        items.append(it)
    assert d.audio.decoded_files == 2                                       # each recording decoded once for six clips
    assert d.lips.misses == 4 and d.lips.hits == 2                          # sentence 0 and 3 came back from the device cache
    batch = cf(items)                                                       # the reference's collate contract on device-resident items
    assert batch["audio"].is_cuda and batch["lip1"].is_cuda and batch["audio"].shape[0] == 3
    assert (batch["mask1"][0, items[0]["mask1"].numel():] == 3).all()
    # a clip past the end of the recording is cut like a numpy slice (:16): sentence 3 of speaker B ends at 9 s of a 5 s file
    n_b = int(np.ceil(len(wavs["B"]) * 16000 / sr_b))
    assert items[1]["audio"].numel() == max(int(5.125 * 16000) - int(2.25 * 16000), n_b - 3 * 16000)


def test_dataset_iteration_and_errors(tmp_path):
    import random
    ds = pkg("dataset.multi_speaker_dataset")
    _, sents = _make_corpus(tmp_path)
    d = ds.RandomSentencePairDataset(sents, _Tok(), num_pairs_per_epoch=5)
    random.seed(0)
    for i in range(len(d)):
        it = d[i]
        assert it["audio"].is_cuda and it["audio"].abs().max() <= 1.0
    assert d.audio.decoded_files == 2
    np.save(sents[0]["lip_path"], np.zeros((0, 128, 128, 3), np.uint8))    # empty clip -> RuntimeError, as :59-60
    with pytest.raises(RuntimeError):
        d.load_pair(sents[0], sents[2])
    fd = ds.FixedSentencePairDataset([(sents[1], sents[2])], _Tok())
    assert fd[0]["lip1_len"] == 50
    bad = dict(sents[2]); bad["audio_path"] = str(tmp_path / "nope.mp3")
    with pytest.raises((RuntimeError, FileNotFoundError)):
        d.load_pair(sents[1], bad)


@pytest.mark.parametrize("bs,precision", [(2, "fp32"), (1, "bf16"), (3, "bf16")])
def test_dataset_feeds_the_training_step(tmp_path, bs, precision):
    """files -> FixedSentencePairDataset -> torch DataLoader (collate_fn, no workers) -> MultimodalTrainer.train_step: the callers on the input
    side of the hot path and the path itself in one loop, as main.py:88-129 wires them (tiny wav2vec2 configuration, fp32 mode)."""
    from test_step_gpu import build as build_trainer
    ds = pkg("dataset.multi_speaker_dataset"); cf = pkg("dataset.collate_fn").collate_fn; init = pkg("utils.init")
    _, sents = _make_corpus(tmp_path)
    pairs = [(sents[0], sents[2]), (sents[1], sents[3]), (sents[2], sents[1]), (sents[3], sents[0])]
    loader = torch.utils.data.DataLoader(ds.FixedSentencePairDataset(pairs, _Tok()), batch_size=bs, shuffle=False, collate_fn=cf, num_workers=0)
    t = build_trainer(init.W2V2_TINY, precision)
    for m in (t.visual_encoder, t.audio_encoder, t.fusion_module, t.decoder1):
        m.train()
    before = t.decoder1.net[0].weight.detach().clone()
    losses = []
    for batch in loader:
        assert batch["audio"].is_cuda and set(batch.keys()) >= {"lip1", "lip2", "text1", "text2", "audio", "mask1", "mask2", "audio_lengths"}
        out = t.train_step(batch)
        losses.append(float(out["total"].detach()))
    assert len(losses) == (4 + bs - 1) // bs and all(np.isfinite(l) and l > 0 for l in losses)
    assert not torch.equal(before, t.decoder1.net[0].weight.detach())
    ev_loss, wer = t.evaluate(loader)                                     # main.py:168 on the same loader: greedy decode + WER on the device
    assert np.isfinite(ev_loss) and 0.0 <= wer
    assert np.isfinite(t.train_epoch(loader))                             # main.py:165
