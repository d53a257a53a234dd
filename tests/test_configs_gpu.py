"""Parity at the other BASELINE.json configurations.

* config 3 (long-form, 15 s clips: T_enc = 749, 375 lip frames): the HIP path against the CPU oracle on a wav2vec2-large
  SHAPED model cut to 10 layers (so that the oracle finishes in seconds) — exercises the LDS-tiled attention over
  12 key tiles, the 375-step BiLSTM and the resampling 749 -> 375.
* config 2 size (batch 32 x 4 s, full wav2vec2-large): size-independent properties — bit-exact batch-permutation
  equivariance of the forward, and fp32-vs-bf16 agreement of the losses.
"""
import pytest
import torch

from conftest import pkg

pytestmark = pytest.mark.gpu


def _modules(cfg, precision):
    init = pkg("utils.init"); enc = pkg("model.encoder"); fm = pkg("model.fusion_module"); dm = pkg("model.decoder")
    pkg("precision").set_precision(precision)
    ae = enc.AudioEncoder(dict(cfg), freeze=True).cuda(); ae.load_state_dict(init.w2v2_state_dict(cfg))
    fu = fm.CrossAttentionFusion(512, cfg["hidden_size"], 512).cuda(); fu.load_state_dict(init.fusion_state_dict(512, cfg["hidden_size"], 512))
    de = dm.CTCDecoder(1024, 800, 3).cuda(); de.load_state_dict(init.decoder_state_dict(1024, 800))
    return ae, fu, de


def test_longform_15s_vs_oracle_fp32():
    from oracle import av_oracle as O
    init = pkg("utils.init"); synth = pkg("dataset.synthetic"); L = pkg("_lib"); ops = pkg("ops")
    cfg = dict(init.W2V2_LARGE, num_hidden_layers=10)
    batch = synth.make_batch(2, 15.0, seed=7, ragged=True)            # items of 15 s and 11.25 s
    assert batch["audio"].shape[1] == 240000 and batch["lip1"].shape[1] == 375
    ae, fu, de = _modules(cfg, "fp32")
    ae.eval(); fu.eval(); de.eval()
    mask = batch["mask1"]
    with torch.no_grad():
        sd_a, sd_f, sd_d = init.w2v2_state_dict(cfg), init.fusion_state_dict(512, 1024, 512), init.decoder_state_dict(1024, 800)
        r_last, r_mid = O.audio_forward(sd_a, cfg, batch["audio"], mask != 3)
        T_enc = r_last.shape[1]
        assert T_enc == 749
        m_ds = O.downsample_mask(mask, T_enc)
        g = torch.Generator().manual_seed(1)
        vis = torch.randn(2, 375, 512, generator=g)
        r_f, r_il = O.fusion_forward(sd_f, vis, r_last, m_ds)
        r_lp = O.decoder_forward(sd_d, r_f)
        last, mid = ae(batch["audio"].cuda(), attention_mask=(mask != 3).cuda())
        md = torch.empty((2, T_enc), dtype=torch.long, device="cuda")
        L.check(L.lib().av_mask_downsample(ops.ptr(mask.cuda().contiguous()), ops.ptr(md), 2, mask.shape[1], T_enc, ops.stream()))
        assert torch.equal(md.cpu(), m_ds)
        f, il = fu(vis.cuda(), last, md)
        lp = de(f)
    assert float((last.cpu() - r_last).abs().max()) < 1e-3 and float((mid.cpu() - r_mid).abs().max()) < 1e-3
    assert torch.equal(il.cpu(), r_il)
    assert float((f.cpu() - r_f).abs().max()) < 1e-3
    assert float((lp.cpu() - r_lp).abs().max()) < 1e-3          # BASELINE gate: 1e-3 fp32


def test_c2_size_permutation_equivariance_and_modes():
    init = pkg("utils.init"); synth = pkg("dataset.synthetic")
    cfg = init.W2V2_LARGE
    B = 32
    batch = synth.make_batch(B, 4.0, seed=3, ragged=True)
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(2))
    outs = {}
    for prec in ("bf16", "fp32"):
        ae, fu, de = _modules(cfg, prec)
        ae.eval(); fu.eval(); de.eval()
        with torch.no_grad():
            wav, m = batch["audio"].cuda(), (batch["mask1"] != 3).cuda()
            last, mid = ae(wav, attention_mask=m)
            last_p, mid_p = ae(wav[perm.cuda()].contiguous(), attention_mask=m[perm.cuda()].contiguous())
        # every kernel on the path is row/item-wise: permuting the batch permutes the result bit for bit
        assert torch.equal(last[perm.cuda()], last_p) and torch.equal(mid[perm.cuda()], mid_p), prec
        outs[prec] = last.float().cpu()
        del ae, fu, de
        torch.cuda.empty_cache()
    err = float((outs["bf16"] - outs["fp32"]).abs().max())
    print("C2-size audio features: max |bf16 - fp32| =", err, "feature scale", float(outs["fp32"].abs().max()))
    assert err < 0.25
