"""CPU: local HF wav2vec2 checkpoint interop (model/encoder.py:83 of the reference loads a HF directory by name; offline only
LOCAL directories exist).  Key remapping as ``from_pretrained`` does it, strictness, and a round trip through the installed
``transformers`` library's own ``save_pretrained``."""
import json
import os

import pytest
import torch

from conftest import pkg


def _hf_config_json(cfg, **extra):
    c = {k: (list(v) if isinstance(v, tuple) else v) for k, v in cfg.items()}
    c.update(feat_extract_norm="layer", do_stable_layer_norm=True, model_type="wav2vec2", **extra)
    c.setdefault("conv_bias", True)
    return c


def _write_dir(path, cfg, sd, fname="model.safetensors"):
    os.makedirs(path, exist_ok=True)
    with open(os.path.join(path, "config.json"), "w") as f:
        json.dump(_hf_config_json(cfg), f)
    if sd is None:
        return
    if fname.endswith(".safetensors"):
        from safetensors.torch import save_file
        save_file({k: v.contiguous() for k, v in sd.items()}, os.path.join(path, fname))
    else:
        torch.save(dict(sd), os.path.join(path, fname))


@pytest.mark.parametrize("form", ["plain", "ctc_prefixed", "legacy_weight_norm", "bin"])
def test_local_checkpoint_key_forms(tmp_path, form):
    init = pkg("utils.init"); enc = pkg("model.encoder")
    cfg = init.W2V2_TINY
    sd = init.w2v2_state_dict(cfg, seed=11, prefix="")
    disk = dict(sd)
    if form == "ctc_prefixed":                       # Wav2Vec2ForCTC fine-tune (e.g. kresnik/wav2vec2-large-xlsr-korean): prefix + head
        disk = {"wav2vec2." + k: v for k, v in sd.items()}
        disk["lm_head.weight"] = torch.zeros(32, cfg["hidden_size"]); disk["lm_head.bias"] = torch.zeros(32)
    if form == "legacy_weight_norm":
        p = "encoder.pos_conv_embed.conv."
        disk[p + "weight_g"] = disk.pop(p + "parametrizations.weight.original0")
        disk[p + "weight_v"] = disk.pop(p + "parametrizations.weight.original1")
    d = str(tmp_path / form)
    _write_dir(d, cfg, disk, "pytorch_model.bin" if form == "bin" else "model.safetensors")
    ae = enc.AudioEncoder(d, freeze=True)
    got = ae.model.state_dict()
    assert sorted(got) == sorted(sd)
    for k in sd:
        assert torch.equal(got[k], sd[k]), k
    assert ae.output_dim == cfg["hidden_size"] and not any(p.requires_grad for p in ae.parameters())
    # config.json carries no regularisation keys here: HF defaults apply (a real checkpoint trains with dropout / LayerDrop / SpecAugment)
    assert ae.model.cfg["layerdrop"] == 0.1 and ae.model.cfg["mask_time_prob"] == 0.05


def test_missing_weights_raise_instead_of_random_init(tmp_path):
    init = pkg("utils.init"); enc = pkg("model.encoder")
    cfg = init.W2V2_TINY
    d = str(tmp_path / "noweights")
    _write_dir(d, cfg, None)
    with pytest.raises(FileNotFoundError):
        enc.AudioEncoder(d)
    ae = enc.AudioEncoder(d, random_init=True)           # explicit opt-in
    assert len(ae.model.state_dict()) == len(init.w2v2_state_dict(cfg, prefix=""))
    with pytest.raises(FileNotFoundError):
        enc.AudioEncoder("kresnik/wav2vec2-large-xlsr-korean")     # a model NAME would need the network


def test_mismatched_checkpoint_raises(tmp_path):
    init = pkg("utils.init"); enc = pkg("model.encoder")
    cfg = init.W2V2_TINY
    sd = init.w2v2_state_dict(cfg, seed=11, prefix="")
    bad = dict(sd); bad.pop("encoder.layers.3.attention.q_proj.weight")
    d = str(tmp_path / "bad")
    _write_dir(d, cfg, bad)
    with pytest.raises(KeyError):
        enc.AudioEncoder(d)
    extra = dict(sd); extra["encoder.layers.99.layer_norm.weight"] = torch.zeros(cfg["hidden_size"])
    d2 = str(tmp_path / "extra")
    _write_dir(d2, cfg, extra)
    with pytest.raises(KeyError):
        enc.AudioEncoder(d2)


@pytest.mark.parametrize("conv_bias", [True, False])
def test_round_trip_through_transformers_save_pretrained(tmp_path, conv_bias):
    """A directory written by the installed HF library itself (Wav2Vec2ForCTC.save_pretrained: ``wav2vec2.`` prefix, ``lm_head``,
    the library's current weight-norm naming) loads, and every tensor equals the HF module's."""
    tf = pytest.importorskip("transformers")
    init = pkg("utils.init"); enc = pkg("model.encoder")
    cfg = init.W2V2_TINY
    hc = tf.Wav2Vec2Config(**_hf_config_json(cfg, conv_bias=conv_bias), vocab_size=32, hidden_dropout=0.07, layerdrop=0.03, mask_time_prob=0.04)
    torch.manual_seed(0)
    hf = tf.Wav2Vec2ForCTC(hc)
    d = str(tmp_path / "hf")
    hf.save_pretrained(d, safe_serialization=True)
    ae = enc.AudioEncoder(d)
    want = hf.wav2vec2.state_dict()
    got = ae.model.state_dict()
    if not conv_bias:                                    # the kernels always add a conv bias: zeros stand in for the absent tensors
        for i in range(7):
            assert float(got.pop(f"feature_extractor.conv_layers.{i}.conv.bias").abs().max()) == 0
    assert sorted(got) == sorted(want)
    for k in want:
        assert torch.equal(got[k], want[k]), k
    assert ae.model.cfg["hidden_dropout"] == 0.07 and ae.model.cfg["layerdrop"] == 0.03 and ae.model.cfg["mask_time_prob"] == 0.04
