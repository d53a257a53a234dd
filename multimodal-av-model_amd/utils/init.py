"""Seeded random initialisation of every module's state_dict (CPU generator => identical on every box).

There are no pretrained weights offline, so parity and benchmarks run on seeded random weights.
Key names and shapes are the reference's checkpoint contract: visual 129 keys
(model/encoder.py:6-75), audio 422 keys = ``model.`` + HF wav2vec2 names (model/encoder.py:83), fusion 30 keys
(model/fusion_module.py:9-27), decoder ``net.0.*`` (model/decoder.py:9-11).
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Dict, Tuple

import torch

W2V2_LARGE = dict(
    hidden_size=1024, num_hidden_layers=24, num_attention_heads=16, intermediate_size=4096,
    conv_dim=(512,) * 7, conv_kernel=(10, 3, 3, 3, 3, 2, 2), conv_stride=(5, 2, 2, 2, 2, 2, 2),
    num_conv_pos_embeddings=128, num_conv_pos_embedding_groups=16, layer_norm_eps=1e-5,
)
W2V2_TINY = dict(
    hidden_size=64, num_hidden_layers=12, num_attention_heads=4, intermediate_size=128,
    conv_dim=(32,) * 7, conv_kernel=(10, 3, 3, 3, 3, 2, 2), conv_stride=(5, 2, 2, 2, 2, 2, 2),
    num_conv_pos_embeddings=16, num_conv_pos_embedding_groups=4, layer_norm_eps=1e-5,
)


class _Gen:
    def __init__(self, seed: int):
        self.g = torch.Generator(device="cpu")
        self.g.manual_seed(seed)

    def normal(self, shape, std=1.0, mean=0.0):
        return torch.randn(shape, generator=self.g, dtype=torch.float32) * std + mean

    def uniform(self, shape, lo, hi):
        return torch.rand(shape, generator=self.g, dtype=torch.float32) * (hi - lo) + lo


def _linear(g: _Gen, sd, name, out_f, in_f, gain=1.0):
    sd[name + ".weight"] = g.normal((out_f, in_f), gain / math.sqrt(in_f))
    sd[name + ".bias"] = g.normal((out_f,), 0.02)


def _norm(g: _Gen, sd, name, c):
    sd[name + ".weight"] = g.normal((c,), 0.1, 1.0)
    sd[name + ".bias"] = g.normal((c,), 0.05)


def _bn(g: _Gen, sd, name, c):
    _norm(g, sd, name, c)
    sd[name + ".running_mean"] = g.normal((c,), 0.1)
    sd[name + ".running_var"] = g.uniform((c,), 0.8, 1.2)
    sd[name + ".num_batches_tracked"] = torch.zeros((), dtype=torch.long)


def visual_state_dict(seed: int = 1) -> "OrderedDict[str, torch.Tensor]":
    g = _Gen(seed)
    sd = OrderedDict()
    sd["frontend3D.0.weight"] = g.normal((64, 1, 5, 7, 7), math.sqrt(2.0 / 245))
    _bn(g, sd, "frontend3D.1", 64)
    sd["frontend3D.2.weight"] = g.normal((64,), 0.05, 0.25)
    inpl = 64
    for li, planes in ((1, 64), (2, 128), (3, 256), (4, 512)):
        for bi in range(2):
            p = f"trunk.layer{li}.{bi}"
            cin = inpl if bi == 0 else planes
            sd[p + ".conv1.weight"] = g.normal((planes, cin, 3, 3), math.sqrt(2.0 / (9 * cin)))
            _bn(g, sd, p + ".bn1", planes)
            sd[p + ".relu.weight"] = g.normal((planes,), 0.05, 0.25)
            sd[p + ".conv2.weight"] = g.normal((planes, planes, 3, 3), math.sqrt(1.0 / (9 * planes)))
            _bn(g, sd, p + ".bn2", planes)
            if bi == 0 and (li != 1):
                sd[p + ".downsample.0.weight"] = g.normal((planes, cin, 1, 1), math.sqrt(1.0 / cin))
                _bn(g, sd, p + ".downsample.1", planes)
        inpl = planes
    return sd


def w2v2_state_dict(cfg: dict, seed: int = 2, prefix: str = "model.") -> "OrderedDict[str, torch.Tensor]":
    g = _Gen(seed)
    sd = OrderedDict()
    H, I = cfg["hidden_size"], cfg["intermediate_size"]
    sd["masked_spec_embed"] = g.uniform((H,), 0.0, 1.0)
    cin = 1
    for i, (c, k) in enumerate(zip(cfg["conv_dim"], cfg["conv_kernel"])):
        p = f"feature_extractor.conv_layers.{i}"
        sd[p + ".conv.weight"] = g.normal((c, cin, k), math.sqrt(2.0 / (cin * k)))
        sd[p + ".conv.bias"] = g.normal((c,), 0.02)
        _norm(g, sd, p + ".layer_norm", c)
        cin = c
    _norm(g, sd, "feature_projection.layer_norm", cin)
    _linear(g, sd, "feature_projection.projection", H, cin)
    kp, gp = cfg["num_conv_pos_embeddings"], cfg["num_conv_pos_embedding_groups"]
    sd["encoder.pos_conv_embed.conv.bias"] = g.normal((H,), 0.02)
    sd["encoder.pos_conv_embed.conv.parametrizations.weight.original0"] = g.uniform((1, 1, kp), 0.5, 1.5)
    sd["encoder.pos_conv_embed.conv.parametrizations.weight.original1"] = g.normal((H, H // gp, kp), 0.05)
    _norm(g, sd, "encoder.layer_norm", H)
    for li in range(cfg["num_hidden_layers"]):
        p = f"encoder.layers.{li}"
        for nm in ("k_proj", "v_proj", "q_proj", "out_proj"):
            _linear(g, sd, f"{p}.attention.{nm}", H, H, 0.7 if nm == "out_proj" else 1.0)
        _norm(g, sd, p + ".layer_norm", H)
        _linear(g, sd, p + ".feed_forward.intermediate_dense", I, H)
        _linear(g, sd, p + ".feed_forward.output_dense", H, I, 0.7)
        _norm(g, sd, p + ".final_layer_norm", H)
    return OrderedDict((prefix + k, v) for k, v in sd.items())


def fusion_state_dict(visual_dim: int, audio_dim: int, fused_dim: int, seed: int = 3) -> "OrderedDict[str, torch.Tensor]":
    g = _Gen(seed)
    sd = OrderedDict()
    E = fused_dim
    _linear(g, sd, "visual_proj", E, visual_dim)
    _linear(g, sd, "audio_proj", E, audio_dim)
    for nm in ("cross_attn_visual", "cross_attn_audio"):       # cross_attn_visual is declared but never called
        sd[nm + ".in_proj_weight"] = g.normal((3 * E, E), 1.0 / math.sqrt(E))
        sd[nm + ".in_proj_bias"] = g.normal((3 * E,), 0.02)
        _linear(g, sd, nm + ".out_proj", E, E)
    _linear(g, sd, "fusion_proj", E, E)
    for layer in (0, 1):
        for sfx in ("", "_reverse"):
            in_f = E if layer == 0 else 2 * E
            k = 1.0 / math.sqrt(E)
            sd[f"temporal_model.weight_ih_l{layer}{sfx}"] = g.uniform((4 * E, in_f), -k, k)
            sd[f"temporal_model.weight_hh_l{layer}{sfx}"] = g.uniform((4 * E, E), -k, k)
            sd[f"temporal_model.bias_ih_l{layer}{sfx}"] = g.uniform((4 * E,), -k, k)
            sd[f"temporal_model.bias_hh_l{layer}{sfx}"] = g.uniform((4 * E,), -k, k)
    return sd


def decoder_state_dict(input_dim: int, vocab: int, seed: int = 4) -> "OrderedDict[str, torch.Tensor]":
    g = _Gen(seed)
    sd = OrderedDict()
    _linear(g, sd, "net.0", vocab, input_dim, 8.0)   # wide logits: argmax decodes are not near-ties
    return sd


def projection_params(d: int, out: int = 128, seed: int = 5) -> Tuple[torch.Tensor, torch.Tensor]:
    """Weights of the untracked nn.Linear(D,128) the trainer creates lazily (model/trainer.py:105-106)."""
    g = _Gen(seed)
    k = 1.0 / math.sqrt(d)
    return g.uniform((out, d), -k, k), g.uniform((out,), -k, k)


def legacy_state_dict(vocab_size: int = 200, hidden_dim: int = 256, seed: int = 7, in_channels: int = 3, mel_dim: int = 80
                      ) -> "OrderedDict[str, torch.Tensor]":
    """Seeded state_dict of the legacy mel + GRU model (key names of 이전 버전/multimodal_ctc_korean.py:8-55: lip_encoder.cnn.{0,3}.*,
    lip_encoder.rnn.*, audio_encoder.rnn.*, fc.*)."""
    g = _Gen(seed)
    sd = OrderedDict()
    sd["lip_encoder.cnn.0.weight"] = g.normal((32, in_channels, 3, 3), math.sqrt(2.0 / (9 * in_channels)))
    sd["lip_encoder.cnn.0.bias"] = g.normal((32,), 0.05)
    sd["lip_encoder.cnn.3.weight"] = g.normal((64, 32, 3, 3), math.sqrt(2.0 / (9 * 32)))
    sd["lip_encoder.cnn.3.bias"] = g.normal((64,), 0.05)
    H = hidden_dim
    for pre, in0 in (("lip_encoder.rnn.", 64 * 24 * 24), ("audio_encoder.rnn.", mel_dim)):
        for layer in range(2):
            in_f = in0 if layer == 0 else 2 * H
            for suf in ("", "_reverse"):
                sd[f"{pre}weight_ih_l{layer}{suf}"] = g.normal((3 * H, in_f), 1.0 / math.sqrt(in_f))
                sd[f"{pre}weight_hh_l{layer}{suf}"] = g.normal((3 * H, H), 1.0 / math.sqrt(H))
                sd[f"{pre}bias_ih_l{layer}{suf}"] = g.normal((3 * H,), 0.05)
                sd[f"{pre}bias_hh_l{layer}{suf}"] = g.normal((3 * H,), 0.05)
    sd["fc.weight"] = g.normal((vocab_size, 4 * H), 1.0 / math.sqrt(4 * H))
    sd["fc.bias"] = g.normal((vocab_size,), 0.02)
    return sd


def legacy_batch(batch: int, steps: int, vocab_size: int, seed: int = 11, label_len: int = None, size: int = 96, mel_dim: int = 80):
    """Synthetic batch in the layout of the legacy collate_fn (이전 버전/train_ctc_korea.py:54-77): frames U[0,1) [B,T,3,96,96] (:30),
    mel >= 0 [B,T,80], labels time-major (L, B) without blanks or repeats, lengths."""
    g = _Gen(seed)
    L_ = label_len or batch
    fa = g.uniform((batch, steps, 3, size, size), 0.0, 1.0)
    fb = g.uniform((batch, steps, 3, size, size), 0.0, 1.0)
    mel = g.normal((batch, steps, mel_dim), 1.0).abs()
    mel_len = torch.full((batch,), steps, dtype=torch.long)

    def labels():
        lab = torch.zeros((L_, batch), dtype=torch.long)
        for b in range(batch):
            perm = torch.randperm(vocab_size - 1, generator=g.g)[:L_] + 1        # distinct non-blank ids
            lab[:, b] = perm
        lens = torch.randint(1, min(L_, max(1, steps // 2)) + 1, (L_ if False else batch,), generator=g.g)
        return lab, lens

    la, na = labels()
    lb, nb = labels()
    return fa, fb, mel, mel_len, la, na, lb, nb
