"""TEST INFRASTRUCTURE — plain-PyTorch fp32 CPU restatement of the reference audio-visual CTC path.

Every function works on a flat ``dict[str, Tensor]`` that uses the reference's state_dict key names, so the
same weights can be loaded into the reference modules (``tests/golden/make_golden.py``), into this restatement
and into the HIP product.  Citations are ``/root/reference`` file:line, ``hf:`` =
transformers/models/wav2vec2/modeling_wav2vec2.py (5.15.0), ``torch:`` = torch/nn (2.10.0).

Parity status: pinned by tests/golden/*.npz (generated from the imported reference, see oracle/__init__.py).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = Dict[str, Tensor]

# ----------------------------------------------------------------------------------------------------------
# wav2vec2 configuration (config-driven, SURVEY §8c: values are not hard-coded in the kernels)
# ----------------------------------------------------------------------------------------------------------
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


def sub(sd: SD, prefix: str) -> SD:
    """View of ``sd`` with ``prefix`` stripped (tensors are shared, so in-place updates propagate)."""
    n = len(prefix)
    return {k[n:]: v for k, v in sd.items() if k.startswith(prefix)}


# ----------------------------------------------------------------------------------------------------------
# a1/a2  VisualEncoder  (model/encoder.py:6-75)
# ----------------------------------------------------------------------------------------------------------
def _bn(sd: SD, p: str, x: Tensor, training: bool) -> Tensor:
    """nn.BatchNorm2d/3d: batch statistics + running-stat update in train mode, eps 1e-5, momentum .1."""
    if training and (p + "num_batches_tracked") in sd:
        sd[p + "num_batches_tracked"] += 1
    return F.batch_norm(x, sd[p + "running_mean"], sd[p + "running_var"], sd[p + "weight"], sd[p + "bias"],
                        training=training, momentum=0.1, eps=1e-5)


def _basic_block(sd: SD, p: str, x: Tensor, stride: int, training: bool) -> Tensor:
    """BasicBlock.forward, model/encoder.py:16-22 (one PReLU module shared by both activations)."""
    a = sd[p + "relu.weight"]
    out = F.conv2d(x, sd[p + "conv1.weight"], None, stride, 1)
    out = F.prelu(_bn(sd, p + "bn1.", out, training), a)
    out = F.conv2d(out, sd[p + "conv2.weight"], None, 1, 1)
    out = _bn(sd, p + "bn2.", out, training)
    if (p + "downsample.0.weight") in sd:
        idt = F.conv2d(x, sd[p + "downsample.0.weight"], None, stride, 0)
        idt = _bn(sd, p + "downsample.1.", idt, training)
    else:
        idt = x
    return F.prelu(out + idt, a)


def visual_forward(sd: SD, x: Tensor, training: bool) -> Tensor:
    """VisualEncoder.forward, model/encoder.py:69-75.  x [B,1,T,96,96] -> [B,T,512].
    In train mode the BN running statistics inside ``sd`` are updated in place (trainer.py:54)."""
    B = x.shape[0]
    y = F.conv3d(x, sd["frontend3D.0.weight"], None, (1, 2, 2), (2, 3, 3))
    y = _bn(sd, "frontend3D.1.", y, training)
    y = F.prelu(y, sd["frontend3D.2.weight"])
    y = F.max_pool3d(y, (1, 3, 3), (1, 2, 2), (0, 1, 1))
    y = y.transpose(1, 2).contiguous().view(B * y.shape[2], 64, y.shape[3], y.shape[4])
    for li, stride in ((1, 1), (2, 2), (3, 2), (4, 2)):
        y = _basic_block(sd, f"trunk.layer{li}.0.", y, stride, training)
        y = _basic_block(sd, f"trunk.layer{li}.1.", y, 1, training)
    y = y.mean(dim=(2, 3))                      # AdaptiveAvgPool2d(1).flatten(1), model/encoder.py:52
    return y.view(B, -1, 512)


# ----------------------------------------------------------------------------------------------------------
# a3-a11  AudioEncoder / Wav2Vec2Model (model/encoder.py:80-100, hf:275-802, 997-1036, 1319-1375)
# ----------------------------------------------------------------------------------------------------------
def feat_lengths(cfg: dict, lengths: Tensor) -> Tensor:
    """hf:997-1015 integer length law L <- floor((L-k)/s)+1 per conv layer."""
    for k, s in zip(cfg["conv_kernel"], cfg["conv_stride"]):
        lengths = torch.div(lengths - k, s, rounding_mode="floor") + 1
    return lengths


def pos_conv_weight(sd: SD) -> Tensor:
    """weight_norm(dim=2): w = v * g / ||v|| with the norm over dims (0,1) per tap (hf:355)."""
    g = sd["encoder.pos_conv_embed.conv.parametrizations.weight.original0"]
    v = sd["encoder.pos_conv_embed.conv.parametrizations.weight.original1"]
    return v * (g / v.norm(2, dim=(0, 1), keepdim=True))


def w2v2_forward(sd: SD, cfg: dict, wav: Tensor, attention_mask: Optional[Tensor]) -> Tuple[Tensor, List[Tensor]]:
    """Wav2Vec2Model.forward in deterministic mode (dropout/LayerDrop/SpecAugment zeroed).
    Returns (last_hidden_state, hidden_states[0..L]) like hf:1319-1375 with output_hidden_states=True."""
    eps = cfg["layer_norm_eps"]
    h = wav[:, None]
    for i, (k, s) in enumerate(zip(cfg["conv_kernel"], cfg["conv_stride"])):          # hf:291-299
        p = f"feature_extractor.conv_layers.{i}."
        h = F.conv1d(h, sd[p + "conv.weight"], sd[p + "conv.bias"], stride=s)
        h = F.layer_norm(h.transpose(1, 2), (h.shape[1],), sd[p + "layer_norm.weight"], sd[p + "layer_norm.bias"], 1e-5)
        h = F.gelu(h.transpose(1, 2))
    feats = h.transpose(1, 2)                                                         # [B,T,512]
    B, T, _ = feats.shape
    keep = None
    if attention_mask is not None:                                                    # hf:1018-1036
        n = feat_lengths(cfg, attention_mask.long().sum(-1))
        keep = torch.arange(T)[None, :] < n[:, None]
    h = F.layer_norm(feats, (feats.shape[-1],), sd["feature_projection.layer_norm.weight"],
                     sd["feature_projection.layer_norm.bias"], eps)                   # hf:429-434
    h = F.linear(h, sd["feature_projection.projection.weight"], sd["feature_projection.projection.bias"])
    if keep is not None:
        h = h * keep[..., None].to(h.dtype)                                           # hf:752-755
    # positional conv, hf:360-368: grouped conv, drop last (even kernel), GELU, residual
    kpos = cfg["num_conv_pos_embeddings"]
    pc = F.conv1d(h.transpose(1, 2), pos_conv_weight(sd), sd["encoder.pos_conv_embed.conv.bias"],
                  padding=kpos // 2, groups=cfg["num_conv_pos_embedding_groups"])
    if kpos % 2 == 0:
        pc = pc[:, :, :-1]
    h = h + F.gelu(pc).transpose(1, 2)
    nh = cfg["num_attention_heads"]
    hd = cfg["hidden_size"] // nh
    hidden_states = []
    for li in range(cfg["num_hidden_layers"]):                                        # hf:611-654
        hidden_states.append(h)
        p = f"encoder.layers.{li}."
        x = F.layer_norm(h, (h.shape[-1],), sd[p + "layer_norm.weight"], sd[p + "layer_norm.bias"], eps)
        q = F.linear(x, sd[p + "attention.q_proj.weight"], sd[p + "attention.q_proj.bias"]).view(B, T, nh, hd).transpose(1, 2)
        k = F.linear(x, sd[p + "attention.k_proj.weight"], sd[p + "attention.k_proj.bias"]).view(B, T, nh, hd).transpose(1, 2)
        v = F.linear(x, sd[p + "attention.v_proj.weight"], sd[p + "attention.v_proj.bias"]).view(B, T, nh, hd).transpose(1, 2)
        s = torch.matmul(q, k.transpose(2, 3)) * (hd ** -0.5)                         # hf:438-463
        if keep is not None and not bool(keep.all()):
            s = s.masked_fill(~keep[:, None, None, :], float("-inf"))
        a = torch.matmul(torch.softmax(s, dim=-1), v).transpose(1, 2).reshape(B, T, nh * hd)
        h = h + F.linear(a, sd[p + "attention.out_proj.weight"], sd[p + "attention.out_proj.bias"])
        x = F.layer_norm(h, (h.shape[-1],), sd[p + "final_layer_norm.weight"], sd[p + "final_layer_norm.bias"], eps)
        x = F.gelu(F.linear(x, sd[p + "feed_forward.intermediate_dense.weight"], sd[p + "feed_forward.intermediate_dense.bias"]))
        h = h + F.linear(x, sd[p + "feed_forward.output_dense.weight"], sd[p + "feed_forward.output_dense.bias"])
    h = F.layer_norm(h, (h.shape[-1],), sd["encoder.layer_norm.weight"], sd["encoder.layer_norm.bias"], eps)  # hf:791
    hidden_states.append(h)
    return h, hidden_states


def audio_forward(sd: SD, cfg: dict, wav: Tensor, attention_mask: Optional[Tensor]) -> Tuple[Tensor, Tensor]:
    """AudioEncoder.forward, model/encoder.py:89-100.  ``sd`` keys carry the ``model.`` prefix."""
    last, hs = w2v2_forward(sub(sd, "model."), cfg, wav, attention_mask)
    mid = torch.stack(hs[6:10], dim=0).mean(dim=0)
    return last, mid


# ----------------------------------------------------------------------------------------------------------
# a12-a14  CrossAttentionFusion (model/fusion_module.py:29-67)
# ----------------------------------------------------------------------------------------------------------
def downsample_mask(mask: Tensor, t_out: int) -> Tensor:
    """F.interpolate(mode='nearest') legacy law src = floor(i*in/out) (trainer.py:98)."""
    t_in = mask.shape[-1]
    # torch computes floor(i * float32(in/out)); restated with the same float32 scale
    scale = torch.tensor(t_in / t_out, dtype=torch.float32)
    idx_f = torch.floor(torch.arange(t_out, dtype=torch.float32) * scale).long().clamp_(max=t_in - 1)
    return mask[..., idx_f]


def gather_speech(audio_feat: Tensor, mask: Tensor) -> Tuple[Tensor, Tensor]:
    """fusion_module.py:40-48: keep frames with mask in {1,2}, left-pack, zero-pad to the batch max."""
    B, T, D = audio_feat.shape
    sm = (mask != 0) & (mask != 3)
    cnt = sm.sum(1)
    tmax = int(cnt.max())
    out = audio_feat.new_zeros(B, tmax, D)
    om = mask.new_zeros(B, tmax)
    for b in range(B):
        n = int(cnt[b])
        out[b, :n] = audio_feat[b][sm[b]]
        om[b, :n] = mask[b][sm[b]]
    return out, om


def lerp_time(x: Tensor, t_out: int) -> Tensor:
    """F.interpolate(linear, align_corners=True) over time (fusion_module.py:51). x [B,T,D] -> [B,t_out,D]."""
    B, T, D = x.shape
    if T == t_out:
        return x
    if t_out == 1:
        return x[:, :1]
    pos = torch.arange(t_out, dtype=torch.float32) * (float(T - 1) / float(t_out - 1))
    i0 = pos.floor().long().clamp_(max=T - 1)
    i1 = (i0 + 1).clamp_(max=T - 1)
    w1 = (pos - i0.float())[None, :, None]
    return x[:, i0] * (1.0 - w1) + x[:, i1] * w1


def mha(sd: SD, p: str, q_in: Tensor, kv_in: Tensor, nh: int) -> Tensor:
    """nn.MultiheadAttention(batch_first) forward without masks/dropout (torch:functional.py:6206-6606):
    packed in-proj, q scaled by 1/sqrt(hd) BEFORE QK^T, out-proj."""
    E = q_in.shape[-1]
    W, bI = sd[p + "in_proj_weight"], sd[p + "in_proj_bias"]
    B, Tq, _ = q_in.shape
    Tk = kv_in.shape[1]
    hd = E // nh
    q = F.linear(q_in, W[:E], bI[:E]).view(B, Tq, nh, hd).transpose(1, 2) * (1.0 / math.sqrt(hd))
    k = F.linear(kv_in, W[E:2 * E], bI[E:2 * E]).view(B, Tk, nh, hd).transpose(1, 2)
    v = F.linear(kv_in, W[2 * E:], bI[2 * E:]).view(B, Tk, nh, hd).transpose(1, 2)
    a = torch.softmax(q @ k.transpose(2, 3), dim=-1) @ v
    a = a.transpose(1, 2).reshape(B, Tq, E)
    return F.linear(a, sd[p + "out_proj.weight"], sd[p + "out_proj.bias"])


def lstm_dir(x: Tensor, w_ih: Tensor, w_hh: Tensor, b_ih: Tensor, b_hh: Tensor, reverse: bool) -> Tensor:
    """One direction of nn.LSTM: gates i,f,g,o; c=f*c+i*g; h=o*tanh(c); zero initial state."""
    B, T, _ = x.shape
    H = w_hh.shape[1]
    gx = F.linear(x, w_ih, b_ih + b_hh)
    h = x.new_zeros(B, H)
    c = x.new_zeros(B, H)
    outs = [None] * T
    order = range(T - 1, -1, -1) if reverse else range(T)
    for t in order:
        g = gx[:, t] + h @ w_hh.t()
        i, f, gg, o = g.chunk(4, dim=1)
        c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
        h = torch.sigmoid(o) * torch.tanh(c)
        outs[t] = h
    return torch.stack(outs, dim=1)


def bilstm2(sd: SD, p: str, x: Tensor) -> Tensor:
    """nn.LSTM(num_layers=2, bidirectional=True, batch_first=True), fusion_module.py:21-27,64."""
    for layer in (0, 1):
        f = lstm_dir(x, sd[f"{p}weight_ih_l{layer}"], sd[f"{p}weight_hh_l{layer}"],
                     sd[f"{p}bias_ih_l{layer}"], sd[f"{p}bias_hh_l{layer}"], False)
        r = lstm_dir(x, sd[f"{p}weight_ih_l{layer}_reverse"], sd[f"{p}weight_hh_l{layer}_reverse"],
                     sd[f"{p}bias_ih_l{layer}_reverse"], sd[f"{p}bias_hh_l{layer}_reverse"], True)
        x = torch.cat([f, r], dim=-1)
    return x


def fusion_forward(sd: SD, visual_feat: Tensor, audio_feat: Tensor, mask: Tensor, num_heads: int = 4
                   ) -> Tuple[Tensor, Tensor]:
    """CrossAttentionFusion.forward, fusion_module.py:29-67."""
    T_v = visual_feat.shape[1]
    a, m = gather_speech(audio_feat, mask)
    if a.shape[1] != T_v:
        a = lerp_time(a, T_v)
        m = downsample_mask(m, T_v)
    v = F.linear(visual_feat, sd["visual_proj.weight"], sd["visual_proj.bias"])
    a = F.linear(a, sd["audio_proj.weight"], sd["audio_proj.bias"])
    a2v = mha(sd, "cross_attn_audio.", a, v, num_heads)
    fused = F.linear(a2v, sd["fusion_proj.weight"], sd["fusion_proj.bias"])
    seq = bilstm2(sd, "temporal_model.", fused)
    return seq, (m != 0).sum(1)


# ----------------------------------------------------------------------------------------------------------
# a15-a17  CTC head, CTC loss, contrastive loss
# ----------------------------------------------------------------------------------------------------------
def decoder_forward(sd: SD, x: Tensor) -> Tensor:
    """CTCDecoder.forward (target=None), model/decoder.py:24-25,35."""
    return F.log_softmax(F.linear(x, sd["net.0.weight"], sd["net.0.bias"]), dim=-1)


def ctc_loss(log_probs: Tensor, targets: Tensor, in_len: Tensor, tgt_len: Tensor, blank: int) -> Tensor:
    """nn.CTCLoss(blank, zero_infinity=True), mean reduction (trainer.py:25,116-117)."""
    return F.ctc_loss(log_probs.transpose(0, 1), targets, in_len, tgt_len, blank=blank, reduction="mean",
                      zero_infinity=True)


def contrastive(mid: Tensor, flat_mask: Tensor, pw: Optional[Tensor], pb: Optional[Tensor]) -> Tensor:
    """contrastive_loss_with_mask, contrastive.py:8-44 (temperature .07, weights 1.0/0.3, mean over ALL entries)."""
    B, T, D = mid.shape
    f = mid.reshape(B * T, D)
    valid = flat_mask != 3
    f, m = f[valid], flat_mask[valid]
    if pw is not None:
        f = F.linear(f, pw, pb)
    f = F.normalize(f, dim=1)
    strong, weak, neg = f[m == 2], f[m == 1], f[m == 0]
    total = torch.zeros((), dtype=mid.dtype)
    if len(weak) > 0 and len(strong) > 0:
        total = total + 1.0 * (-F.log_softmax(weak @ strong.t() / 0.07, dim=1).mean())
    if len(weak) > 0 and len(neg) > 0:
        total = total + 0.3 * (-F.log_softmax(weak @ neg.t() / 0.07, dim=1).mean())
    return total


def greedy_ctc(log_probs_tv: Tensor, blank: int) -> List[int]:
    """== beam_search.simple_beam_search best beam: per-frame argmax, collapse repeats, drop blank
    (prev IS updated on blank, beam_search.py:36-40)."""
    ids = log_probs_tv.argmax(-1).tolist()
    out, prev = [], None
    for i in ids:
        if i != prev and i != blank:
            out.append(i)
        prev = i
    return out


# ----------------------------------------------------------------------------------------------------------
# a18/a22  one training step (trainer.py:62-125) with the freeze policy of main.py:26-31,100-106
# ----------------------------------------------------------------------------------------------------------
def trainable_keys(audio_sd: SD, fusion_sd: SD, dec_sd: SD) -> Dict[str, List[str]]:
    aud = [k for k in audio_sd if any(f"encoder.layers.{i}." in k for i in range(6, 10))]
    return {"audio": aud, "fusion": list(fusion_sd.keys()), "decoder": list(dec_sd.keys())}


def forward_losses(vis: SD, aud: SD, fus: SD, dec: SD, cfg: dict, batch: Dict[str, Tensor], proj: Tuple[Tensor, Tensor],
                   training: bool, blank: int = 3, lambda_: float = 0.1, dedup_audio: bool = True) -> Dict[str, Tensor]:
    """trainer.py:66-119 for one batch (fp32, deterministic mode)."""
    lip1 = batch["lip1"].permute(0, 2, 1, 3, 4).contiguous()
    lip2 = batch["lip2"].permute(0, 2, 1, 3, 4).contiguous()
    audio, mask1, mask2 = batch["audio"], batch["mask1"], batch["mask2"]
    vf1 = visual_forward(vis, lip1, training)
    vf2 = visual_forward(vis, lip2, training)
    a1, mid1 = audio_forward(aud, cfg, audio, mask1 != 3)
    if dedup_audio:          # attn_mask1 == attn_mask2 always -> identical tensors in deterministic mode
        a2, mid2 = a1, mid1
    else:
        a2, mid2 = audio_forward(aud, cfg, audio, mask2 != 3)
    T_enc = a1.shape[1]
    m1 = downsample_mask(mask1, T_enc)
    m2 = downsample_mask(mask2, T_enc)
    c1 = contrastive(mid1, m1.reshape(-1), proj[0], proj[1])
    c2 = contrastive(mid2, m2.reshape(-1), proj[0], proj[1])
    f1, il1 = fusion_forward(fus, vf1, a1, m1)
    f2, il2 = fusion_forward(fus, vf2, a2, m2)
    lp1 = decoder_forward(dec, f1)
    lp2 = decoder_forward(dec, f2)
    l1 = ctc_loss(lp1, batch["text1"], il1, batch["text1_lengths"], blank)
    l2 = ctc_loss(lp2, batch["text2"], il2, batch["text2_lengths"], blank)
    total = (l1 + l2) / 2 + lambda_ * (c1 + c2) / 2
    return dict(visual_feat1=vf1, visual_feat2=vf2, audio_last=a1, audio_mid=mid1, mask1_ds=m1, mask2_ds=m2,
                fused1=f1, fused2=f2, input_lengths1=il1, input_lengths2=il2, log_probs1=lp1, log_probs2=lp2,
                loss1=l1, loss2=l2, contrast1=c1, contrast2=c2, total=total)


def adam_step(params: Dict[str, Tensor], grads: Dict[str, Tensor], state: Dict[str, dict], lr: float,
              betas=(0.9, 0.999), eps: float = 1e-8) -> None:
    """torch.optim.Adam defaults, no weight decay (trainer.py:34-39)."""
    for k, p in params.items():
        g = grads.get(k)
        if g is None:
            continue
        st = state.setdefault(k, dict(step=0, m=torch.zeros_like(p), v=torch.zeros_like(p)))
        st["step"] += 1
        st["m"].mul_(betas[0]).add_(g, alpha=1 - betas[0])
        st["v"].mul_(betas[1]).addcmul_(g, g, value=1 - betas[1])
        bc1 = 1 - betas[0] ** st["step"]
        bc2 = 1 - betas[1] ** st["step"]
        denom = (st["v"].sqrt() / math.sqrt(bc2)).add_(eps)
        p.data.addcdiv_(st["m"], denom, value=-lr / bc1)


def train_step(vis: SD, aud: SD, fus: SD, dec: SD, cfg: dict, batch: Dict[str, Tensor], proj, opt_state: Dict[str, dict],
               lr: float = 1e-4, lr_audio: float = 2e-5, lambda_: float = 0.1, blank: int = 3, dedup_audio: bool = True):
    """One full deterministic fp32 training step: forward, autograd backward, Adam.  Returns (outputs, grads)."""
    tk = trainable_keys(aud, fus, dec)
    leaves = {}
    for name, sd, keys in (("audio", aud, tk["audio"]), ("fusion", fus, tk["fusion"]), ("decoder", dec, tk["decoder"])):
        for k in keys:
            sd[k] = sd[k].detach().requires_grad_(True)
            leaves[f"{name}.{k}"] = sd[k]
    out = forward_losses(vis, aud, fus, dec, cfg, batch, proj, True, blank, lambda_, dedup_audio)
    names = list(leaves)
    gl = torch.autograd.grad(out["total"], [leaves[n] for n in names], allow_unused=True)
    grads = {n: g for n, g in zip(names, gl)}
    with torch.no_grad():
        for name, sd, keys, lr_ in (("audio", aud, tk["audio"], lr_audio), ("fusion", fus, tk["fusion"], lr),
                                    ("decoder", dec, tk["decoder"], lr)):
            ps = {f"{name}.{k}": sd[k] for k in keys}
            adam_step(ps, grads, opt_state, lr_)
            for k in keys:
                sd[k] = sd[k].detach()
    return {k: (v.detach() if torch.is_tensor(v) else v) for k, v in out.items()}, grads
