"""wav2vec2 (stable-layer-norm / XLSR architecture) forward + backward on the HIP kernels.

Mirrors the arithmetic of HuggingFace ``Wav2Vec2Model`` as the reference uses it (model/encoder.py:80-100;
hf:275-299 feature encoder, hf:422-434 projection, hf:326-379 positional conv, hf:611-654 + 729-802 pre-LN
encoder, hf:997-1036 length law) with the parameter names of the HF checkpoint, so ``main.py:26-31``'s
name-based freeze policy and ``load_state_dict`` keep working.  Config-driven.

Layout in HBM: activations are channel-last ``[B, T, C]``; the residual stream is fp32, GEMM operands are in the
compute dtype (fp32 parity mode / bf16 perf mode); Q,K,V live in one packed ``[B, T, 3, heads, hd]`` buffer.
Backward is hand-written (no autograd graph inside): it stops at the lowest layer that owns a trainable
parameter, so nothing below encoder layer 6 is ever differentiated.
"""
from __future__ import annotations

import ctypes
import json
import os
from types import SimpleNamespace
from typing import Dict, List, Optional

import torch
import torch.nn as nn

from .. import _lib as L
from .. import ops
from ..precision import compute_dtype, is_lp
from ..utils import init as _init
from ..utils.shadow import ParamCache

Tensor = torch.Tensor
# attention-dropout keep bits evaluated once per step on a side stream (0: every kernel generates its masks itself; A/B runs)
DROP_BITS = os.environ.get("AVAMD_ATTN_DROPBITS", "1") != "0"
DROP_BITS_ALL = os.environ.get("AVAMD_ATTN_DROPBITS_ALL", "0") != "0"
# FFN activation site saves its gradient factor instead of the pre-activation (bf16 mode); AVAMD_FFN_GF=0 = recompute gelu' and the mask in the backward
FFN_GF = os.environ.get("AVAMD_FFN_GF", "1") != "0"
# one native call per encoder layer forward (av_w2v2_layer_fwd) instead of seven per-kernel calls: same kernels, same arguments, fewer host round trips
# - for the launch-bound regime only (at most NATIVE_MAX_ROWS tokens per pass: config 1's 2 x 1 s step 21.1 -> 17.8 ms); at the benchmark size the
# device is the bottleneck and the burst of seven launches measured 0.9 % SLOWER (1001 / 1008 vs 1014 / 1015 utt/s, same box) - the per-kernel
# calls there interleave better with the other pass's and the lip streams' launches.  AVAMD_W2V2_NATIVE=0 never, 2 always.
NATIVE_LAYER = int(os.environ.get("AVAMD_W2V2_NATIVE", "1"))
NATIVE_MAX_ROWS = 4096
# the second audio pass of a step runs on its own stream beside the first (HBM-bound row kernels of one pass overlap MFMA-bound GEMMs of the other)
PASS_STREAMS = os.environ.get("AVAMD_PASS_STREAMS", "1") != "0"


def param_shapes(cfg: dict) -> Dict[str, tuple]:
    H, I = cfg["hidden_size"], cfg["intermediate_size"]
    shapes: Dict[str, tuple] = {"masked_spec_embed": (H,)}
    cin = 1
    for i, (c, k) in enumerate(zip(cfg["conv_dim"], cfg["conv_kernel"])):
        p = f"feature_extractor.conv_layers.{i}"
        shapes[p + ".conv.weight"] = (c, cin, k); shapes[p + ".conv.bias"] = (c,)
        shapes[p + ".layer_norm.weight"] = (c,); shapes[p + ".layer_norm.bias"] = (c,)
        cin = c
    shapes["feature_projection.layer_norm.weight"] = (cin,); shapes["feature_projection.layer_norm.bias"] = (cin,)
    shapes["feature_projection.projection.weight"] = (H, cin); shapes["feature_projection.projection.bias"] = (H,)
    kp, gp = cfg["num_conv_pos_embeddings"], cfg["num_conv_pos_embedding_groups"]
    shapes["encoder.pos_conv_embed.conv.bias"] = (H,)
    shapes["encoder.pos_conv_embed.conv.parametrizations.weight.original0"] = (1, 1, kp)
    shapes["encoder.pos_conv_embed.conv.parametrizations.weight.original1"] = (H, H // gp, kp)
    shapes["encoder.layer_norm.weight"] = (H,); shapes["encoder.layer_norm.bias"] = (H,)
    for li in range(cfg["num_hidden_layers"]):
        p = f"encoder.layers.{li}"
        for nm in ("k_proj", "v_proj", "q_proj", "out_proj"):
            shapes[f"{p}.attention.{nm}.weight"] = (H, H); shapes[f"{p}.attention.{nm}.bias"] = (H,)
        shapes[p + ".layer_norm.weight"] = (H,); shapes[p + ".layer_norm.bias"] = (H,)
        shapes[p + ".feed_forward.intermediate_dense.weight"] = (I, H); shapes[p + ".feed_forward.intermediate_dense.bias"] = (I,)
        shapes[p + ".feed_forward.output_dense.weight"] = (H, I); shapes[p + ".feed_forward.output_dense.bias"] = (H,)
        shapes[p + ".final_layer_norm.weight"] = (H,); shapes[p + ".final_layer_norm.bias"] = (H,)
    return shapes


def build_param_tree(root: nn.Module, shapes: Dict[str, tuple]) -> None:
    """Nested plain nn.Modules whose dotted parameter names equal the HF checkpoint keys."""
    for name, shape in shapes.items():
        parts = name.split(".")
        m = root
        for p in parts[:-1]:
            if p not in m._modules:
                m.add_module(p, nn.Module())
            m = m._modules[p]
        m.register_parameter(parts[-1], nn.Parameter(torch.zeros(shape, dtype=torch.float32)))


def conv_out_lengths(cfg: dict, n):
    """hf:997-1015  L <- floor((L-k)/s)+1 (works on ints and integer tensors)."""
    for k, s in zip(cfg["conv_kernel"], cfg["conv_stride"]):
        n = (n - k) // s + 1
    return n


STOCHASTIC_DEFAULTS = dict(hidden_dropout=0.0, attention_dropout=0.0, activation_dropout=0.0, feat_proj_dropout=0.0, layerdrop=0.0,
                           mask_time_prob=0.0, mask_time_length=10, mask_time_min_masks=2, mask_feature_prob=0.0, mask_feature_length=10,
                           mask_feature_min_masks=0)
# dropout streams: layer*8 + site; sites 0 attention-out, 1 FFN activation, 2 FFN out, 3 attention probabilities
S_FEATPROJ, S_POS = 0xF0000001, 0xF0000002


def specaugment_mask(batch: int, seq_len: int, mask_prob: float, mask_length: int, lengths, min_masks: int):
    """Time-mask spans of SpecAugment as HF draws them (hf:101-217 `_compute_mask_indices`): same numpy-global-RNG call
    sequence (one `rand`, then one `choice(..., replace=False)` per item), so a given `np.random.seed` gives the same mask."""
    import numpy as np
    if mask_length < 1 or mask_length > seq_len:
        raise ValueError(f"mask_length={mask_length} must be in [1, {seq_len}]")
    eps = np.random.rand(1).item()

    def nspans(L):
        n = max(int(mask_prob * L / mask_length + eps), min_masks)
        if n * mask_length > seq_len:
            n = seq_len // mask_length
        if L - (mask_length - 1) < n:
            n = max(L - (mask_length - 1), 0)
        return n

    mask = np.zeros((batch, seq_len), dtype=bool)
    max_n = nspans(seq_len)
    if max_n == 0:
        return mask
    for b, L in enumerate(lengths):
        n = nspans(int(L))
        idx = np.random.choice(np.arange(int(L) - (mask_length - 1)), n, replace=False)
        dummy = seq_len - 1 if len(idx) == 0 else int(idx[0])
        for s0 in list(idx) + [dummy] * (max_n - n):
            for o in range(mask_length):
                mask[b, min(int(s0) + o, seq_len - 1)] = True
    return mask


class Wav2Vec2ModelHIP(nn.Module):
    """Drop-in for the ``transformers.Wav2Vec2Model`` instance the reference stores in ``AudioEncoder.model``."""

    def __init__(self, cfg: dict):
        super().__init__()
        self.cfg = dict(cfg)
        self.config = SimpleNamespace(**cfg, output_hidden_states=True)
        build_param_tree(self, param_shapes(cfg))
        self._cache = ParamCache()
        self._names = [n for n, _ in self.named_parameters()]
        # data-parallel hooks (parallel/dp.py): per-layer gradient bucket -> async all-reduce, joined at the end of backward
        self.grad_ready = None
        self.grad_flat_ready = None     # the same for a layer whose gradients already live in one flat buffer (GradArena): no packing copy
        self.grad_wait = None
        self._arenas = {}               # layer -> GradArena: persistent flat gradient buffer the backward kernels write into
        self.grad_pre = None            # called when this model's backward starts (the trainer reduces the head gradients there)
        # host RNG sources of the train-mode regularisers.  None = torch's global generator (what HF uses).  Under data parallelism the
        # trainer installs a LayerDrop generator that is IDENTICAL on every rank (a dropped trainable layer issues no gradient bucket, so
        # ranks that disagreed would issue different collective sequences) and a dropout-seed generator that DIFFERS per rank
        self.layerdrop_generator = None
        self.dropout_generator = None
        # bookkeeping of the LayerDrop draws: encoder layers executed forward / with a backward / with weight gradients since the last reset
        # (all passes), and optionally the list of dropped layers per pass (diagnostics, tests)
        self.layers_executed = self.layers_executed_bwd = self.layers_executed_bwd_tr = 0
        self.dropped_log = None

    # ---- parameter access ------------------------------------------------------------------------------------
    def P(self, name: str) -> Tensor:
        pc = self.__dict__.setdefault("_pcache", {})
        hit = pc.get(name)
        if hit is None:
            m = self
            for p in name.split("."):
                m = m._modules[p] if p in m._modules else m._parameters[p]
            pc[name] = hit = m
        return hit

    def c(self, name: str, dtype) -> Tensor:
        """Parameter in the compute dtype (cached)."""
        p = self.P(name)
        if dtype == torch.float32:
            return p.data
        return self._cache.get(("c", name), [p], dtype, lambda: ops.cast(p.data, dtype), flat=True)

    def qkv_w(self, li: int, dtype):
        ps = [self.P(f"encoder.layers.{li}.attention.{n}_proj.weight") for n in ("q", "k", "v")]
        return self._cache.get(("qkvw", li), ps, dtype, lambda: ops.cast(torch.cat([p.data for p in ps], 0).contiguous(), dtype), flat=True)

    def qkv_b(self, li: int):
        ps = [self.P(f"encoder.layers.{li}.attention.{n}_proj.bias") for n in ("q", "k", "v")]
        return self._cache.get(("qkvb", li), ps, torch.float32, lambda: torch.cat([p.data for p in ps], 0).contiguous())

    def conv_w(self, i: int, dtype):
        """Conv1d weight [C_out, C_in, k] -> [C_out, k*C_in] (tap-major K, matching the channel-last im2col rows)."""
        p = self.P(f"feature_extractor.conv_layers.{i}.conv.weight")
        return self._cache.get(("convw", i), [p], dtype,
                               lambda: ops.cast(p.data.permute(0, 2, 1).contiguous().view(p.shape[0], -1), dtype))

    def pos_w(self, dtype):
        """weight_norm(dim=2) folded once (frozen): w = v*g/||v||_(0,1); layout [group][c_out][tap][c_in] (hf:355)."""
        g = self.P("encoder.pos_conv_embed.conv.parametrizations.weight.original0")
        v = self.P("encoder.pos_conv_embed.conv.parametrizations.weight.original1")
        G = self.cfg["num_conv_pos_embedding_groups"]

        def make():
            w = v.data * (g.data / v.data.norm(2, dim=(0, 1), keepdim=True))          # [H, H/G, k]
            Hd, Cg, k = w.shape
            w = w.view(G, Hd // G, Cg, k).permute(0, 1, 3, 2).contiguous()            # [G, co, k, ci]
            return ops.cast(w.view(G, Hd // G, k * Cg), dtype)
        return self._cache.get(("posw",), [g, v], dtype, make)

    def warm_caches(self, dtype) -> None:
        """Materialise, on the CURRENT stream, every compute-dtype copy / re-layout the forward reads (and rebuild the stale ones).  The two
        audio passes of a step run on two streams: a cache rebuilt lazily by pass 1's host code is written on the main stream AFTER the
        point the pass-2 stream waited for, i.e. pass 2 could read a freshly allocated, not yet written buffer (the cause of the divergence
        of the as-executed batch-64 run in round 2, profiles/r03_nan_hunt_*).  Called before the streams fork, so that no cache is ever
        built inside a forked region."""
        if dtype == torch.float32:
            self.pos_w(dtype)
            for li in range(self.cfg["num_hidden_layers"]):
                self.qkv_w(li, dtype); self.qkv_b(li)
            for i in range(1, len(self.cfg["conv_kernel"])):
                self.conv_w(i, dtype)
            return
        for i in range(1, len(self.cfg["conv_kernel"])):
            self.conv_w(i, dtype)
        self.c("feature_projection.projection.weight", dtype)
        self.pos_w(dtype)
        for li in range(self.cfg["num_hidden_layers"]):
            p = f"encoder.layers.{li}."
            self.qkv_w(li, dtype); self.qkv_b(li)
            self.c(p + "attention.out_proj.weight", dtype)
            self.c(p + "feed_forward.intermediate_dense.weight", dtype)
            self.c(p + "feed_forward.output_dense.weight", dtype)

    # ---- which layers need a backward -----------------------------------------------------------------------
    def trainable_layers(self) -> List[bool]:
        if getattr(self, "_layer_params", None) is None:
            self._layer_params = [list(self._modules["encoder"]._modules["layers"]._modules[str(li)].parameters())
                                  for li in range(self.cfg["num_hidden_layers"])]
        return [any(p.requires_grad for p in ps) for ps in self._layer_params]

    def check_freeze_policy(self):
        bad = [n for n, p in self.named_parameters() if p.requires_grad and not n.startswith("encoder.layers.")]
        if bad:
            raise NotImplementedError(
                "the HIP backward covers encoder.layers.* (the reference trains layers 6-9 only, main.py:26-31); "
                f"freeze these parameters: {bad[:4]}{'...' if len(bad) > 4 else ''}")

    # ---- forward ---------------------------------------------------------------------------------------------
    def features(self, wav: Tensor, dtype) -> Tensor:
        """K1: 7 x (conv + LayerNorm + GELU) -> [B, T_enc, C] in the compute dtype (frozen, nothing saved)."""
        cfg = self.cfg
        B, T_in = wav.shape
        ks, ss, cs = cfg["conv_kernel"], cfg["conv_stride"], cfg["conv_dim"]
        L0 = (T_in - ks[0]) // ss[0] + 1
        p = "feature_extractor.conv_layers.0."
        h = torch.empty((B, L0, cs[0]), dtype=dtype, device=wav.device)
        L.check(L.lib().av_conv0_ln_gelu(ops.ptr(wav), ops.ptr(self.P(p + "conv.weight").data), ops.ptr(self.P(p + "conv.bias").data),
                                         ops.ptr(self.P(p + "layer_norm.weight").data), ops.ptr(self.P(p + "layer_norm.bias").data),
                                         ops.ptr(h), ops.dt(h), B, T_in, L0, cs[0], ks[0], ss[0], 1e-5, ops.stream()), "av_conv0_ln_gelu")
        Lin, cin = L0, cs[0]
        for i in range(1, len(ks)):
            p = f"feature_extractor.conv_layers.{i}."
            Lout = (Lin - ks[i]) // ss[i] + 1
            y = torch.empty((B, Lout, cs[i]), dtype=dtype, device=wav.device)
            # strided GEMM: im2col row t of a channel-last signal is the contiguous slice x[s*t : s*t+k, :]
            ops.gemm(h, self.conv_w(i, dtype), y, M=Lout, N=cs[i], K=ks[i] * cin, lda=ss[i] * cin, ldb=ks[i] * cin, ldc=cs[i],
                     bias=self.P(p + "conv.bias").data, batch=B, sA=Lin * cin, sC=Lout * cs[i])
            h = ops.layernorm_fwd(y, self.P(p + "layer_norm.weight").data, self.P(p + "layer_norm.bias").data, out_dtype=dtype,
                                  eps=1e-5, act=L.ACT_GELU)
            Lin, cin = Lout, cs[i]
        return h

    def encode(self, wav: Tensor, attention_mask: Optional[Tensor], save: bool, valid_lengths=None):
        """Returns (last fp32, mid fp32, ctx).  ctx holds what backward needs when ``save``.  ``valid_lengths`` (optional host ints: number
        of valid samples per item = attention_mask.sum(-1)) keeps the length law, the frame mask and SpecAugment's span count on the host:
        no device index arithmetic (~25 tiny launches) and no device -> host read-back per pass."""
        cfg = self.cfg
        dtype = compute_dtype()
        dev = wav.device
        eps = cfg["layer_norm_eps"]
        Hd, nh = cfg["hidden_size"], cfg["num_attention_heads"]
        hd = Hd // nh
        nl = cfg["num_hidden_layers"]
        # train-mode stochastic regularisers of wav2vec2 (hf:701,774,1272-1316); all probabilities default to 0
        sc = {k: cfg.get(k, v) for k, v in STOCHASTIC_DEFAULTS.items()}
        tm = self.training
        hd_p = sc["hidden_dropout"] if tm else 0.0
        at_p = sc["attention_dropout"] if tm else 0.0
        ac_p = sc["activation_dropout"] if tm else 0.0
        fp_p = sc["feat_proj_dropout"] if tm else 0.0
        ld_p = sc["layerdrop"] if tm else 0.0
        seed = int(torch.randint(0, 2 ** 62, (1,), generator=self.dropout_generator).item()) if (hd_p or at_p or ac_p or fp_p) else 0
        # the conv feature extractor is frozen and has no stochastic op, so two passes over the SAME waveform tensor (the reference's
        # audio_encoder(audio, mask1) / (audio, mask2), model/trainer.py:88-96) share its output while the trainer holds the window open
        fc = getattr(self, "_feat_cache", None)
        if fc is not None and fc.get("src") is wav and fc.get("dtype") == dtype:
            feats = fc["feats"]
        else:
            src = wav
            wav = wav.contiguous().float()
            feats = self.features(wav, dtype)
            if fc is not None:
                fc["src"], fc["dtype"], fc["feats"] = src, dtype, feats
                if feats.is_cuda:                                # a pass on another stream waits for exactly this point
                    fc["evt"] = torch.cuda.Event(); fc["evt"].record()
        B, T, C = feats.shape
        klen = keep = None
        n_host = None
        want_sm = tm and sc["mask_time_prob"] > 0 and cfg.get("apply_spec_augment", True)
        smt = None
        if attention_mask is not None and valid_lengths is not None:
            # host-side metadata of the pass in ONE upload: key lengths (int32), frame keep mask, SpecAugment time mask (hf:1272-1296, drawn
            # from numpy's global RNG as HF does - before the feature-axis mask, which keeps HF's draw order)
            import numpy as np
            n_host = np.asarray([int(conv_out_lengths(cfg, int(v))) for v in valid_lengths], dtype=np.int64)
            o_keep = (4 * B + 15) // 16 * 16
            o_sm = o_keep + (B * T + 15) // 16 * 16
            pack = np.zeros(o_sm + (B * T if want_sm else 0), dtype=np.uint8)
            pack[:4 * B] = np.clip(n_host, 1, T).astype(np.int32).view(np.uint8)
            pack[o_keep:o_keep + B * T] = (np.arange(T)[None, :] < n_host[:, None]).astype(np.uint8).reshape(-1)
            if want_sm:
                sm = specaugment_mask(B, T, sc["mask_time_prob"], sc["mask_time_length"], n_host.tolist(), sc["mask_time_min_masks"])
                pack[o_sm:o_sm + B * T] = sm.astype(np.uint8).reshape(-1)
            dpack = ops.h2d_async(pack, dev)
            klen = dpack[:4 * B].view(torch.int32)
            keep = dpack[o_keep:o_keep + B * T].view(B, T)
            if want_sm:
                smt = dpack[o_sm:o_sm + B * T].view(B, T)
        elif attention_mask is not None:
            n = conv_out_lengths(cfg, attention_mask.long().sum(-1))
            klen = n.clamp(min=1, max=T).to(torch.int32)
            keep = (torch.arange(T, device=dev)[None, :] < n[:, None]).to(torch.uint8).contiguous()
        x = ops.layernorm_fwd(feats, self.P("feature_projection.layer_norm.weight").data,
                              self.P("feature_projection.layer_norm.bias").data, out_dtype=dtype, eps=eps)
        h = ops.linear(x, self.c("feature_projection.projection.weight", dtype), self.P("feature_projection.projection.bias").data,
                       out_dtype=torch.float32)
        if fp_p > 0:
            h = ops.cast_dropout(h, torch.float32, (fp_p, seed, S_FEATPROJ))             # hf:433
        if want_sm:                                                                      # hf:1272-1296 (host numpy RNG, as HF)
            if smt is None:
                lengths = n.tolist() if attention_mask is not None else [T] * B
                sm = specaugment_mask(B, T, sc["mask_time_prob"], sc["mask_time_length"], lengths, sc["mask_time_min_masks"])
                smt = ops.h2d_async(sm.astype("uint8"), dev)
            L.check(L.lib().av_overwrite_rows(ops.ptr(h), ops.dt(h), ops.ptr(smt), ops.ptr(self.P("masked_spec_embed").data), B * T, Hd,
                                              ops.stream()), "av_overwrite_rows")
        if tm and sc["mask_feature_prob"] > 0 and cfg.get("apply_spec_augment", True):   # hf:1298-1316: drawn AFTER the time mask, same numpy RNG
            fmask = specaugment_mask(B, Hd, sc["mask_feature_prob"], sc["mask_feature_length"], [Hd] * B, sc["mask_feature_min_masks"])
            fmt = ops.h2d_async(fmask.astype("uint8"), dev)
            L.check(L.lib().av_zero_feature_cols(ops.ptr(h), ops.dt(h), ops.ptr(fmt), B, T, Hd, ops.stream()), "av_zero_feature_cols")
        if keep is not None:
            ops.mask_rows_(h, keep)                                                      # hf:752-755
        # positional conv (grouped, k taps) as an implicit-im2col GEMM per group, + bias, GELU, + residual
        kp, G = cfg["num_conv_pos_embeddings"], cfg["num_conv_pos_embedding_groups"]
        Cg = Hd // G
        hT = ops.cast(h, dtype)
        hs0 = torch.empty_like(h)
        conv = dict(cT=1, cH=T, cW=1, cCtot=Hd, cCin=Cg, cCoff=0, cKt=1, cKh=kp, cKw=1, cSh=1, cSw=1, cPt=0, cPh=kp // 2, cPw=0,
                    cOh=T, cOw=1)
        ops.gemm(hT, self.pos_w(dtype), hs0, M=B * T, N=Cg, K=kp * Cg, lda=0, ldb=kp * Cg, ldc=Hd, a_mode=L.A_CONV2D, conv=conv,
                 bias=self.P("encoder.pos_conv_embed.conv.bias").data, act=L.ACT_GELU, R=h, ldr=Hd, batch=G, sA=Cg, sB=Cg * kp * Cg,
                 sC=Cg, sR=Cg, sBias=Cg)
        h = hs0
        if hd_p > 0:
            h = ops.cast_dropout(h, torch.float32, (hd_p, seed, S_POS))                  # hf:765
        train = self.trainable_layers() if save else [False] * nl
        first = train.index(True) if any(train) else nl
        # LayerDrop decisions (hf:774-789), drawn up front in layer order (the same draws the loop would make)
        dropped = [bool(ld_p > 0 and float(torch.rand([], generator=self.layerdrop_generator)) < ld_p) for _ in range(nl)]
        self.layers_executed += nl - sum(dropped)                  # executed (not expected) work: bench.py's FLOP accounting
        if save:
            self.layers_executed_bwd += sum(1 for li in range(first, nl) if not dropped[li])
            self.layers_executed_bwd_tr += sum(1 for li in range(first, nl) if not dropped[li] and train[li])
        if self.dropped_log is not None:
            self.dropped_log.append([li for li, d in enumerate(dropped) if d])
        # Attention-dropout keep bits of the layers that get a backward: ONE generator evaluation per probability (instead of one in the
        # forward and two in the backward), on a side stream - the kernels depend on no data and are pure VALU work beside the GEMMs
        amasks, amask_evt = {}, None
        bits_from = 0 if DROP_BITS_ALL else first               # AVAMD_ATTN_DROPBITS_ALL=1: keep bits also for the layers without a backward
        if at_p > 0 and save and first < nl and dev.type == "cuda" and DROP_BITS and ops.attention_mask_shape_ok(dtype, B, T, T, hd):
            if getattr(self, "_mask_stream", None) is None:
                self._mask_stream = torch.cuda.Stream(device=dev)
            main = torch.cuda.current_stream(dev)
            self._mask_stream.wait_stream(main)
            amask_evts = {}
            with torch.cuda.stream(self._mask_stream):
                for li in range(bits_from, nl):
                    if not dropped[li]:
                        amasks[li] = ops.attention_dropmask(B, nh, T, T, (at_p, seed, li * 8 + 3), dev)
                        if li < first or li == nl - 1 or (li - first) % 6 == 5:           # layers without a backward come first and are
                            amask_evts[li] = torch.cuda.Event()                          # waited for one by one, the rest in groups
                            amask_evts[li].record(self._mask_stream)
                last_evt = torch.cuda.Event(); last_evt.record(self._mask_stream)
            amask_evt = (amask_evts, last_evt)
            for m in amasks.values():
                m.record_stream(main)
        mid = torch.empty_like(h) if nl >= 10 else None
        saved = [None] * nl
        scale = hd ** -0.5
        # one native call per layer (AVAMD_W2V2_NATIVE=0: one call per kernel; the per-kernel path also serves the fp32 parity mode and bench.py's
        # probe legs, whose timing hooks live in ops.gemm / ops.attention_fwd)
        native = (NATIVE_LAYER and (NATIVE_LAYER >= 2 or B * T <= NATIVE_MAX_ROWS) and is_lp(dtype) and dev.type == "cuda"
                  and ops.GemmProbe.active is None and ops.AttnProbe.active is None and Hd % nh == 0)
        if native and getattr(self, "_largs", None) is None:
            self._largs = L.W2v2LayerArgs()
        for li in range(nl):
            if mid is not None and 6 <= li <= 9:
                ops.axpby(0.25, h, 1.0 if li > 6 else 0.0, mid)                          # model/encoder.py:97-99 (the first term writes: no zero fill)
            p = f"encoder.layers.{li}."
            keep_ctx = save and li >= first
            if dropped[li]:                                                              # LayerDrop (hf:774-789): identity layer
                if keep_ctx:
                    saved[li] = "skipped"
                continue
            amask = amasks.get(li)
            if native:
                # the layer's seven launches from ONE native call (csrc/w2v2_layer.hip: the same kernels with the same arguments)
                if amask is not None and amask_evt is not None:
                    evts, last_evt = amask_evt
                    nxt = min((l for l in evts if l >= li), default=None)
                    torch.cuda.current_stream(dev).wait_event(evts.pop(nxt) if nxt is not None else last_evt)
                    for l in [l for l in evts if l < li]:
                        evts.pop(l)
                    if nxt is None:
                        amask_evt = None
                I = cfg["intermediate_size"]
                M = B * T
                x1 = torch.empty((B, T, Hd), dtype=dtype, device=dev); x2 = torch.empty((B, T, Hd), dtype=dtype, device=dev)
                st4 = torch.empty((4, M), dtype=torch.float32, device=dev)
                mu1, rs1, mu2, rs2 = st4[0], st4[1], st4[2], st4[3]
                qkv = torch.empty((B, T, 3, nh, hd), dtype=dtype, device=dev)
                ao = torch.empty((B, T, nh, hd), dtype=dtype, device=dev)
                lse = torch.empty((B, nh, T), dtype=torch.float32, device=dev) if keep_ctx else None
                h2 = torch.empty((B, T, Hd), dtype=torch.float32, device=dev); h3 = torch.empty((B, T, Hd), dtype=torch.float32, device=dev)
                u = torch.empty((B, T, I), dtype=dtype, device=dev) if keep_ctx else None
                g = torch.empty((B, T, I), dtype=dtype, device=dev)
                gf = FFN_GF and keep_ctx
                a = self._largs
                a.B, a.T, a.hidden, a.heads, a.inter, a.lp, a.gf, a.stream_base = B, T, Hd, nh, I, L.AV_BF16, int(gf), li * 8
                a.eps, a.scale, a.hd_p, a.at_p, a.ac_p, a.seed = eps, scale, hd_p, at_p, ac_p, seed
                a.ln1_g = self.P(p + "layer_norm.weight").data.data_ptr(); a.ln1_b = self.P(p + "layer_norm.bias").data.data_ptr()
                a.ln2_g = self.P(p + "final_layer_norm.weight").data.data_ptr(); a.ln2_b = self.P(p + "final_layer_norm.bias").data.data_ptr()
                a.b_qkv = self.qkv_b(li).data_ptr(); a.b_o = self.P(p + "attention.out_proj.bias").data.data_ptr()
                a.b_1 = self.P(p + "feed_forward.intermediate_dense.bias").data.data_ptr(); a.b_2 = self.P(p + "feed_forward.output_dense.bias").data.data_ptr()
                a.w_qkv = self.qkv_w(li, dtype).data_ptr(); a.w_o = self.c(p + "attention.out_proj.weight", dtype).data_ptr()
                a.w_1 = self.c(p + "feed_forward.intermediate_dense.weight", dtype).data_ptr(); a.w_2 = self.c(p + "feed_forward.output_dense.weight", dtype).data_ptr()
                a.h, a.klen, a.amask = h.data_ptr(), ops.ptr(klen), ops.ptr(amask)
                a.x1, a.qkv, a.ao, a.x2, a.u, a.g = x1.data_ptr(), qkv.data_ptr(), ao.data_ptr(), x2.data_ptr(), ops.ptr(u), g.data_ptr()
                a.mu1, a.rs1, a.lse, a.h2, a.mu2, a.rs2, a.h3 = mu1.data_ptr(), rs1.data_ptr(), ops.ptr(lse), h2.data_ptr(), mu2.data_ptr(), rs2.data_ptr(), h3.data_ptr()
                L.check(L.lib().av_w2v2_layer_fwd(ctypes.byref(a), ops.stream()), "av_w2v2_layer_fwd")
                if keep_ctx:
                    tr = train[li]
                    saved[li] = dict(h=h, mu1=mu1, rs1=rs1, qkv=qkv, ao=ao, lse=lse, amask=amask, h2=h2, mu2=mu2, rs2=rs2, u=u, gf=gf,
                                     x1=x1 if tr else None, x2=x2 if tr else None, g=g if tr else None)
                h = h3
                continue
            x1, mu1, rs1 = ops.layernorm_fwd(h, self.P(p + "layer_norm.weight").data, self.P(p + "layer_norm.bias").data,
                                             out_dtype=dtype, eps=eps, save_stats=True)
            qkv = ops.linear(x1, self.qkv_w(li, dtype), self.qkv_b(li), out_dtype=dtype).view(B, T, 3, nh, hd)
            if amask is not None and amask_evt is not None:
                evts, last_evt = amask_evt
                nxt = min((l for l in evts if l >= li), default=None)           # the first recorded event at or after this layer's mask
                torch.cuda.current_stream(dev).wait_event(evts.pop(nxt) if nxt is not None else last_evt)
                for l in [l for l in evts if l < li]:
                    evts.pop(l)
                if nxt is None:
                    amask_evt = None
            ao, lse = ops.attention_fwd(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], klen, scale, need_lse=keep_ctx,
                                        drop=(at_p, seed, li * 8 + 3), drop_mask=amask)
            h2 = ops.linear(ao.view(B, T, Hd), self.c(p + "attention.out_proj.weight", dtype), self.P(p + "attention.out_proj.bias").data,
                            out_dtype=torch.float32, R=h, drop=(hd_p, seed, li * 8 + 0))
            x2, mu2, rs2 = ops.layernorm_fwd(h2, self.P(p + "final_layer_norm.weight").data, self.P(p + "final_layer_norm.bias").data,
                                             out_dtype=dtype, eps=eps, save_stats=True)
            u = torch.empty((B, T, cfg["intermediate_size"]), dtype=dtype, device=dev) if keep_ctx else None
            # bf16: the saved tensor is the site's gradient factor gelu'(u) o mask / (1 - p) instead of u, so the dX product of the
            # backward ends in one multiply (no erf / exp / mask regeneration while its matrix pipe waits)
            gf = FFN_GF and keep_ctx and is_lp(dtype)
            g = ops.linear(x2, self.c(p + "feed_forward.intermediate_dense.weight", dtype),
                           self.P(p + "feed_forward.intermediate_dense.bias").data, out_dtype=dtype, act=L.ACT_GELU_GF if gf else L.ACT_GELU, C2=u,
                           drop=(ac_p, seed, li * 8 + 1))
            h3 = ops.linear(g, self.c(p + "feed_forward.output_dense.weight", dtype), self.P(p + "feed_forward.output_dense.bias").data,
                            out_dtype=torch.float32, R=h2, drop=(hd_p, seed, li * 8 + 2))
            if keep_ctx:
                tr = train[li]
                saved[li] = dict(h=h, mu1=mu1, rs1=rs1, qkv=qkv, ao=ao, lse=lse, amask=amask, h2=h2, mu2=mu2, rs2=rs2, u=u, gf=gf,
                                 x1=x1 if tr else None, x2=x2 if tr else None, g=g if tr else None)
            h = h3
        last, muf, rsf = ops.layernorm_fwd(h, self.P("encoder.layer_norm.weight").data, self.P("encoder.layer_norm.bias").data,
                                           out_dtype=torch.float32, eps=eps, save_stats=True)
        if mid is None:
            mid = torch.zeros_like(last)
        ctx = None
        if save and first < nl:
            ctx = dict(saved=saved, hL=h, muf=muf, rsf=rsf, klen=klen, first=first, train=train, B=B, T=T, dtype=dtype,
                       seed=seed, hd_p=hd_p, at_p=at_p, ac_p=ac_p)
        return last, mid, ctx

    # ---- backward --------------------------------------------------------------------------------------------
    def backward(self, ctx: dict, dlast: Optional[Tensor], dmid: Optional[Tensor]) -> Dict[str, Tensor]:
        """Gradients of the trainable encoder-layer parameters given d(last_hidden_state) and d(mid)."""
        return self.backward_multi([ctx], [dlast], [dmid])

    def backward_multi(self, ctxs: List[dict], dlasts: List[Optional[Tensor]], dmids: List[Optional[Tensor]]) -> Dict[str, Tensor]:
        """Backward of several forward passes over the SAME parameters (the reference's two audio passes per step, model/trainer.py:94-95)
        walked layer by layer: for every layer the passes run one after the other and the later ones ACCUMULATE their weight gradients into
        the first one's buffers (GEMM / column-sum accumulate forms: no separate gradient-sum pass), so that under data parallelism a layer's
        gradients leave in ONE bucket per step instead of one per pass."""
        cfg = self.cfg
        nl = cfg["num_hidden_layers"]
        Hd = cfg["hidden_size"]
        grads: Dict[str, Tensor] = {}
        states = []
        for ctx, dlast, dmid in zip(ctxs, dlasts, dmids):
            B, T = ctx["B"], ctx["T"]
            dev = ctx["hL"].device
            if dlast is not None:
                dh = ops.layernorm_bwd(ctx["hL"], dlast.contiguous().float(), self.P("encoder.layer_norm.weight").data, ctx["muf"], ctx["rsf"])
            else:
                dh = torch.zeros((B, T, Hd), dtype=torch.float32, device=dev)
            # dh_lp: bf16 copy of dh (with the hidden-dropout mask of the consuming site folded in) when it is current
            states.append(dict(dh=dh, dh_lp=None, dmid=dmid.contiguous().float() if dmid is not None else None))
        first = min(c["first"] for c in ctxs)
        dev = ctxs[0]["hL"].device
        # Two passes: the second one walks the layers on its own stream, ONE LAYER BEHIND the first (its weight-gradient products accumulate
        # into the buffers the first pass creates for that layer), so that the HBM-bound row kernels of one pass overlap the GEMMs of the other
        two_streams = PASS_STREAMS and len(ctxs) == 2 and dev.type == "cuda"
        if two_streams:
            main = torch.cuda.current_stream(dev)
            if getattr(self, "_pass_stream", None) is None:
                self._pass_stream = torch.cuda.Stream(device=dev)
            side = self._pass_stream
            side.wait_stream(main)                                  # d(last) / d(mid) of pass 2 and its final-LayerNorm backward were enqueued on main
            for t in _tensors_of(states[1]):
                t.record_stream(side)
        for li in range(nl - 1, first - 1, -1):
            p = f"encoder.layers.{li}."
            for i, (ctx, st) in enumerate(zip(ctxs, states)):
                if li < ctx["first"]:
                    continue
                if two_streams and i == 1:
                    done0 = torch.cuda.Event(); done0.record(main)   # pass 1 has written this layer's gradient buffers
                    side.wait_event(done0)
                    with torch.cuda.stream(side):
                        before = set(grads)
                        self._layer_backward(ctx, st, li, grads)
                    for k in grads:                                 # allocator bookkeeping across the two streams
                        if k.startswith(p):
                            grads[k].record_stream(side if k in before else main)
                else:
                    self._layer_backward(ctx, st, li, grads)
            if self.grad_ready is not None:                         # the layer's gradients (all passes) go out while the next layer's backward runs
                keys = [k for k in grads if k.startswith(p)]
                ar = self._arenas.get(li)
                # Only a layer that HAS gradients synchronises anything, and then it is the collective's stream that waits for the second
                # pass - not the main stream: with `main.wait_stream(side)` after every layer (the round-3 form) the two passes' backward ran
                # in lockstep under data parallelism (the one-rank RCCL leg: 67.5 ms per step against 64.0 without a reducer)
                if keys and ar is not None and self.grad_flat_ready is not None and all(ar.owns(grads[k]) for k in keys):
                    self.grad_flat_ready(ar, (side,) if two_streams else ())      # the layer's gradients ARE the bucket: reduced in place
                elif keys:
                    if two_streams:
                        main.wait_stream(side)                      # packing path (first step: layout still unknown): the copy runs on main
                    for k, v in zip(keys, self.grad_ready([grads[k] for k in keys])):
                        grads[k] = v
        if two_streams:
            main.wait_stream(side)
            for t in _tensors_of(states[1]):
                t.record_stream(main)
        for ar in self._arenas.values():
            ar.finalize()                                           # first backward: the layout is known now, the buffers exist from the next step on
        if self.grad_wait is not None:
            self.grad_wait()
        return grads

    def begin_grad_step(self) -> None:
        """Called by the trainer when a step's gradients are cleared: the flat gradient buckets may be written again."""
        for ar in self._arenas.values():
            ar.begin_step()

    def _layer_backward(self, ctx: dict, st: dict, li: int, grads: Dict[str, Tensor]) -> None:
        """One encoder layer of one pass: updates st['dh'] (gradient of the residual stream below the layer) and adds the layer's weight
        gradients to ``grads`` (created on first use, accumulated in place afterwards)."""
        cfg = self.cfg
        dtype = ctx["dtype"]
        B, T = ctx["B"], ctx["T"]
        Hd, nh, I = cfg["hidden_size"], cfg["num_attention_heads"], cfg["intermediate_size"]
        hd = Hd // nh
        nl = cfg["num_hidden_layers"]
        scale = hd ** -0.5
        fuse_lp = is_lp(dtype)
        dh, dh_lp, dmid_c = st["dh"], st["dh_lp"], st["dmid"]
        if dmid_c is not None and 6 <= li + 1 <= 9 and li + 1 < nl:
            ops.axpby(0.25, dmid_c, 1.0, dh)
            dh_lp = None                                             # dh changed after its bf16 copy was written
        s = ctx["saved"][li]
        if isinstance(s, str):                                       # LayerDrop skipped this layer: identity
            st["dh_lp"] = dh_lp
            return
        tr = ctx["train"][li]
        p = f"encoder.layers.{li}."
        M = B * T
        seed, hd_p, at_p, ac_p = ctx["seed"], ctx["hd_p"], ctx["at_p"], ctx["ac_p"]

        ar = self._arenas.get(li)
        if ar is None and tr:
            from ..parallel.dp import GradArena
            ar = self._arenas[li] = GradArena()
        dev_ = ctx["hL"].device

        if (not tr and fuse_lp and NATIVE_LAYER and (NATIVE_LAYER >= 2 or M <= NATIVE_MAX_ROWS) and dev_.type == "cuda"
                and ops.GemmProbe.active is None and ops.AttnProbe.active is None):
            # a layer whose weights take no gradient: its eight launches from ONE native call (csrc/w2v2_layer.hip: av_w2v2_layer_bwd_dx)
            wt = [ops.transpose_cached(w) for w in (self.c(p + "feed_forward.output_dense.weight", dtype), self.c(p + "feed_forward.intermediate_dense.weight", dtype),
                                                    self.c(p + "attention.out_proj.weight", dtype), self.qkv_w(li, dtype))]
            if all(w is not None for w in wt):
                lower = li - 1
                lp_ok = hd_p == 0 or (lower >= ctx["first"] and not isinstance(ctx["saved"][lower], str) and not (dmid_c is not None and 6 <= lower + 1 <= 9))
                lp_in = dh_lp if dh_lp is not None else None
                e16 = lambda *shape: torch.empty(shape, dtype=dtype, device=dev_)
                dh3_t = None if lp_in is not None else e16(M, Hd)
                du, dx2, dh2_lp, dao, dqkv, dx1 = e16(M, I), e16(M, Hd), e16(M, Hd), e16(M, Hd), e16(B, T, 3, nh, hd), e16(M, Hd)
                dh2 = torch.empty((B, T, Hd), dtype=torch.float32, device=dev_); dh_new = torch.empty((B, T, Hd), dtype=torch.float32, device=dev_)
                delta = torch.empty((B, nh, T), dtype=torch.float32, device=dev_)
                dh_new_lp = e16(B, T, Hd) if lp_ok else None
                a = getattr(self, "_lbargs", None)
                if a is None:
                    a = self._lbargs = L.W2v2LayerBwdArgs()
                a.B, a.T, a.hidden, a.heads, a.inter, a.lp, a.gf, a.stream_base, a.lower_stream = B, T, Hd, nh, I, L.AV_BF16, int(bool(s.get("gf"))), li * 8, lower * 8 + 2
                a.scale, a.hd_p, a.at_p, a.ac_p, a.seed = scale, hd_p, at_p, ac_p, seed
                a.ln1_g = self.P(p + "layer_norm.weight").data.data_ptr(); a.ln2_g = self.P(p + "final_layer_norm.weight").data.data_ptr()
                a.w_2t, a.w_1t, a.w_ot, a.w_qkvt = (w.data_ptr() for w in wt)
                a.dh, a.h, a.mu1, a.rs1, a.lse = dh.data_ptr(), s["h"].data_ptr(), s["mu1"].data_ptr(), s["rs1"].data_ptr(), s["lse"].data_ptr()
                a.h2, a.mu2, a.rs2 = s["h2"].data_ptr(), s["mu2"].data_ptr(), s["rs2"].data_ptr()
                a.dh_lp, a.qkv, a.ao, a.amask, a.u = ops.ptr(lp_in), s["qkv"].data_ptr(), s["ao"].data_ptr(), ops.ptr(s["amask"]), s["u"].data_ptr()
                a.klen = ops.ptr(ctx["klen"])
                a.dh3_t, a.du, a.dx2, a.dh2_lp, a.dao, a.dqkv, a.dx1, a.dh_out_lp = (ops.ptr(dh3_t), du.data_ptr(), dx2.data_ptr(), dh2_lp.data_ptr(), dao.data_ptr(),
                                                                                   dqkv.data_ptr(), dx1.data_ptr(), ops.ptr(dh_new_lp))
                a.dh2, a.delta, a.dh_out = dh2.data_ptr(), delta.data_ptr(), dh_new.data_ptr()
                L.check(L.lib().av_w2v2_layer_bwd_dx(ctypes.byref(a), ops.stream()), "av_w2v2_layer_bwd_dx")
                ctx["saved"][li] = None
                st["dh"], st["dh_lp"] = dh_new, dh_new_lp
                return

        def wgrad(key, dy, x):                                       # dW (+)= dy^T x; the first writer of a step writes into the layer's flat bucket
            if key in grads:
                ops.matmul_tn(dy, x, out=grads[key], accumulate=True)
            else:
                grads[key] = ops.matmul_tn(dy, x, out=ar.out(key, (dy.shape[1], x.shape[1]), dev_))

        def bgrad(key, dy):                                          # db (+)= column sums
            if key in grads:
                ops.colsum(dy, out=grads[key], accumulate=True)
            else:
                grads[key] = ops.colsum_into(dy, ar.out(key, (dy.shape[-1],), dev_, vec=True))

        def gb_target(key):                                          # packed LayerNorm gradients: an earlier pass's tensor, else a zeroed bucket view
            hit = grads.get(key)
            return hit if hit is not None else ar.out(key, (2 * Hd,), dev_, vec=True)

        dh3 = dh
        dh3_t = dh_lp if (fuse_lp and dh_lp is not None) else ops.cast_dropout(dh3, dtype, (hd_p, seed, li * 8 + 2))   # FFN-output dropout mask
        W2 = self.c(p + "feed_forward.output_dense.weight", dtype)            # [Hd, I]
        if s.get("gf"):                                              # s["u"] holds gelu'(u) o mask / (1 - p)
            du = ops.matmul_nn(dh3_t.view(M, Hd), W2, out_dtype=dtype, act=L.ACT_MUL_AUX, aux=s["u"].view(M, I), b_is_weight=True)
        else:
            du = ops.matmul_nn(dh3_t.view(M, Hd), W2, out_dtype=dtype, act=L.ACT_MUL_GELU_GRAD, aux=s["u"].view(M, I), b_is_weight=True,
                               drop=(ac_p, seed, li * 8 + 1))
        if tr:
            wgrad(p + "feed_forward.output_dense.weight", dh3_t.view(M, Hd), s["g"].view(M, I))
            bgrad(p + "feed_forward.output_dense.bias", (dh3_t if hd_p > 0 else dh3).view(M, Hd))
        W1 = self.c(p + "feed_forward.intermediate_dense.weight", dtype)      # [I, Hd]
        dx2 = ops.matmul_nn(du, W1, out_dtype=dtype, b_is_weight=True)
        if tr:
            wgrad(p + "feed_forward.intermediate_dense.weight", du, s["x2"].view(M, Hd))
            bgrad(p + "feed_forward.intermediate_dense.bias", du)
        ln2 = p + "final_layer_norm."
        r = ops.layernorm_bwd(s["h2"], dx2.view(B, T, Hd), self.P(ln2 + "weight").data, s["mu2"], s["rs2"], dres=dh3,
                              want_param_grads=tr, lp_copy=fuse_lp, lp_drop=(hd_p, seed, li * 8 + 0),   # consumer: this layer's attention-output dropout
                              gb_acc=gb_target(ln2 + "_gb") if tr else None, packed_gb=True)
        dh2_lp = None
        if fuse_lp:
            r, dh2_lp = r[:-1], r[-1]
            r = r if tr else r[0]
        if tr:
            dh2, gb = r
            grads[ln2 + "_gb"] = gb
        else:
            dh2 = r
        dh2_t = dh2_lp if dh2_lp is not None else ops.cast_dropout(dh2, dtype, (hd_p, seed, li * 8 + 0))
        Wo = self.c(p + "attention.out_proj.weight", dtype)
        dao = ops.matmul_nn(dh2_t.view(M, Hd), Wo, out_dtype=dtype, b_is_weight=True).view(B, T, nh, hd)
        if tr:
            wgrad(p + "attention.out_proj.weight", dh2_t.view(M, Hd), s["ao"].view(M, Hd))
            bgrad(p + "attention.out_proj.bias", (dh2_t if hd_p > 0 else dh2).view(M, Hd))
        qkv = s["qkv"]
        dqkv = torch.empty_like(qkv)
        ops.attention_bwd(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], dao, dqkv[:, :, 0], dqkv[:, :, 1], dqkv[:, :, 2], ctx["klen"], scale,
                          o=s["ao"], lse=s["lse"], drop=(at_p, seed, li * 8 + 3), drop_mask=s["amask"])
        dx1 = ops.matmul_nn(dqkv.view(M, 3 * Hd), self.qkv_w(li, dtype), out_dtype=dtype, b_is_weight=True)
        if tr:
            wgrad(p + "attention._qkv_w", dqkv.view(M, 3 * Hd), s["x1"].view(M, Hd))      # packed [3 Hd, Hd]: split into q / k / v at the end
            bgrad(p + "attention._qkv_b", dqkv.view(M, 3 * Hd))
        # the copy of dh serves the FFN-output dropout site of the layer below - unless that layer was dropped (LayerDrop: its site
        # never ran, dh passes through to another site) or dmid is added to dh first (both only with dropout on)
        lower = li - 1
        lp_ok = fuse_lp and (hd_p == 0 or (lower >= ctx["first"] and not isinstance(ctx["saved"][lower], str)
                                          and not (dmid_c is not None and 6 <= lower + 1 <= 9)))
        ln1 = p + "layer_norm."
        r = ops.layernorm_bwd(s["h"], dx1.view(B, T, Hd), self.P(ln1 + "weight").data, s["mu1"], s["rs1"], dres=dh2,
                              want_param_grads=tr, lp_copy=lp_ok, lp_drop=(hd_p, seed, lower * 8 + 2),
                              gb_acc=gb_target(ln1 + "_gb") if tr else None, packed_gb=True)
        dh_lp = None
        if lp_ok:
            r, dh_lp = r[:-1], r[-1]
            r = r if tr else r[0]
        if tr:
            dh, gb = r
            grads[ln1 + "_gb"] = gb
        else:
            dh = r
        ctx["saved"][li] = None            # free as we go
        st["dh"], st["dh_lp"] = dh, dh_lp

    @staticmethod
    def unpack_grads(grads: Dict[str, Tensor], Hd: int) -> Dict[str, Tensor]:
        """Packed accumulators of _layer_backward -> HF parameter names (views, no copies)."""
        out: Dict[str, Tensor] = {}
        for k, v in grads.items():
            if k.endswith("._qkv_w") or k.endswith("._qkv_b"):
                base, kind = k[: -len("_qkv_w")], ("weight" if k.endswith("w") else "bias")
                for j, n in enumerate(("q", "k", "v")):
                    out[f"{base}{n}_proj.{kind}"] = v[j * Hd:(j + 1) * Hd]
            elif k.endswith("._gb"):
                base = k[: -len("_gb")]
                out[base + "weight"], out[base + "bias"] = v[:Hd], v[Hd:]
            else:
                out[k] = v
        return out


def _tensors_of(obj):
    """All CUDA tensors inside nested tuples / lists / dicts."""
    if torch.is_tensor(obj):
        if obj.is_cuda:
            yield obj
    elif isinstance(obj, dict):
        for v in obj.values():
            yield from _tensors_of(v)
    elif isinstance(obj, (list, tuple)):
        for v in obj:
            yield from _tensors_of(v)


class _EncodeFn(torch.autograd.Function):
    """Autograd boundary: (wav, masks of 1 or 2 passes, trainable params...) -> (last, mid) per pass.  Two passes = the reference's
    audio_encoder(audio, mask1) / audio_encoder(audio, mask2) (model/trainer.py:94-95) as ONE node: their backward runs interleaved per layer."""

    @staticmethod
    def forward(fctx, model: Wav2Vec2ModelHIP, wav, masks, valid, names, *params):
        outs, ctxs = [], []
        fc = getattr(model, "_feat_cache", None)
        own_cache = len(masks) == 2 and fc is None           # both passes read the same waveform: one conv feature-extractor run
        if own_cache:
            model._feat_cache = fc = {}
        two_streams = PASS_STREAMS and len(masks) == 2 and wav.is_cuda
        if two_streams:
            dev = wav.device
            main = torch.cuda.current_stream(dev)
            if getattr(model, "_pass_stream", None) is None:
                model._pass_stream = torch.cuda.Stream(device=dev)
            side = model._pass_stream
            model.warm_caches(compute_dtype())                       # cache (re)builds happen HERE, on main, ahead of the fork
            start = torch.cuda.Event(); start.record(main)           # everything enqueued before this forward (weights of the last Adam step ...)
        for i, (am, vl) in enumerate(zip(masks, valid)):
            if two_streams and i == 1:
                # pass 2 on its own stream: it needs the shared conv features of pass 1 (event recorded right behind them), nothing else of it
                side.wait_event(start)
                if fc.get("evt") is not None:
                    side.wait_event(fc["evt"])
                with torch.cuda.stream(side):
                    last, mid, ctx = model.encode(wav, am, save=True, valid_lengths=vl)
                main.wait_stream(side)
                # allocator bookkeeping: blocks of the side pool that the main stream reads later (outputs, everything saved for the backward),
                # and main-pool blocks the side stream read
                for t in _tensors_of((last, mid, ctx)):
                    t.record_stream(main)
                for t in _tensors_of((wav, am, fc.get("feats"))):
                    t.record_stream(side)
            else:
                last, mid, ctx = model.encode(wav, am, save=True, valid_lengths=vl)
            outs += [last, mid]
            ctxs.append(ctx)
        if own_cache:
            model._feat_cache = None
        fctx.model, fctx.ctxs, fctx.names = model, ctxs, names
        return tuple(outs)

    @staticmethod
    def backward(fctx, *douts):
        live = [(c, douts[2 * i], douts[2 * i + 1]) for i, c in enumerate(fctx.ctxs) if c is not None]
        if not live:
            return (None, None, None, None, None) + tuple(None for _ in fctx.names)
        if fctx.model.grad_pre is not None:
            fctx.model.grad_pre()
        g = fctx.model.backward_multi([c for c, _, _ in live], [a for _, a, _ in live], [b for _, _, b in live])
        g = fctx.model.unpack_grads(g, fctx.model.cfg["hidden_size"])
        fctx.ctxs = None
        return (None, None, None, None, None) + tuple(g.get(n) for n in fctx.names)


def w2v2_apply(model: Wav2Vec2ModelHIP, wav: Tensor, attention_mask, second_mask=None, two_passes: bool = False, valid_lengths=None):
    """One pass -> (last, mid); ``two_passes`` -> (last1, mid1, last2, mid2) for (attention_mask, second_mask).  ``valid_lengths``: optional
    host-side sample counts per item, one list per pass (see Wav2Vec2ModelHIP.encode)."""
    masks = (attention_mask, second_mask) if two_passes else (attention_mask,)
    valid = tuple(valid_lengths) if valid_lengths is not None else (None,) * len(masks)
    if getattr(model, "_np", None) is None:
        model._np = list(model.named_parameters())
    flags = tuple(p.requires_grad for _, p in model._np)
    if torch.is_grad_enabled() and any(flags):
        if getattr(model, "_np_flags", None) != flags:          # the freeze policy changed: re-validate, rebuild the lists
            model.check_freeze_policy()
            model._np_flags = flags
            model._np_train = ([n for (n, p), f in zip(model._np, flags) if f], [p for (n, p), f in zip(model._np, flags) if f])
        names, params = model._np_train
        return _EncodeFn.apply(model, wav, masks, valid, names, *params)
    outs = []
    with torch.no_grad():
        for am, vl in zip(masks, valid):
            last, mid, _ = model.encode(wav, am, save=False, valid_lengths=vl)
            outs += [last, mid]
    return tuple(outs)


def load_local_config(path: str) -> dict:
    """``config.json`` of a LOCAL HF wav2vec2 directory (no network is ever touched)."""
    with open(os.path.join(path, "config.json"), "r", encoding="utf-8") as f:
        c = json.load(f)
    if c.get("feat_extract_norm", "layer") != "layer" or not c.get("do_stable_layer_norm", True):
        raise NotImplementedError("only the stable-layer-norm / feat_extract_norm='layer' (XLSR) architecture is built")
    keys = ("hidden_size", "num_hidden_layers", "num_attention_heads", "intermediate_size", "conv_dim", "conv_kernel", "conv_stride",
            "num_conv_pos_embeddings", "num_conv_pos_embedding_groups", "layer_norm_eps")
    out = {k: (tuple(c[k]) if isinstance(c[k], list) else c[k]) for k in keys}
    hf_defaults = dict(hidden_dropout=0.1, attention_dropout=0.1, activation_dropout=0.1, feat_proj_dropout=0.0, layerdrop=0.1,
                       mask_time_prob=0.05, mask_time_length=10, mask_time_min_masks=2, mask_feature_prob=0.0, mask_feature_length=10,
                       mask_feature_min_masks=0, apply_spec_augment=True)
    for k, v in hf_defaults.items():          # a real checkpoint trains with its own regularisation settings
        out[k] = c.get(k, v)
    out["conv_bias"] = bool(c.get("conv_bias", False))       # HF default False; XLSR-53 checkpoints have True
    return out


_DROP_PREFIXES = ("lm_head.", "quantizer.", "project_hid.", "project_q.", "dropout_features.")


def remap_hf_keys(sd: Dict[str, Tensor]) -> Dict[str, Tensor]:
    """State-dict keys of a local HF wav2vec2 checkpoint -> the ``Wav2Vec2Model`` names this module uses, the way
    ``from_pretrained`` resolves them (model/encoder.py:83 loads a ``Wav2Vec2ForCTC`` fine-tune into ``Wav2Vec2Model``):
    the ``wav2vec2.`` prefix of task heads is stripped, head / pre-training tensors (``lm_head.*``, ``quantizer.*``,
    ``project_hid.*``, ``project_q.*``) are dropped, and the legacy weight-norm names ``weight_g`` / ``weight_v`` of the
    positional convolution become ``parametrizations.weight.original0`` / ``original1``."""
    out: Dict[str, Tensor] = {}
    for k, v in sd.items():
        if k.startswith("wav2vec2."):
            k = k[len("wav2vec2."):]
        if k.startswith(_DROP_PREFIXES):
            continue
        if k.endswith("pos_conv_embed.conv.weight_g"):
            k = k[:-len("weight_g")] + "parametrizations.weight.original0"
        elif k.endswith("pos_conv_embed.conv.weight_v"):
            k = k[:-len("weight_v")] + "parametrizations.weight.original1"
        out[k] = v
    return out


def load_local_weights(path: str) -> Optional[Dict[str, Tensor]]:
    """Tensors of a LOCAL HF directory: ``model.safetensors`` or ``pytorch_model.bin`` (``weights_only=True``: nothing from the
    file is executed).  None if the directory holds neither."""
    st = os.path.join(path, "model.safetensors")
    if os.path.exists(st):
        from safetensors.torch import load_file
        return remap_hf_keys(load_file(st))
    pt = os.path.join(path, "pytorch_model.bin")
    if os.path.exists(pt):
        return remap_hf_keys(torch.load(pt, map_location="cpu", weights_only=True))
    return None
