"""CrossAttentionFusion on the HIP kernels — same constructor / forward contract and state_dict keys as the
reference (model/fusion_module.py:5-67): speech-frame gather -> linear resample to T_v -> two projections ->
ONE cross-attention (audio queries, visual keys/values, 4 heads) -> Linear -> 2-layer BiLSTM(512).

torch.nn modules are used ONLY as parameter containers (names + default init); their forward is never called.
Forward and backward are explicit kernel sequences; the data-dependent gather/pad/interpolate and the CTC
``input_lengths`` are computed on the device without host syncs.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch
import torch.nn as nn

from .. import _lib as L
from .. import ops
from ..precision import compute_dtype, is_lp
from ..utils.shadow import ParamCache

Tensor = torch.Tensor
# one persistent launch per BiLSTM layer (bf16, H = 512) instead of one launch per time step; AVAMD_LSTM_PERSISTENT=0 disables
import os as _os
PERSISTENT_LSTM = _os.environ.get("AVAMD_LSTM_PERSISTENT", "1") != "0"
# in-projection + attention core of the cross-attention in one kernel (bf16, embed 512 = 4 x 128, T_v <= 112); 0 = separate GEMMs + attention
FUSED_XATTN = _os.environ.get("AVAMD_FUSED_XATTN", "1") != "0"


def lstm_forward(mod: "CrossAttentionFusion", x_tm: Tensor, save: bool):
    """x_tm [T,B,E] compute dtype -> (out_bt [B,T,2H] compute dtype, ctx)."""
    lstm = mod.temporal_model
    dtype = x_tm.dtype
    T, B, _ = x_tm.shape
    H = lstm.hidden_size
    dev = x_tm.device
    layers = []
    inp = x_tm
    out_bt = None
    for layer in range(lstm.num_layers):
        names = [f"weight_ih_l{layer}", f"weight_ih_l{layer}_reverse", f"weight_hh_l{layer}", f"weight_hh_l{layer}_reverse",
                 f"bias_ih_l{layer}", f"bias_ih_l{layer}_reverse", f"bias_hh_l{layer}", f"bias_hh_l{layer}_reverse"]
        P = [getattr(lstm, n) for n in names]
        wih = mod._cache.get(("wih", layer), P[0:2], dtype, lambda: ops.cast(torch.cat([P[0].data, P[1].data], 0).contiguous(), dtype), flat=True)
        whh = mod._cache.get(("whh", layer), P[2:4], dtype, lambda: ops.cast(torch.stack([P[2].data, P[3].data], 0).contiguous(), dtype), flat=True)
        bias = mod._cache.get(("b", layer), P[4:8], torch.float32,
                              lambda: torch.cat([P[4].data + P[6].data, P[5].data + P[7].data], 0).contiguous())
        gx = ops.linear(inp, wih, bias, out_dtype=torch.float32)                         # [T,B,8H] = [T,B,2,4H]
        hseq = torch.empty((T, B, 2 * H), dtype=dtype, device=dev)
        cseq = torch.empty((T, B, 2, H), dtype=torch.float32, device=dev)
        gates = torch.empty((T, B, 2, 4 * H), dtype=dtype, device=dev) if save else None
        last = layer == lstm.num_layers - 1
        if last:
            out_bt = torch.empty((B, T, 2 * H), dtype=dtype, device=dev)
        st = ops.stream()
        if PERSISTENT_LSTM and is_lp(dtype) and H == 512:
            cnt = torch.empty(L.LSTM_COUNTER_INTS, dtype=torch.int32, device=dev)
            L.check(L.lib().av_lstm_fwd_layer(ops.ptr(gx), ops.ptr(whh), ops.ptr(hseq), ops.ptr(cseq), ops.ptr(gates),
                                              ops.ptr(out_bt) if last else None, ops.ptr(cnt), T, B, H, st), "av_lstm_fwd_layer")
            mod._lstm_flags.append(cnt)
        else:
            fn = L.lib().av_lstm_fwd_step
            for s in range(T):
                L.check(fn(ops.ptr(gx), ops.ptr(whh), ops.ptr(hseq), ops.ptr(cseq), ops.ptr(gates), ops.ptr(out_bt) if last else None,
                           ops.dt(hseq), T, B, H, s, st), "av_lstm_fwd_step")
        if save:
            layers.append(dict(inp=inp, hseq=hseq, cseq=cseq, gates=gates, wih=wih, whh=whh, names=names))
        inp = hseq
    return out_bt, (layers if save else None)


def lstm_backward(mod: "CrossAttentionFusion", layers, dout_bt: Tensor, grads: Dict[str, Tensor]) -> Tensor:
    """dout_bt [B,T,2H] fp32 -> d(x_tm) [T,B,E] compute dtype; fills ``grads`` with temporal_model.* gradients (written into the module's
    flat gradient bucket when it has one: ``mod._grad_out``)."""
    lstm = mod.temporal_model
    A = mod._grad_out
    H = lstm.hidden_size
    dout, do_bs, do_ts = dout_bt, dout_bt.shape[1] * 2 * H, 2 * H
    dx = None
    for layer in range(lstm.num_layers - 1, -1, -1):
        c = layers[layer]
        hseq, gates, cseq, inp = c["hseq"], c["gates"], c["cseq"], c["inp"]
        T, B, _ = hseq.shape
        dtype = hseq.dtype
        dev = hseq.device
        whhT = ops.cast(c["whh"].transpose(1, 2).contiguous(), dtype)                    # [2][H][4H]
        dgates = torch.empty((T, B, 2, 4 * H), dtype=dtype, device=dev)
        dc = torch.empty((2, B, H), dtype=torch.float32, device=dev)
        st = ops.stream()
        if PERSISTENT_LSTM and is_lp(dtype) and H == 512:
            cnt = torch.empty(L.LSTM_COUNTER_INTS, dtype=torch.int32, device=dev)
            L.check(L.lib().av_lstm_bwd_layer(ops.ptr(dout), ops.dt(dout), do_bs, do_ts, ops.ptr(dgates), ops.ptr(whhT), ops.ptr(gates),
                                              ops.ptr(cseq), ops.ptr(dc), ops.ptr(cnt), T, B, H, st), "av_lstm_bwd_layer")
            mod._lstm_flags.append(cnt)
        else:
            fn = L.lib().av_lstm_bwd_step
            for s in range(T):
                L.check(fn(ops.ptr(dout), ops.dt(dout), do_bs, do_ts, ops.ptr(dgates), ops.ptr(whhT), ops.ptr(gates), ops.ptr(cseq),
                           ops.ptr(dc), ops.dt(dgates), T, B, H, s, st), "av_lstm_bwd_step")
        M = T * B
        dg2 = dgates.view(M, 8 * H)
        in_f = inp.shape[-1]
        dwih = ops.matmul_tn(dg2, inp.view(M, in_f), out=A(f"lstm.dwih{layer}", (8 * H, in_f), dev))     # [8H, in]
        db = ops.colsum_into(dg2, A(f"lstm.db{layer}", (8 * H,), dev, True))             # [8H]
        n = c["names"]
        grads["temporal_model." + n[0]] = dwih[:4 * H]
        grads["temporal_model." + n[1]] = dwih[4 * H:]
        grads["temporal_model." + n[4]] = db[:4 * H]; grads["temporal_model." + n[6]] = db[:4 * H]       # (separate view objects: each
        grads["temporal_model." + n[5]] = db[4 * H:]; grads["temporal_model." + n[7]] = db[4 * H:]       # is handed to autograd once)
        dwhh = A(f"lstm.dwhh{layer}", (2, 4 * H, H), dev)
        if dwhh is None:
            dwhh = torch.zeros((2, 4 * H, H), dtype=torch.float32, device=dev)
        elif T <= 1:
            dwhh.zero_()
        if T > 1:
            # forward chain: dgates[t] pairs with h[t-1]; reverse chain: dgates[t] pairs with h[t+1].  Time-major buffers
            # make the one-step shift a row offset of B; the column slices are transposed to the K-contiguous fast form.
            h2 = hseq.view(M, 2 * H)
            ops.matmul_tn(dg2[B:, :4 * H], h2[:M - B, :H], out=dwhh[0])
            ops.matmul_tn(dg2[:M - B, 4 * H:], h2[B:, H:], out=dwhh[1])
        grads["temporal_model." + n[2]] = dwhh[0]
        grads["temporal_model." + n[3]] = dwhh[1]
        dx = ops.matmul_nn(dg2, c["wih"], out_dtype=dtype).view(T, B, in_f)              # time-major
        dout, do_bs, do_ts = dx, in_f, B * in_f                                          # next (lower) layer reads time-major
    return dx


class _FusionFn(torch.autograd.Function):
    @staticmethod
    def forward(fctx, mod: "CrossAttentionFusion", save: bool, names, groups: int, visual_feat, audio_feat, mask, *params):
        dtype = compute_dtype()
        dev = audio_feat.device
        E, nh = mod.fused_dim, mod.num_heads
        hd = E // nh
        B, Tv, Dv = visual_feat.shape
        _, Ta, Da = audio_feat.shape
        af = audio_feat.contiguous().float()
        mask = mask.contiguous().long()
        ws = torch.empty(B * Ta + B + groups, dtype=torch.int32, device=dev)
        a_in = torch.empty((B, Tv, Da), dtype=torch.float32, device=dev)
        m_out = torch.empty((B, Tv), dtype=torch.long, device=dev)
        lens = torch.empty((B,), dtype=torch.long, device=dev)
        L.check(L.lib().av_fusion_gather_lerp_fwd(ops.ptr(af), ops.ptr(mask), ops.ptr(ws), ops.ptr(a_in), ops.ptr(m_out), ops.ptr(lens),
                                                  B, Ta, Tv, Da, groups, ops.stream()), "av_fusion_gather_lerp_fwd")
        c = lambda p: mod.cparam(p, dtype)
        vis_t = ops.cast(visual_feat.contiguous().float(), dtype)
        a_in_t = ops.cast(a_in, dtype)
        v = ops.linear(vis_t, c(mod.visual_proj.weight), mod.visual_proj.bias.data)
        a = ops.linear(a_in_t, c(mod.audio_proj.weight), mod.audio_proj.bias.data)
        mha = mod.cross_attn_audio
        Win = c(mha.in_proj_weight)
        bin_ = mha.in_proj_bias.data
        scale = hd ** -0.5                                   # torch scales q by 1/sqrt(hd) (torch:functional.py:6578)
        if FUSED_XATTN and is_lp(dtype) and E == 512 and nh == 4 and Tv <= 112:
            o, q, kv, lse = ops.fusion_xattn_fwd(a, v, Win, bin_, nh, scale, save)
        else:
            q = ops.linear(a, Win[:E], bin_[:E].contiguous()).view(B, Tv, nh, hd)
            kv = ops.linear(v, Win[E:], bin_[E:].contiguous()).view(B, Tv, 2, nh, hd)
            o, lse = ops.attention_fwd(q, kv[:, :, 0], kv[:, :, 1], None, scale, need_lse=save)
        a2v = ops.linear(o.view(B, Tv, E), c(mha.out_proj.weight), mha.out_proj.bias.data)
        fused = ops.linear(a2v, c(mod.fusion_proj.weight), mod.fusion_proj.bias.data)
        x_tm = torch.empty((Tv, B, E), dtype=dtype, device=dev)
        L.check(L.lib().av_permute_bt(ops.ptr(fused), ops.dt(fused), ops.ptr(x_tm), ops.dt(x_tm), B, Tv, E, ops.stream()), "av_permute_bt")
        out_bt, lctx = lstm_forward(mod, x_tm, save)
        out = ops.cast(out_bt, torch.float32)
        if save:
            fctx.saved = dict(ws=ws, vis_t=vis_t, a_in_t=a_in_t, v=v, a=a, q=q, kv=kv, o=o, lse=lse, a2v=a2v, fused=fused, lstm=lctx,
                              shape=(B, Tv, Ta, Da, Dv), dtype=dtype, groups=groups)
        else:
            fctx.saved = None
        fctx.mod, fctx.names = mod, names
        fctx.need_audio = audio_feat.requires_grad
        fctx.need_visual = visual_feat.requires_grad
        fctx.mark_non_differentiable(lens, m_out)
        return out, lens, m_out

    @staticmethod
    def backward(fctx, dout, _dl, _dm):
        mod, s = fctx.mod, fctx.saved
        n_extra = 7
        if s is None:
            return (None,) * (n_extra + len(fctx.names))
        B, Tv, Ta, Da, Dv = s["shape"]
        dtype = s["dtype"]
        E, nh = mod.fused_dim, mod.num_heads
        hd = E // nh
        M = B * Tv
        c = lambda p: mod.cparam(p, dtype)
        g: Dict[str, Tensor] = {}
        dx_tm = lstm_backward(mod, s["lstm"], dout.contiguous().float(), g)               # [Tv,B,E]
        dfused = torch.empty((B, Tv, E), dtype=dtype, device=dout.device)
        L.check(L.lib().av_permute_bt(ops.ptr(dx_tm), ops.dt(dx_tm), ops.ptr(dfused), ops.dt(dfused), Tv, B, E, ops.stream()), "av_permute_bt")
        df2 = dfused.view(M, E)
        dev = dout.device
        A = mod._grad_out
        g["fusion_proj.weight"] = ops.matmul_tn(df2, s["a2v"].view(M, E), out=A("fusion_proj.weight", (E, E), dev))
        g["fusion_proj.bias"] = ops.colsum_into(df2, A("fusion_proj.bias", (E,), dev, True))
        da2v = ops.matmul_nn(df2, c(mod.fusion_proj.weight))
        mha = mod.cross_attn_audio
        g["cross_attn_audio.out_proj.weight"] = ops.matmul_tn(da2v, s["o"].view(M, E), out=A("xa.out_proj.weight", (E, E), dev))
        g["cross_attn_audio.out_proj.bias"] = ops.colsum_into(da2v, A("xa.out_proj.bias", (E,), dev, True))
        do = ops.matmul_nn(da2v, c(mha.out_proj.weight)).view(B, Tv, nh, hd)
        q, kv = s["q"], s["kv"]
        if FUSED_XATTN and is_lp(dtype) and E == 512 and nh == 4 and Tv <= 112:
            dq, dkv = ops.fusion_xattn_bwd(q, kv, s["o"], do.contiguous(), s["lse"], hd ** -0.5)
        else:
            dq = torch.empty_like(q); dkv = torch.empty_like(kv)
            ops.attention_bwd(q, kv[:, :, 0], kv[:, :, 1], do, dq, dkv[:, :, 0], dkv[:, :, 1], None, hd ** -0.5, o=s["o"], lse=s["lse"])
        Win = c(mha.in_proj_weight)
        dq2, dkv2 = dq.view(M, E), dkv.view(M, 2 * E)
        dWin = A("xa.in_proj_weight", (3 * E, E), dev)
        if dWin is None:
            dWin = torch.empty((3 * E, E), dtype=torch.float32, device=dev)
        ops.matmul_tn(dq2, s["a"].view(M, E), out=dWin[:E])
        ops.matmul_tn(dkv2, s["v"].view(M, E), out=dWin[E:])
        g["cross_attn_audio.in_proj_weight"] = dWin
        dbin = A("xa.in_proj_bias", (3 * E,), dev, True)
        zeroed = dbin is not None
        if dbin is None:
            dbin = torch.empty(3 * E, dtype=torch.float32, device=dev)
        ops.colsum(dq2, out=dbin[:E], accumulate=zeroed); ops.colsum(dkv2, out=dbin[E:], accumulate=zeroed)
        g["cross_attn_audio.in_proj_bias"] = dbin
        da = ops.matmul_nn(dq2, Win[:E])
        dv = ops.matmul_nn(dkv2, Win[E:])
        g["audio_proj.weight"] = ops.matmul_tn(da, s["a_in_t"].view(M, Da), out=A("audio_proj.weight", (E, Da), dev))
        g["audio_proj.bias"] = ops.colsum_into(da, A("audio_proj.bias", (E,), dev, True))
        g["visual_proj.weight"] = ops.matmul_tn(dv, s["vis_t"].view(M, Dv), out=A("visual_proj.weight", (E, Dv), dev))
        g["visual_proj.bias"] = ops.colsum_into(dv, A("visual_proj.bias", (E,), dev, True))
        d_audio = d_visual = None
        if fctx.need_audio:
            da_in = ops.matmul_nn(da, c(mod.audio_proj.weight), out_dtype=torch.float32)       # [M, Da]
            d_audio = torch.empty((B, Ta, Da), dtype=torch.float32, device=dout.device)
            L.check(L.lib().av_fusion_gather_lerp_bwd(ops.ptr(da_in), ops.ptr(s["ws"]), ops.ptr(d_audio), B, Ta, Tv, Da, s["groups"], ops.stream()),
                    "av_fusion_gather_lerp_bwd")
        if fctx.need_visual:
            d_visual = ops.matmul_nn(dv, c(mod.visual_proj.weight), out_dtype=torch.float32).view(B, Tv, Dv)
        fctx.saved = None
        return (None, None, None, None, d_visual, d_audio, None) + tuple(g.get(n) for n in fctx.names)


class CrossAttentionFusion(nn.Module):
    def __init__(self, visual_dim, audio_dim, fused_dim, num_heads=4):
        super().__init__()
        self.visual_proj = nn.Linear(visual_dim, fused_dim)
        self.audio_proj = nn.Linear(audio_dim, fused_dim)
        # declared by the reference but never called (model/fusion_module.py:14 vs :61): kept for the checkpoint keys
        self.cross_attn_visual = nn.MultiheadAttention(embed_dim=fused_dim, num_heads=num_heads, batch_first=True)
        self.cross_attn_audio = nn.MultiheadAttention(embed_dim=fused_dim, num_heads=num_heads, batch_first=True)
        self.fusion_proj = nn.Linear(fused_dim, fused_dim)
        self.temporal_model = nn.LSTM(input_size=fused_dim, hidden_size=fused_dim, num_layers=2, batch_first=True, bidirectional=True)
        self.fused_dim, self.num_heads = fused_dim, num_heads
        self._cache = ParamCache()
        self.grad_arena = None         # parallel.dp.GradArena shared with the CTC head (the trainer's decoder + fusion gradient bucket)
        self._lstm_flags = []          # arrival/timeout words of the persistent LSTM launches (checked lazily)
        self._flag_host, self._flag_evt, self._flag_n = None, None, 0
        if fused_dim % 32 or (fused_dim // num_heads) not in (16, 32, 64, 128):
            raise ValueError("fused_dim must be a multiple of 32 with head_dim in {16,32,64,128} for the HIP kernels")

    # ---- timeout words of the persistent BiLSTM launches.  A launch whose workgroups cannot all be resident (a GPU shared with
    # another process that also runs persistent kernels) gives up after a bounded spin and raises counters[2]; its results are
    # invalid.  The trainer stages an asynchronous copy of the pending words into pinned memory right before the step's one
    # synchronising call (F.ctc_loss) and looks at them right after it, so a timeout is reported in the step it happened in
    # (forward launches) or in the next one (backward launches) without adding a synchronisation.
    def stage_flag_check(self):
        if not self._lstm_flags or self._flag_evt is not None:
            return
        n = min(len(self._lstm_flags), 16)
        if self._flag_host is None:
            self._flag_host = torch.zeros(16, dtype=torch.int32).pin_memory()
        words = torch.stack([c[2] for c in self._lstm_flags[:n]])
        self._flag_host[:n].copy_(words, non_blocking=True)
        self._flag_evt = torch.cuda.Event(); self._flag_evt.record()
        self._flag_n = n
        del self._lstm_flags[:n]

    def finish_flag_check(self):
        if self._flag_evt is None or not self._flag_evt.query():
            return
        bad = int(self._flag_host[: self._flag_n].max()) != 0
        self._flag_evt = None
        if bad:
            raise RuntimeError("persistent LSTM kernel: inter-workgroup wait timed out (the step that launched it is invalid); "
                               "AVAMD_LSTM_PERSISTENT=0 selects the per-step kernels (e.g. on a GPU shared with other processes)")

    def _grad_out(self, name: str, shape, device, vec: bool = False):
        """View of the flat gradient bucket for this gradient, or None (no bucket / layout not known yet / already written this step).
        ``vec``: a zeroed view of the bucket's vector zone, to be accumulated into."""
        return self.grad_arena.out("fusion." + name, shape, device, vec) if self.grad_arena is not None else None

    def cparam(self, p: Tensor, dtype) -> Tensor:
        if dtype == torch.float32:
            return p.data
        return self._cache.get(("c", id(p)), [p], dtype, lambda: ops.cast(p.data.contiguous(), dtype), flat=True)

    def forward(self, visual_feat, audio_feat, mask=None, groups: int = 1):
        """visual_feat [B,T_v,D_v], audio_feat [B,T_a,D_a], mask [B,T_a] (0/3 ignore, 1/2 use) ->
        (fused [B,T_v,2*fused_dim], input_lengths int64 [B] on mask.device).
        ``groups`` (extension): the batch holds ``groups`` independent calls stacked along dim 0; the reference's
        "pad to the batch maximum" (:46) is then evaluated per group, so the result equals ``groups`` separate calls."""
        if mask is None:
            raise ValueError("CrossAttentionFusion: mask is required (the reference dereferences it unconditionally, :66)")
        if not audio_feat.is_cuda:
            raise RuntimeError("CrossAttentionFusion (HIP): inputs must be on the GPU; there is no CPU fallback")
        if getattr(self, "_np", None) is None:
            self._np = [(n, p) for n, p in self.named_parameters() if not n.startswith("cross_attn_visual.")]
        names = [n for n, p in self._np if p.requires_grad]
        params = [p for n, p in self._np if p.requires_grad]
        save = torch.is_grad_enabled() and (bool(names) or audio_feat.requires_grad or visual_feat.requires_grad)
        self.finish_flag_check()                # timeout words staged earlier (no synchronisation: pinned copy + event query)
        if len(self._lstm_flags) > 64:          # nobody staged them (module used outside the trainer): blocking check of the oldest
            old, self._lstm_flags = self._lstm_flags[:32], self._lstm_flags[32:]
            if int(torch.stack(old)[:, 2].max()) != 0:
                raise RuntimeError("persistent LSTM kernel: inter-workgroup wait timed out (results of that step are invalid)")
        out, lens, m_out = _FusionFn.apply(self, save, names, groups, visual_feat, audio_feat, mask, *params)
        self.last_mask = m_out
        return out, lens
