"""The reference's earlier model: mel spectrogram + lip frames -> two bidirectional GRU encoders -> Linear -> CTC,
on the HIP kernels.  Same class names, constructor arguments, ``forward`` contract and state_dict keys as
``이전 버전/multimodal_ctc_korean.py:8-55``; ``train_step`` is the body of the loop in ``이전 버전/train_ctc_korea.py:82-109``.

torch.nn modules are parameter containers only (names + default init); forward and backward are explicit kernel sequences:

* lip front-end (:11-17): frames are packed channel-last (3 -> 8 zero-padded channels so that an im2col row is 16-byte granular), the
  two 3x3 convolutions are implicit-im2col MFMA GEMMs with the bias in the epilogue, ReLU + MaxPool2d(2) is one pass (the second one
  writes the (C, H', W') order the reference flattens for the GRU, :25); backward = routed max gradients, dX as a convolution of dY
  with the flipped filter, dW = dY^T im2col(X) in frame chunks;
* GRUs (:19,32): the input projections of all time steps are one GEMM per layer (both directions: N = 6H), the recurrence runs in
  per-step kernels (csrc/legacy.hip, both directions per launch), the weight gradients are GEMMs over all steps (time-major
  buffers: the one-step shift between dgh and h is a row offset);
* the two speakers share the lip encoder and are batched into one pass (no batch-coupled op anywhere in this model, so the
  result equals two calls); ``fc`` on [lip | audio] (:48-52) is one GEMM over both speakers.

There is no CPU fallback: inputs must live on the GPU.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import _lib as L
from .. import ops
from ..precision import compute_dtype
from ..utils.shadow import ParamCache

Tensor = torch.Tensor
CP = 8                     # packed input channels of the first convolution
IM2COL_CHUNK_BYTES = 512 << 20


class LipEncoder(nn.Module):
    def __init__(self, in_channels=3, hidden_dim=256):
        super().__init__()
        self.cnn = nn.Sequential(nn.Conv2d(in_channels, 32, kernel_size=3, padding=1), nn.ReLU(), nn.MaxPool2d(2),
                                 nn.Conv2d(32, 64, kernel_size=3, padding=1), nn.ReLU(), nn.MaxPool2d(2))
        self.rnn = nn.GRU(input_size=64 * 24 * 24, hidden_size=hidden_dim, num_layers=2, batch_first=True, bidirectional=True)

    def forward(self, frames):
        raise RuntimeError("LipEncoder (HIP): called through MultimodalCTCKoreanModel.forward")


class AudioEncoder(nn.Module):
    def __init__(self, input_dim=80, hidden_dim=256):
        super().__init__()
        self.rnn = nn.GRU(input_size=input_dim, hidden_size=hidden_dim, num_layers=2, batch_first=True, bidirectional=True)

    def forward(self, mel):
        raise RuntimeError("AudioEncoder (HIP): called through MultimodalCTCKoreanModel.forward")


# ------------------------------------------------------------------------------------------------------------------- GRU
def _gru_names(layer: int) -> List[str]:
    return [f"weight_ih_l{layer}", f"weight_ih_l{layer}_reverse", f"weight_hh_l{layer}", f"weight_hh_l{layer}_reverse",
            f"bias_ih_l{layer}", f"bias_ih_l{layer}_reverse", f"bias_hh_l{layer}", f"bias_hh_l{layer}_reverse"]


def gru_forward(rnn: nn.GRU, cache: ParamCache, x_tm: Tensor, save: bool):
    """x_tm [T, B, in] compute dtype -> (out_bt [B, T, 2H] compute dtype, ctx)."""
    dtype = x_tm.dtype
    T, B, _ = x_tm.shape
    H = rnn.hidden_size
    dev = x_tm.device
    layers = []
    inp = x_tm
    out_bt = None
    for layer in range(rnn.num_layers):
        names = _gru_names(layer)
        P = [getattr(rnn, n) for n in names]
        wih = cache.get(("wih", layer), P[0:2], dtype, lambda: ops.cast(torch.cat([P[0].data, P[1].data], 0).contiguous(), dtype), flat=True)     # [6H, in]
        whh = cache.get(("whh", layer), P[2:4], dtype, lambda: ops.cast(torch.stack([P[2].data, P[3].data], 0).contiguous(), dtype), flat=True)   # [2, 3H, H]
        bih = cache.get(("bih", layer), P[4:6], torch.float32, lambda: torch.cat([P[4].data, P[5].data], 0).contiguous())
        bhh = cache.get(("bhh", layer), P[6:8], torch.float32, lambda: torch.cat([P[6].data, P[7].data], 0).contiguous())
        gx = ops.linear(inp, wih, bih, out_dtype=torch.float32)                            # [T, B, 6H] = [T, B, 2, 3H]
        hseq = torch.empty((T, B, 2 * H), dtype=dtype, device=dev)
        hf = torch.empty((T, B, 2, H), dtype=torch.float32, device=dev)
        gates = torch.empty((T, B, 2, 4 * H), dtype=torch.float32, device=dev) if save else None
        last = layer == rnn.num_layers - 1
        if last:
            out_bt = torch.empty((B, T, 2 * H), dtype=dtype, device=dev)
        fn, st = L.lib().av_gru_fwd_step, ops.stream()
        for s in range(T):
            L.check(fn(ops.ptr(gx), ops.ptr(whh), ops.ptr(bhh), ops.ptr(hseq), ops.ptr(hf), ops.ptr(gates), ops.ptr(out_bt) if last else None,
                       ops.dt(hseq), T, B, H, s, st), "av_gru_fwd_step")
        if save:
            layers.append(dict(inp=inp, hseq=hseq, hf=hf, gates=gates, wih=wih, whh=whh, names=names))
        inp = hseq
    return out_bt, (layers if save else None)


def gru_backward(rnn: nn.GRU, layers, dout_bt: Tensor, grads: Dict[str, Tensor], prefix: str, need_dx: bool) -> Optional[Tensor]:
    """dout_bt [B, T, 2H] fp32 -> d(x_tm) [T, B, in] compute dtype (None unless ``need_dx``); adds the rnn.* gradients to ``grads``
    (accumulating into entries that exist: the lip encoder's GRU runs once for both speakers, so there is nothing to add there, but the
    contract is the same as for the convolutions)."""
    H = rnn.hidden_size
    dout, do_bs, do_ts = dout_bt, dout_bt.shape[1] * 2 * H, 2 * H
    dx = None
    for layer in range(rnn.num_layers - 1, -1, -1):
        c = layers[layer]
        hseq, hf, gates, inp = c["hseq"], c["hf"], c["gates"], c["inp"]
        T, B, _ = hseq.shape
        dtype, dev = hseq.dtype, hseq.device
        whhT = c["whh"].transpose(1, 2).contiguous()                                       # [2, H, 3H]
        dgi = torch.empty((T, B, 2, 3 * H), dtype=dtype, device=dev)
        dgh = torch.empty((T, B, 2, 3 * H), dtype=dtype, device=dev)
        dhc = torch.empty((2, B, H), dtype=torch.float32, device=dev)
        fn, st = L.lib().av_gru_bwd_step, ops.stream()
        for s in range(T):
            L.check(fn(ops.ptr(dout), ops.dt(dout), do_bs, do_ts, ops.ptr(dgi), ops.ptr(dgh), ops.ptr(whhT), ops.ptr(gates), ops.ptr(hf),
                       ops.ptr(dhc), ops.dt(dgi), T, B, H, s, st), "av_gru_bwd_step")
        M = T * B
        n = c["names"]
        in_f = inp.shape[-1]
        dgi2, dgh2 = dgi.view(M, 6 * H), dgh.view(M, 6 * H)
        dwih = ops.matmul_tn(dgi2, inp.view(M, in_f))                                      # [6H, in]
        dbi, dbh = ops.colsum(dgi2), ops.colsum(dgh2)
        grads[prefix + n[0]] = dwih[:3 * H]; grads[prefix + n[1]] = dwih[3 * H:]
        grads[prefix + n[4]] = dbi[:3 * H]; grads[prefix + n[5]] = dbi[3 * H:]
        grads[prefix + n[6]] = dbh[:3 * H]; grads[prefix + n[7]] = dbh[3 * H:]
        dwhh = torch.zeros((2, 3 * H, H), dtype=torch.float32, device=dev)
        if T > 1:
            # forward chain: dgh[t] pairs with h[t-1]; reverse chain: dgh[t] pairs with h[t+1] (time-major: a row offset of B)
            h2 = hseq.view(M, 2 * H)
            ops.matmul_tn(dgh2[B:, :3 * H], h2[:M - B, :H], out=dwhh[0])
            ops.matmul_tn(dgh2[:M - B, 3 * H:], h2[B:, H:], out=dwhh[1])
        grads[prefix + n[2]] = dwhh[0]; grads[prefix + n[3]] = dwhh[1]
        if layer > 0 or need_dx:
            dx = ops.matmul_nn(dgi2, c["wih"], out_dtype=dtype if layer == 0 else torch.float32, b_is_weight=True).view(T, B, in_f)
        if layer > 0:
            dout, do_bs, do_ts = dx, in_f, B * in_f                                        # time-major gradient of the layer below's output
    return dx if need_dx else None


# ------------------------------------------------------------------------------------------------------------------- model
def _conv3(x: Tensor, N: int, H: int, W: int, Cin: int, w_tap: Tensor, bias: Tensor, dtype) -> Tensor:
    """3x3 / stride 1 / pad 1 convolution of a channel-last image as an implicit-im2col GEMM: [N*H*W, Cout]."""
    Cout = w_tap.shape[0]
    y = torch.empty((N * H * W, Cout), dtype=dtype, device=x.device)
    geo = dict(cT=1, cH=H, cW=W, cCtot=Cin, cCin=Cin, cCoff=0, cKt=1, cKh=3, cKw=3, cSh=1, cSw=1, cPt=0, cPh=1, cPw=1, cOh=H, cOw=W)
    ops.gemm(x, w_tap, y, M=N * H * W, N=Cout, K=9 * Cin, lda=0, ldb=9 * Cin, ldc=Cout, a_mode=L.A_CONV2D, bias=bias, conv=geo)
    return y


def _relu_pool(y: Tensor, N: int, H: int, W: int, C: int, nchw_out: bool) -> Tensor:
    out = torch.empty((N, C * (H // 2) * (W // 2)) if nchw_out else (N, H // 2, W // 2, C), dtype=y.dtype, device=y.device)
    L.check(L.lib().av_relu_maxpool2_fwd(ops.ptr(y), ops.ptr(out), ops.dt(y), N, H, W, C, int(nchw_out), ops.stream()), "av_relu_maxpool2_fwd")
    return out


def _relu_pool_bwd(y: Tensor, dy: Tensor, N: int, H: int, W: int, C: int, nchw_dy: bool) -> Tensor:
    dx = torch.empty((N * H * W, C), dtype=y.dtype, device=y.device)
    L.check(L.lib().av_relu_maxpool2_bwd(ops.ptr(y), ops.ptr(dy), ops.ptr(dx), ops.dt(y), N, H, W, C, int(nchw_dy), ops.stream()), "av_relu_maxpool2_bwd")
    return dx


def _conv_wgrad(x: Tensor, dy: Tensor, N: int, H: int, W: int, C: int) -> Tensor:
    """dW [Cout, 9 C] (tap-major) = dY^T im2col(X), accumulated over frame chunks that keep the explicit im2col below a fixed size."""
    Cout = dy.shape[1]
    per = H * W * 9 * C * x.element_size()
    step = max(1, IM2COL_CHUNK_BYTES // per)
    dw = torch.zeros((Cout, 9 * C), dtype=torch.float32, device=x.device)
    x4 = x.view(N, H, W, C)
    for n0 in range(0, N, step):
        n1 = min(N, n0 + step)
        cols = torch.empty(((n1 - n0) * H * W, 9 * C), dtype=x.dtype, device=x.device)
        L.check(L.lib().av_im2col3(ops.ptr(x4[n0:n1]), ops.ptr(cols), ops.dt(x), n1 - n0, H, W, C, ops.stream()), "av_im2col3")
        ops.matmul_tn(dy[n0 * H * W:n1 * H * W], cols, out=dw, accumulate=True)
    return dw


class _LegacyFn(torch.autograd.Function):
    @staticmethod
    def forward(fctx, mod: "MultimodalCTCKoreanModel", save: bool, names, frames_A, frames_B, mel, *params):
        dtype = compute_dtype()
        dev = mel.device
        lip, aud = mod.lip_encoder, mod.audio_encoder
        B, T, C, Hh, Ww = frames_A.shape
        if frames_B.shape != frames_A.shape or mel.shape[0] != B:
            raise ValueError("MultimodalCTCKoreanModel: frames_A / frames_B / mel batch shapes differ")
        if mel.shape[1] != T:
            raise RuntimeError(f"MultimodalCTCKoreanModel: lip ({T}) and mel ({mel.shape[1]}) sequence lengths differ (torch.cat along the "
                               "feature axis fails in the reference too, :48-49)")
        if Hh % 4 or Ww % 4 or 64 * (Hh // 4) * (Ww // 4) != lip.rnn.input_size:
            raise RuntimeError(f"LipEncoder: {Hh}x{Ww} frames give {64 * (Hh // 4) * (Ww // 4)} features, the GRU expects {lip.rnn.input_size}")
        N = 2 * B * T                                                                       # both speakers in one pass
        fr = torch.cat([frames_A, frames_B], 0).contiguous().float()                        # [2B, T, C, H, W]
        x0 = torch.empty((N, Hh, Ww, CP), dtype=dtype, device=dev)
        L.check(L.lib().av_nchw_to_nhwc(ops.ptr(fr), ops.ptr(x0), ops.dt(x0), N, C, Hh, Ww, CP, ops.stream()), "av_nchw_to_nhwc")
        c1, c2 = lip.cnn[0], lip.cnn[3]
        w1 = mod._cache.get("w1", [c1.weight], dtype, lambda: ops.cast(F.pad(c1.weight.data.permute(0, 2, 3, 1), (0, CP - C)).reshape(32, 9 * CP).contiguous(), dtype))
        w2 = mod._cache.get("w2", [c2.weight], dtype, lambda: ops.cast(c2.weight.data.permute(0, 2, 3, 1).reshape(64, 9 * 32).contiguous(), dtype))
        y1 = _conv3(x0, N, Hh, Ww, CP, w1, c1.bias.data, dtype)                             # [N H W, 32]
        p1 = _relu_pool(y1, N, Hh, Ww, 32, False)                                           # [N, H/2, W/2, 32]
        H2, W2 = Hh // 2, Ww // 2
        y2 = _conv3(p1, N, H2, W2, 32, w2, c2.bias.data, dtype)                             # [N H2 W2, 64]
        p2 = _relu_pool(y2, N, H2, W2, 64, True)                                            # [N, 64 H4 W4] in (C, H', W') order
        Fdim = p2.shape[1]
        x_tm = torch.empty((T, 2 * B, Fdim), dtype=dtype, device=dev)
        L.check(L.lib().av_permute_bt(ops.ptr(p2), ops.dt(p2), ops.ptr(x_tm), ops.dt(x_tm), 2 * B, T, Fdim, ops.stream()), "av_permute_bt")
        lip_out, lctx = gru_forward(lip.rnn, mod._cache_lip, x_tm, save)                    # [2B, T, 2H]
        m_in = ops.cast(mel.contiguous().float(), dtype)                                    # [B, T, 80]
        m_tm = torch.empty((T, B, mel.shape[2]), dtype=dtype, device=dev)
        L.check(L.lib().av_permute_bt(ops.ptr(m_in), ops.dt(m_in), ops.ptr(m_tm), ops.dt(m_tm), B, T, mel.shape[2], ops.stream()), "av_permute_bt")
        aud_out, actx = gru_forward(aud.rnn, mod._cache_aud, m_tm, save)                    # [B, T, 2H]
        fusion = torch.cat([lip_out, torch.cat([aud_out, aud_out], 0)], dim=-1)             # [2B, T, 4H]  (:48-49; a copy, no arithmetic)
        fc = mod.fc
        wfc = mod._cache.get("fc", [fc.weight], dtype, lambda: ops.cast(fc.weight.data.contiguous(), dtype), flat=True)
        logits = ops.linear(fusion, wfc, fc.bias.data, out_dtype=torch.float32)             # [2B, T, V]
        fctx.saved = dict(x0=x0, y1=y1, p1=p1, y2=y2, x_tm=x_tm, lctx=lctx, actx=actx, fusion=fusion, wfc=wfc, w2=w2, dims=(B, T, C, Hh, Ww),
                          dtype=dtype) if save else None
        fctx.mod, fctx.names = mod, names
        return logits[:B], logits[B:]

    @staticmethod
    def backward(fctx, dA, dB):
        mod, s = fctx.mod, fctx.saved
        if s is None:
            return (None,) * (6 + len(fctx.names))
        B, T, C, Hh, Ww = s["dims"]
        dtype = s["dtype"]
        lip, aud = mod.lip_encoder, mod.audio_encoder
        H = lip.rnn.hidden_size
        g: Dict[str, Tensor] = {}
        N = 2 * B * T
        dlog = ops.cast(torch.cat([dA, dB], 0).contiguous().float(), dtype).view(2 * B * T, -1)   # [2B T, V]
        fus2 = s["fusion"].view(2 * B * T, 4 * H)
        g["fc.weight"] = ops.matmul_tn(dlog, fus2)
        g["fc.bias"] = ops.colsum(dlog)
        dfus = ops.matmul_nn(dlog, s["wfc"], out_dtype=torch.float32, b_is_weight=True).view(2 * B, T, 4 * H)
        d_lip = dfus[..., :2 * H].contiguous()                                              # [2B, T, 2H]
        d_aud = dfus[:B, :, 2 * H:].contiguous()
        ops.axpby(1.0, dfus[B:, :, 2 * H:].contiguous(), 1.0, d_aud)                        # the audio features feed both speakers (:48-49)
        gru_backward(aud.rnn, s["actx"], d_aud, g, "audio_encoder.rnn.", need_dx=False)
        dx_tm = gru_backward(lip.rnn, s["lctx"], d_lip, g, "lip_encoder.rnn.", need_dx=True)     # [T, 2B, F]
        Fdim = dx_tm.shape[-1]
        dp2 = torch.empty((2 * B, T, Fdim), dtype=dtype, device=dx_tm.device)
        L.check(L.lib().av_permute_bt(ops.ptr(dx_tm), ops.dt(dx_tm), ops.ptr(dp2), ops.dt(dp2), T, 2 * B, Fdim, ops.stream()), "av_permute_bt")
        H2, W2 = Hh // 2, Ww // 2
        dy2 = _relu_pool_bwd(s["y2"], dp2, N, H2, W2, 64, True)                             # [N H2 W2, 64]
        dw2 = _conv_wgrad(s["p1"].view(N * H2 * W2, 32), dy2, N, H2, W2, 32)                # [64, 9*32] tap-major
        g["lip_encoder.cnn.3.weight"] = dw2.view(64, 3, 3, 32).permute(0, 3, 1, 2).contiguous()
        g["lip_encoder.cnn.3.bias"] = ops.colsum(dy2)
        # dX of the second convolution = convolution of dY with the flipped filter, input / output channels swapped
        c2 = lip.cnn[3]
        w2f = mod._cache.get("w2f", [c2.weight], dtype,
                             lambda: ops.cast(c2.weight.data.flip(2, 3).permute(1, 2, 3, 0).reshape(32, 9 * 64).contiguous(), dtype))
        dp1 = _conv3(dy2, N, H2, W2, 64, w2f, None, dtype)                                  # [N H2 W2, 32]
        dy1 = _relu_pool_bwd(s["y1"], dp1, N, Hh, Ww, 32, False)                            # [N H W, 32]
        dw1 = _conv_wgrad(s["x0"].view(N * Hh * Ww, CP), dy1, N, Hh, Ww, CP)                # [32, 9*CP]
        g["lip_encoder.cnn.0.weight"] = dw1.view(32, 3, 3, CP)[..., :C].permute(0, 3, 1, 2).contiguous()
        g["lip_encoder.cnn.0.bias"] = ops.colsum(dy1)
        fctx.saved = None
        return (None, None, None, None, None, None) + tuple(g.get(n) for n in fctx.names)


class MultimodalCTCKoreanModel(nn.Module):
    def __init__(self, vocab_size=200, hidden_dim=256):
        super().__init__()
        if hidden_dim % 128:
            raise ValueError("hidden_dim must be a multiple of 128 for the HIP GRU kernels")
        self.lip_encoder = LipEncoder(hidden_dim=hidden_dim)
        self.audio_encoder = AudioEncoder(hidden_dim=hidden_dim)
        self.fc = nn.Linear(4 * hidden_dim, vocab_size)
        self._cache, self._cache_lip, self._cache_aud = ParamCache(), ParamCache(), ParamCache()

    def forward(self, frames_A, frames_B, mel):
        """frames_* [B, T, 3, 96, 96], mel [B, T, 80] -> (logits_A, logits_B) [B, T, vocab] fp32 (:44-54)."""
        if not mel.is_cuda:
            raise RuntimeError("MultimodalCTCKoreanModel (HIP): inputs must be on the GPU; there is no CPU fallback")
        named = list(self.named_parameters())
        names = [n for n, p in named if p.requires_grad]
        params = [p for n, p in named if p.requires_grad]
        save = torch.is_grad_enabled() and bool(names)
        return _LegacyFn.apply(self, save, names, frames_A, frames_B, mel, *params)


class _LogSoftmaxFn(torch.autograd.Function):
    @staticmethod
    def forward(fctx, x):
        y = ops.log_softmax_fwd(x.contiguous())
        fctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(fctx, dy):
        (y,) = fctx.saved_tensors
        return ops.log_softmax_bwd(y, dy.contiguous().float(), torch.float32)


def legacy_losses(logits_A: Tensor, logits_B: Tensor, mel_lengths, label_A, len_A, label_B, len_B) -> Tensor:
    """loss_A + loss_B of train_ctc_korea.py:93-98: log_softmax(2).transpose(0, 1) -> nn.CTCLoss(blank=0, zero_infinity=True) per
    speaker (nn.CTCLoss stays on PyTorch-ROCm as in the main path).  The label tensors are handed to the loss exactly as the reference
    hands them over: its collate pads them time-major (L, B) (:60,70-71) and nn.CTCLoss reads a 2-D target as (batch, length), so the
    reference only runs when L == B and then scores row n of that matrix against item n.  That is the reference's behaviour (the file
    is dead code upstream) and it is reproduced, not repaired; pass (B, L) targets to get the conventional pairing."""
    lpA = _LogSoftmaxFn.apply(logits_A).transpose(0, 1)
    lpB = _LogSoftmaxFn.apply(logits_B).transpose(0, 1)
    la = F.ctc_loss(lpA, label_A, mel_lengths, len_A, blank=0, zero_infinity=True)
    lb = F.ctc_loss(lpB, label_B, mel_lengths, len_B, blank=0, zero_infinity=True)
    return la + lb


def train_step(model: MultimodalCTCKoreanModel, optimizer, batch) -> Tensor:
    """One iteration of the reference's training loop (train_ctc_korea.py:89-104); ``batch`` = the tuple its collate_fn returns."""
    frames_A, frames_B, mel, mel_lengths, label_A, len_A, label_B, len_B = batch
    logits_A, logits_B = model(frames_A, frames_B, mel)
    loss = legacy_losses(logits_A, logits_B, mel_lengths, label_A, len_A, label_B, len_B)
    optimizer.zero_grad()
    loss.backward()
    optimizer.step()
    return loss.detach()
