"""CTCDecoder on the HIP kernels (model/decoder.py:6-35): Linear(input_dim -> vocab) + log_softmax as one MFMA
GEMM + a wavefront-per-row log-softmax.  nn.CTCLoss itself stays on PyTorch-ROCm (BASELINE north_star).
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops
from ..precision import compute_dtype
from ..utils.shadow import ParamCache


class _HeadFn(torch.autograd.Function):
    @staticmethod
    def forward(fctx, x, w, b, wt, arena, wpad):
        dtype = compute_dtype()
        xt = ops.cast(x.contiguous().float(), dtype)
        logits = ops.linear(xt, wt, b.data, out_dtype=torch.float32)
        lp = ops.log_softmax_fwd(logits)
        fctx.save_for_backward(xt, wt, lp)
        fctx.dtype = dtype
        fctx.arena = arena
        fctx.wpad = wpad
        return lp

    @staticmethod
    def backward(fctx, dlp):
        xt, wt, lp = fctx.saved_tensors
        V, D = wt.shape
        wpad = fctx.wpad
        if wpad is not None:
            # bf16: the vocabulary (800) is not a multiple of the fast GEMM's 64-wide K step, which sent dX = dlogits W to the generic kernel
            # (349 us at 12800 x 1024 x 800).  dlogits is written with zero columns up to 832 and multiplies the zero-row-padded weight copy
            Vp = wpad.shape[0]
            dpad = ops.log_softmax_bwd(lp, dlp.contiguous().float(), fctx.dtype, pad_to=64).view(-1, Vp)
            d2 = dpad[:, :V]                                                     # row-strided view for the weight / bias gradients
            dx = ops.matmul_nn(dpad, wpad, out_dtype=torch.float32).view(xt.shape) if fctx.needs_input_grad[0] else None
        else:
            dlogits = ops.log_softmax_bwd(lp, dlp.contiguous().float(), fctx.dtype)
            d2 = dlogits.view(-1, V)
            dx = ops.matmul_nn(d2, wt, out_dtype=torch.float32).view(xt.shape) if fctx.needs_input_grad[0] else None
        ar, dev = fctx.arena, d2.device
        A = (lambda n, shp, vec=False: ar.out("decoder." + n, shp, dev, vec)) if ar is not None else (lambda n, shp, vec=False: None)
        dw = ops.matmul_tn(d2, xt.view(-1, D), out=A("weight", (V, D))) if fctx.needs_input_grad[1] else None
        db = ops.colsum_into(d2, A("bias", (V,), True)) if fctx.needs_input_grad[2] else None
        return dx, dw, db, None, None, None


class CTCDecoder(nn.Module):
    def __init__(self, input_dim, vocab_size, blank_id=0):
        super().__init__()
        self.net = nn.Sequential(nn.Linear(input_dim, vocab_size))     # parameter container: keys net.0.{weight,bias}
        self.ctc_loss = nn.CTCLoss(blank=blank_id, zero_infinity=True)
        self._cache = ParamCache()
        self.grad_arena = None          # parallel.dp.GradArena shared with the fusion module (the trainer's head gradient bucket)
        self._wpad = None               # bf16 weight with zero rows up to a multiple of 64 (see forward)

    def forward(self, x, target=None, input_lengths=None, target_lengths=None):
        """x [B,T,D] -> log-probs [B,T,V], or the CTC loss when ``target`` is given (model/decoder.py:14-35)."""
        if not x.is_cuda:
            raise RuntimeError("CTCDecoder (HIP): input must be on the GPU; there is no CPU fallback")
        lin = self.net[0]
        dtype = compute_dtype()
        wpad = None
        if dtype == torch.float32:
            wt = lin.weight.data
        else:
            V, D = lin.weight.shape
            Vp = (V + 63) // 64 * 64
            if Vp != V and D % 8 == 0:
                # the compute-dtype weight lives at the head of a [Vp, D] buffer whose extra rows are zero: the [V, D] view is the forward's
                # operand and the optimizer's bf16 shadow, the whole buffer the K-padded operand of the backward's dX product
                def build():
                    self._wpad = torch.zeros((Vp, D), dtype=dtype, device=lin.weight.device)
                    self._wpad[:V].copy_(ops.cast(lin.weight.data.contiguous(), dtype))
                    return self._wpad[:V]
                wt = self._cache.get("w", [lin.weight], dtype, build, flat=True)
                wpad = self._wpad
            else:
                wt = self._cache.get("w", [lin.weight], dtype, lambda: ops.cast(lin.weight.data.contiguous(), dtype), flat=True)
        log_probs = _HeadFn.apply(x, lin.weight, lin.bias, wt, self.grad_arena, wpad)
        if target is not None:
            return self.ctc_loss(log_probs.transpose(0, 1), target, input_lengths, target_lengths)
        return log_probs
