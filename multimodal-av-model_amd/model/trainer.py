"""MultimodalTrainer with the reference's surface (model/trainer.py:12-252) on the HIP path.

Same constructor arguments and attributes (.visual_encoder .audio_encoder .fusion_module .decoder1 .optimizer
.device .tokenizer), ``train_epoch(loader) -> float``, ``evaluate(loader) -> (loss, wer)``, ``ctc_decode(ids)``.
Differences that do not change results:
  * compute runs in the package precision mode (bf16 perf / fp32 parity) instead of fp16 autocast + GradScaler;
  * attn_mask1 == attn_mask2 always, so with every stochastic regulariser of wav2vec2 at 0 the two audio
    passes of the reference are bit-identical and the encoder runs ONCE; as soon as dropout / LayerDrop / SpecAugment
    are active in train mode it runs twice with independent masks, as the reference does (``audio_passes`` forces it);
  * the class counts of the contrastive loss are taken from the CPU masks (no host sync);
  * optional data parallelism: bucketed gradient all-reduce over RCCL overlapped with the audio backward.
"""
from __future__ import annotations

import os
from typing import Dict, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops
from .. import _lib as L
from ..beam_search import fast_decode, greedy_batch
from ..contrastive import contrastive_loss_with_mask
from ..optim import AvAdam, AvGradScaler
from ..parallel.dp import GradArena, GradBucketReducer


def word_error_rate(refs, hyps) -> float:
    """jiwer.wer stand-in (jiwer is not installed): total word-level edit distance / total reference words."""
    errs = words = 0
    for r, h in zip(refs, hyps):
        r, h = r.split(), h.split()
        prev = list(range(len(h) + 1))
        for i in range(1, len(r) + 1):
            cur = [i] + [0] * len(h)
            for j in range(1, len(h) + 1):
                cur[j] = min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (r[i - 1] != h[j - 1]))
            prev = cur
        errs += prev[-1]
        words += len(r)
    return errs / max(1, words)


_FUSED_LOSS = os.environ.get("AVAMD_FUSED_LOSS", "1") != "0"
_VIS_STREAMS = int(os.environ.get("AVAMD_VISUAL_STREAMS", "2"))          # 1: both speakers' lip streams on one side stream


class _CombineFn(torch.autograd.Function):
    """total = mean-reduced CTC of both speakers / 2 + lambda (c1 + c2) / 2 (model/trainer.py:111-119) in one kernel."""

    @staticmethod
    def forward(fctx, nll, w, c1, c2, lam):
        out = torch.empty(3, dtype=torch.float32, device=nll.device)
        c1f = c1.detach().reshape(1).float() if c1.is_cuda else None
        c2f = c2.detach().reshape(1).float() if c2.is_cuda else None
        L.check(L.lib().av_loss_combine(ops.ptr(nll.detach().contiguous()), ops.ptr(w.contiguous()), ops.ptr(c1f), ops.ptr(c2f), 0.5 * lam,
                                        nll.numel(), ops.ptr(out), ops.stream()), "av_loss_combine")
        fctx.save_for_backward(w)
        fctx.lam = lam
        total, l1, l2 = out[0].clone(), out[1].clone(), out[2].clone()
        fctx.mark_non_differentiable(l1, l2)
        return total, l1, l2

    @staticmethod
    def backward(fctx, g, _g1, _g2):
        (w,) = fctx.saved_tensors
        gc = g * (0.5 * fctx.lam)
        return g * w, None, gc, gc, None


class MultimodalTrainer:
    def __init__(self, visual_encoder, audio_encoder, fusion_module, decoder1, tokenizer, learning_rate=1e-4, device="cuda",
                 lambda_=0.1, audio_passes: Optional[int] = None, reducer: Optional[GradBucketReducer] = None, pair_batched: bool = True,
                 visual_side_stream: bool = True, loss_scaling: bool = False):
        self.visual_encoder = visual_encoder.to(device)
        self.audio_encoder = audio_encoder.to(device)
        self.fusion_module = fusion_module.to(device)
        self.decoder1 = decoder1.to(device)
        self.tokenizer = tokenizer
        self.device = device
        self.lambda_ = lambda_
        self.audio_passes = audio_passes
        # both speakers go through fusion / BiLSTM / CTC head as ONE 2B batch (items are independent there, so results are
        # identical; it halves the number of latency-bound LSTM step launches), and the frozen visual encoder runs on a
        # side stream concurrently with the wav2vec2 forward
        self.pair_batched = pair_batched
        self.visual_side_stream = visual_side_stream
        self._vstream = None
        self.ctc_loss = nn.CTCLoss(blank=tokenizer.blank_id, zero_infinity=True)       # stays on PyTorch-ROCm
        self.parameters = (list(self.visual_encoder.parameters()) + list(self.audio_encoder.parameters())
                           + list(self.fusion_module.parameters()) + list(self.decoder1.parameters()))
        self.optimizer = AvAdam([
            {"params": list(self.visual_encoder.parameters()), "lr": learning_rate},
            {"params": list(self.audio_encoder.parameters()), "lr": 2e-5},
            {"params": list(self.fusion_module.parameters()), "lr": learning_rate},
            {"params": list(self.decoder1.parameters()), "lr": learning_rate},
        ])
        # model/trainer.py:40 (GradScaler); off by default: bf16 operands need no loss scaling.  On = the reference's overflow-skip /
        # growth / backoff law, evaluated on the device
        self.scaler = AvGradScaler(device=device, enabled=loss_scaling)
        self.projection_layer = None
        self.fixed_projection = None        # (weight, bias) to inject instead of a fresh random layer (parity tests)
        # decoder + fusion gradients are produced straight into ONE flat buffer (no packing copy before their all-reduce, no per-step
        # allocations, a pointer table of the fused Adam that never changes); the wav2vec2 layers keep one such buffer each
        self._head_arena = GradArena()
        self.fusion_module.grad_arena = self.decoder1.grad_arena = self._head_arena
        self.reducer = reducer
        if reducer is not None:
            self.optimizer.grad_scale = 1.0 / reducer.world
            if reducer.world > 1:
                # the collective schedule must not depend on the rank: LayerDrop decisions come from a generator seeded identically
                # everywhere (every rank draws once per layer and pass), dropout masks from one seeded per rank
                import torch.distributed as dist
                m = self.audio_encoder.model
                m.layerdrop_generator = torch.Generator().manual_seed(0x5EED1A7E)
                m.dropout_generator = torch.Generator().manual_seed(0xD120 + 7919 * dist.get_rank(reducer.group))
            self.audio_encoder.model.grad_ready = reducer.reduce_async
            self.audio_encoder.model.grad_flat_ready = reducer.reduce_flat
            self.audio_encoder.model.grad_wait = reducer.wait
            self._head_params = [p for m in (self.decoder1, self.fusion_module) for n, p in m.named_parameters()
                                 if not n.startswith("cross_attn_visual.")]
            # decoder + fusion gradients are complete when the wav2vec2 backward starts (their AccumulateGrad nodes run with top
            # priority right after the fusion backward): start their all-reduce there, under the whole audio backward
            self._head_done = False
            if pair_batched:                                   # one fusion / decoder call per step: no later accumulation into these grads
                self.audio_encoder.model.grad_pre = self._reduce_head_early

    # ------------------------------------------------------------------------------------------------------------
    def _to_dev(self, batch):
        dev = self.device
        nb = lambda t: t.to(dev, non_blocking=True)
        d = {k: nb(batch[k]) for k in ("audio", "mask1", "mask2", "text1", "text2", "text1_lengths", "text2_lengths")}
        # [B,T,1,H,W] -> [B,1,T,H,W]: with one channel the two layouts are byte-identical => a view
        d["lip1"] = nb(batch["lip1"]).permute(0, 2, 1, 3, 4)
        d["lip2"] = nb(batch["lip2"]).permute(0, 2, 1, 3, 4)
        return d

    def _mask_ds(self, mask, T_enc):
        B, Tin = mask.shape
        out = torch.empty((B, T_enc), dtype=torch.long, device=mask.device)
        L.check(L.lib().av_mask_downsample(ops.ptr(mask.contiguous()), ops.ptr(out), B, Tin, T_enc, ops.stream()), "av_mask_downsample")
        return out

    @staticmethod
    def _mask_ds_host(mask_cpu: torch.Tensor, T_enc: int) -> torch.Tensor:
        """Host restatement of av_mask_downsample (nearest index floor(i * Tin / T_enc), fp32 as on the device)."""
        mask_cpu = mask_cpu.cpu()
        Tin = mask_cpu.shape[1]
        scale = torch.tensor(Tin / T_enc, dtype=torch.float32)
        idx = torch.floor(torch.arange(T_enc, dtype=torch.float32) * scale).long().clamp_(max=Tin - 1)
        return mask_cpu[:, idx]

    @staticmethod
    def _class_counts(mask_cpu: torch.Tensor, T_enc: int):
        """(#1, #2, #0) of the down-sampled mask, computed on the host copy of the batch (no device sync when the
        batch arrives from a DataLoader; a device-resident batch should carry precomputed ``_counts1/_counts2``)."""
        c = torch.bincount(MultimodalTrainer._mask_ds_host(mask_cpu, T_enc).reshape(-1).clamp(0, 3), minlength=4).tolist()
        return (c[1], c[2], c[0])

    @staticmethod
    def _fusion_lengths_host(mask_cpu: torch.Tensor, T_enc: int, Tv: int) -> torch.Tensor:
        """Host restatement of the ``input_lengths`` CrossAttentionFusion returns for ONE call (model/fusion_module.py:41-59 of the
        reference; av_fusion_gather_lerp_fwd here): n_b speech frames (mask 1/2) are kept per item and padded to the call's
        maximum Tm, the mask is resampled to Tv frames with nearest index floor(i * Tm / Tv), and the length is the number
        of resampled positions that fall on a kept frame."""
        import numpy as np
        m = MultimodalTrainer._mask_ds_host(mask_cpu, T_enc)
        n = ((m == 1) | (m == 2)).sum(1).numpy().astype(np.int64)
        Tm = int(n.max()) if n.size else 0
        i = np.arange(Tv, dtype=np.float32)
        if Tm == Tv:
            j = np.arange(Tv, dtype=np.int64)
        else:
            j = np.floor(i * (np.float32(Tm) / np.float32(Tv))).astype(np.int64)
            j = np.minimum(j, max(Tm - 1, 0))
        return torch.from_numpy((j[None, :] < n[:, None]).sum(1).astype(np.int64))

    def host_metadata(self, cpu_batch: Dict[str, torch.Tensor], T_enc: int) -> Dict[str, object]:
        """Everything the step would otherwise read back from the device, computed from the HOST copy of a collated batch:
        contrastive class counts and the CTC length vectors (nn.functional.ctc_loss wants its lengths on the host; handing
        it device tensors costs a full device synchronisation per call).  Keys start with ``_``; merge into the batch."""
        Tv = cpu_batch["lip1"].shape[1]
        il = torch.cat([self._fusion_lengths_host(cpu_batch["mask1"], T_enc, Tv), self._fusion_lengths_host(cpu_batch["mask2"], T_enc, Tv)])
        tl = torch.cat([cpu_batch["text1_lengths"].cpu().long(), cpu_batch["text2_lengths"].cpu().long()])
        return {"_counts1": self._class_counts(cpu_batch["mask1"], T_enc), "_counts2": self._class_counts(cpu_batch["mask2"], T_enc),
                "_ctc_input_lengths": il, "_ctc_target_lengths": tl,
                "_same_padding": bool(torch.equal(cpu_batch["mask1"] != 3, cpu_batch["mask2"] != 3)),
                "_audio_valid1": (cpu_batch["mask1"] != 3).sum(1).tolist(), "_audio_valid2": (cpu_batch["mask2"] != 3).sum(1).tolist()}

    def forward_losses(self, batch: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        """model/trainer.py:66-119 for one batch; everything stays on the device."""
        d = self._to_dev(batch)
        use_side = self.visual_side_stream and d["audio"].is_cuda
        if use_side:
            # the frozen visual encoder runs beside the wav2vec2 forward, one stream per speaker: its HBM-bound BatchNorm / pooling passes overlap
            # the other streams' MFMA-bound kernels
            dev_ = d["audio"].device
            if self._vstream is None:
                self._vstream = torch.cuda.Stream(device=dev_)
                self._vstream2 = torch.cuda.Stream(device=dev_) if _VIS_STREAMS >= 2 else self._vstream
            main = torch.cuda.current_stream(dev_)
            two = self._vstream2 is not self._vstream
            evs = []
            from ..precision import compute_dtype
            self.visual_encoder.warm_caches(compute_dtype())       # weight re-layouts are built on main, before the lip streams fork
            self._vstream.wait_stream(main)
            with torch.cuda.stream(self._vstream):
                if two:
                    self.visual_encoder._bn_sync, self.visual_encoder._bn_idx = ("lead", evs), 0
                vf1 = self.visual_encoder(d["lip1"])
            if two:
                self._vstream2.wait_stream(main)
                self.visual_encoder._bn_sync, self.visual_encoder._bn_idx = ("follow", evs), 0
            try:
                with torch.cuda.stream(self._vstream2):
                    vf2 = self.visual_encoder(d["lip2"])
            finally:
                self.visual_encoder._bn_sync = None
        else:
            vf1 = self.visual_encoder(d["lip1"])
            vf2 = self.visual_encoder(d["lip2"])
        attn1 = d["mask1"] != 3
        passes = self.audio_passes
        if passes is None:      # auto: the duplicate pass only differs when a stochastic regulariser is active in train mode
            cfg = getattr(self.audio_encoder.model, "cfg", {})
            knobs = ("hidden_dropout", "attention_dropout", "activation_dropout", "feat_proj_dropout", "layerdrop", "mask_time_prob")
            passes = 2 if (self.audio_encoder.training and any(cfg.get(k, 0) > 0 for k in knobs)) else 1
        if passes == 1:
            # one pass stands for both only if the two padding patterns are the same (they are for the reference's dataset, SURVEY
            # §0.3; a batch from elsewhere falls back to two passes).  Host copy when available, otherwise one device comparison
            same = batch.get("_same_padding")
            if same is None:
                same = bool(torch.equal(batch["mask1"] != 3, batch["mask2"] != 3))
            if not same:
                passes = 2
        if passes == 2:
            self.audio_encoder.model._feat_cache = {}              # both passes read the same waveform: one conv feature-extractor run
        try:
            v1, v2 = batch.get("_audio_valid1"), batch.get("_audio_valid2")      # host copies of the valid lengths (host_metadata)
            if v1 is None and not batch["mask1"].is_cuda:                        # batch straight from a DataLoader: the masks ARE on the host
                v1, v2 = (batch["mask1"] != 3).sum(1).tolist(), (batch["mask2"] != 3).sum(1).tolist()
            if passes == 2:
                a1, mid1, a2, mid2 = self.audio_encoder.forward_pair(d["audio"], attn1, d["mask2"] != 3, v1, v2)
            else:
                a1, mid1 = self.audio_encoder(d["audio"], attention_mask=attn1, valid_lengths=v1)
                a2, mid2 = a1, mid1
        finally:
            self.audio_encoder.model._feat_cache = None
        T_enc, D = a1.shape[1], a1.shape[2]
        m1 = self._mask_ds(d["mask1"], T_enc)
        m2 = self._mask_ds(d["mask2"], T_enc)
        out = {}
        if self.lambda_ != 0:
            if self.projection_layer is None:
                self.projection_layer = nn.Linear(D, 128).to(self.device)          # model/trainer.py:105-106
                if self.fixed_projection is not None:
                    with torch.no_grad():
                        self.projection_layer.weight.copy_(self.fixed_projection[0]); self.projection_layer.bias.copy_(self.fixed_projection[1])
            k1 = batch.get("_counts1") or self._class_counts(batch["mask1"], T_enc)
            k2 = batch.get("_counts2") or self._class_counts(batch["mask2"], T_enc)
            c1 = contrastive_loss_with_mask(mid1, m1.reshape(-1), self.projection_layer, counts=k1)
            c2 = contrastive_loss_with_mask(mid2, m2.reshape(-1), self.projection_layer, counts=k2)
        else:
            c1 = c2 = torch.zeros((), device=a1.device)
        if use_side:
            main.wait_stream(self._vstream)
            if self._vstream2 is not self._vstream:
                main.wait_stream(self._vstream2)
            vf1.record_stream(main); vf2.record_stream(main)
        B = a1.shape[0]
        # the two speakers go through fusion / decoder / CTC as one stacked call only if their lip clips were padded to the same length;
        # the reference's collate pads lip1 and lip2 separately (dataset/collate_fn.py:8-13,27-31), so a real batch can differ - then one
        # call per speaker, as the reference does (found by tests/test_dataset_gpu.py::test_dataset_feeds_the_training_step)
        pair = self.pair_batched and vf1.shape[1] == vf2.shape[1]
        self._pair_step = pair
        if pair:
            f12, il12 = self.fusion_module(torch.cat([vf1, vf2], 0), torch.cat([a1, a2], 0), mask=torch.cat([m1, m2], 0), groups=2)
            lp12 = self.decoder1(f12)
            f1, f2, il1, il2, lp1, lp2 = f12[:B], f12[B:], il12[:B], il12[B:], lp12[:B], lp12[B:]
        else:
            f1, il1 = self.fusion_module(vf1, a1, mask=m1)
            f2, il2 = self.fusion_module(vf2, a2, mask=m2)
            lp1 = self.decoder1(f1)
            lp2 = self.decoder1(f2)
        if pair:
            # one nn.functional.ctc_loss call (PyTorch-ROCm, as north_star prescribes) over the 2B items; 'mean' reduction
            # of nn.CTCLoss = mean_i(nll_i / clamp(target_len_i, 1)) is re-applied per speaker half
            t1, t2 = d["text1"], d["text2"]
            Lm = max(t1.shape[1], t2.shape[1])
            tg = torch.cat([F.pad(t1, (0, Lm - t1.shape[1])), F.pad(t2, (0, Lm - t2.shape[1]))], 0)
            tl = torch.cat([d["text1_lengths"], d["text2_lengths"]], 0)
            il_h, tl_h = batch.get("_ctc_input_lengths"), batch.get("_ctc_target_lengths")     # host copies: no device sync in ctc_loss
            w_ctc = (0.5 / B) / tl.clamp_min(1).to(torch.float32)        # weights of the per-speaker means (before the CTC call: off the sync)
            self.fusion_module.stage_flag_check()                  # BiLSTM timeout words -> pinned memory, visible after ctc_loss's own sync
            nll = F.ctc_loss(lp12.transpose(0, 1), tg, il12 if il_h is None else il_h, tl if tl_h is None else tl_h,
                             blank=self.tokenizer.blank_id, reduction="none", zero_infinity=True)
            self.fusion_module.finish_flag_check()                 # raises if a persistent BiLSTM launch timed out (event query, no sync)
            if nll.is_cuda and nll.dtype == torch.float32 and _FUSED_LOSS:
                total, l1, l2 = _CombineFn.apply(nll, w_ctc, c1, c2, float(self.lambda_))    # one kernel forward, one backward
            else:
                per = nll / tl.clamp_min(1).to(nll.dtype)
                l1, l2 = per[:B].mean(), per[B:].mean()
                total = None
        else:
            total = None
            l1 = self.ctc_loss(lp1.transpose(0, 1), d["text1"], il1, d["text1_lengths"])
            l2 = self.ctc_loss(lp2.transpose(0, 1), d["text2"], il2, d["text2_lengths"])
        if total is None:
            total = (l1 + l2) / 2 + self.lambda_ * (c1 + c2) / 2
        out.update(visual_feat1=vf1, visual_feat2=vf2, audio_last=a1, audio_mid=mid1, fused1=f1, fused2=f2, input_lengths1=il1,
                   input_lengths2=il2, log_probs1=lp1, log_probs2=lp2, loss1=l1, loss2=l2, contrast1=c1, contrast2=c2, total=total)
        return out

    def train_step(self, batch) -> Dict[str, torch.Tensor]:
        """zero_grad -> forward -> backward (-> bucketed all-reduce) -> Adam; no host sync."""
        self.optimizer.zero_grad(set_to_none=True)
        self._head_arena.begin_step()
        begin = getattr(self.audio_encoder.model, "begin_grad_step", None)
        if begin is not None:
            begin()
        if self.reducer is not None:
            self._head_done = False
        out = self.forward_losses(batch)
        self.scaler.scale(out["total"]).backward()                 # model/trainer.py:121
        self._head_arena.finalize()
        if self.reducer is not None:
            # wav2vec2 layer buckets were reduced inside its backward (overlapped), the decoder + fusion bucket at its start; if that
            # backward did not run (nothing trainable below the fusion) the bucket goes now
            if not self._head_done:
                self._reduce_head()
            self.reducer.wait()
            self._head_done = False
        self.scaler.step(self.optimizer)                           # model/trainer.py:122-123 (update() is part of the device-side step)
        self.scaler.update()
        return out

    def _reduce_head_early(self):
        # hook at the start of the wav2vec2 backward: valid only when this step made ONE fusion / decoder call (a batch whose two lip
        # clips were padded differently falls back to one call per speaker: its head bucket goes after the backward, in train_step)
        if getattr(self, "_pair_step", True):
            self._reduce_head()

    def _reduce_head(self):
        # once per step: with two audio passes the wav2vec2 backward (and this hook) runs twice, and p.grad are by then views of the
        # already reduced bucket (a second all-reduce would sum them world_size times)
        if self._head_done:
            return
        hp = [p for p in self._head_params if p.grad is not None]
        ar = self._head_arena
        if hp and ar.flat is not None and all(ar.owns(p.grad) for p in hp):
            self.reducer.reduce_flat(ar)                           # the gradients ARE the bucket: all-reduced in place, nothing is packed
        else:                                                      # first step (layout still being discovered) / two fusion calls per step
            for p, v in zip(hp, self.reducer.reduce_async([p.grad for p in hp])):
                p.grad = v
        self._head_done = True

    def train_epoch(self, dataloader):
        self.visual_encoder.train(); self.audio_encoder.train(); self.fusion_module.train(); self.decoder1.train()
        self.projection_layer = None
        total_loss = 0.0
        for batch_idx, batch in enumerate(dataloader):
            try:
                out = self.train_step(batch)
                total_loss += out["total"].item()
                if batch_idx % 100 == 0:
                    print(f"[Batch {batch_idx}] CTC1: {out['loss1'].item():.4f}, CTC2: {out['loss2'].item():.4f}, "
                          f"Contrast1: {float(out['contrast1']):.4f}, Contrast2: {float(out['contrast2']):.4f}, "
                          f"Total: {out['total'].item():.4f}", flush=True)
            except NotImplementedError:  # a configuration this build does not cover: every later batch would fail the same way
                raise
            except Exception as e:       # model/trainer.py:162-164: skip the batch, keep going
                print(f"Error at batch {batch_idx}: {e}", flush=True)
                if self.reducer is not None and self.reducer.world > 1:
                    # data parallel: a rank that skips a batch no longer issues the collectives its peers are waiting in
                    raise
                continue
        return total_loss / max(1, len(dataloader))

    def ctc_decode(self, pred_ids):
        """model/trainer.py:168-177 (prev is NOT reset on blank — kept as in the reference's debug helper)."""
        result, prev = [], None
        for idx in pred_ids:
            if idx == self.tokenizer.blank_id:
                continue
            if idx != prev:
                result.append(idx)
            prev = idx
        return result

    @torch.no_grad()
    def evaluate(self, dataloader):
        self.visual_encoder.eval(); self.audio_encoder.eval(); self.fusion_module.eval(); self.decoder1.eval()
        refs1, hyps1, refs2, hyps2 = [], [], [], []
        total_loss = 0.0
        lam, self.lambda_ = self.lambda_, 0.0
        try:
            for batch in dataloader:
                out = self.forward_losses(batch)
                total_loss += (out["loss1"].item() + out["loss2"].item()) / 2
                for spk, lp, refs, hyps in (("1", out["log_probs1"], refs1, hyps1), ("2", out["log_probs2"], refs2, hyps2)):
                    ids = greedy_batch(lp, self.tokenizer.blank_id)       # == simple_beam_search best beam
                    txt, tl = batch["text" + spk], batch["text" + spk + "_lengths"]
                    for i, seq in enumerate(ids):
                        hyps.append(fast_decode(seq, self.tokenizer))
                        refs.append(self.tokenizer.decode(txt[i][: int(tl[i])].tolist()))
        finally:
            self.lambda_ = lam
        wer1, wer2 = word_error_rate(refs1, hyps1), word_error_rate(refs2, hyps2)
        avg_wer = (wer1 + wer2) / 2
        avg_loss = total_loss / max(1, len(dataloader))
        print(f"[Eval] WER1: {wer1:.3f}, WER2: {wer2:.3f}, Avg: {avg_wer:.3f}, Loss: {avg_loss:.4f}")
        self.last_decoded = (hyps1, hyps2)
        return avg_loss, avg_wer
