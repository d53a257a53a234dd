"""Decoding helpers with the reference's names (beam_search.py:2-48).

``simple_beam_search`` there scores raw frame paths additively without prefix merging, so its best beam is
always the per-frame argmax path: it equals greedy CTC decoding (pinned in tests/golden).  This
version computes exactly that with one argmax over the whole [T,V] matrix and one host transfer instead of
T*beam_width ``.item()`` calls."""
from __future__ import annotations

import torch


def simple_beam_search(log_probs: torch.Tensor, beam_width=5, blank=0):
    ids = torch.argmax(log_probs, dim=-1).tolist()
    out, prev = [], None
    for i in ids:
        if i != prev and i != blank:
            out.append(i)
        prev = i
    return out


def greedy_batch(log_probs: torch.Tensor, blank: int, lengths: torch.Tensor = None):
    """[B,T,V] -> list of B id lists.  On the GPU: argmax + collapse in one HIP kernel (decode.hip), only the collapsed ids and
    their counts travel to the host."""
    if log_probs.is_cuda and log_probs.dim() == 3 and log_probs.shape[1] <= 4096:
        from . import _lib as L
        from . import ops
        lp = log_probs.detach().float().contiguous()
        B, T, V = lp.shape
        out = torch.empty((B, T), dtype=torch.int32, device=lp.device)
        cnt = torch.empty((B,), dtype=torch.int32, device=lp.device)
        ln = None if lengths is None else lengths.to(device=lp.device, dtype=torch.long).contiguous()
        L.check(L.lib().av_ctc_greedy(ops.ptr(lp), ops.ptr(ln), ops.ptr(out), ops.ptr(cnt), B, T, V, int(blank), ops.stream()), "av_ctc_greedy")
        out_h, cnt_h = out.cpu(), cnt.cpu().tolist()
        return [out_h[i, :cnt_h[i]].tolist() for i in range(B)]
    ids = torch.argmax(log_probs, dim=-1).cpu().tolist()
    res = []
    for row in ids:
        out, prev = [], None
        for i in row:
            if i != prev and i != blank:
                out.append(i)
            prev = i
        res.append(out)
    return res


def fast_decode(ids, tokenizer):
    return "".join(tokenizer.id_to_token[i] for i in ids if i != tokenizer.blank_id and 0 <= i < tokenizer.vocab_size
                   ).replace("▁", " ").strip()
