"""Data-parallel gradient exchange: one process per GPU, RCCL (torch.distributed backend "nccl") over xGMI.

The reference has no distributed code; the build adds exactly one collective: all-reduce of the
63.8 M trainable-and-used gradients.  Buckets follow backward order — decoder + fusion first (ready before the
audio backward starts), then one bucket per wav2vec2 layer 9 -> 6 — each packed into one flat buffer and
all-reduced on a SIDE stream while the next layer's backward runs on the main stream; only the last bucket is
exposed.  The 1/world_size is folded into the Adam kernel (``AvAdam.grad_scale``).  Works on CPU tensors with the
gloo backend (tests) as well.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch
import torch.distributed as dist


class GradBucketReducer:
    def __init__(self, group=None, side_stream: bool = True, always_collective: bool = False):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        # issue the collectives even in a group of one rank (rehearsal of the RCCL path on a one-GPU box: same calls, same streams)
        self.always = always_collective and dist.is_initialized()
        self.pending = []            # (work, flat, event)
        self.stream: Optional[torch.cuda.Stream] = None
        self._use_side = side_stream

    def _side(self, dev):
        if self.stream is None:
            self.stream = torch.cuda.Stream(device=dev)
        return self.stream

    def reduce_async(self, tensors: List[torch.Tensor]) -> List[torch.Tensor]:
        """Pack ``tensors`` into one flat bucket, start its all-reduce (SUM), return views of the bucket that alias
        the reduced values once ``wait()`` has been called."""
        tensors = [t for t in tensors if t is not None]
        if not tensors:
            return []
        flat = torch.cat([t.reshape(-1) for t in tensors])
        views, off = [], 0
        for t in tensors:
            views.append(flat[off:off + t.numel()].view(t.shape))
            off += t.numel()
        if self.world > 1 or self.always:
            if flat.is_cuda and self._use_side:
                s = self._side(flat.device)
                s.wait_stream(torch.cuda.current_stream(flat.device))
                with torch.cuda.stream(s):
                    work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
                flat.record_stream(s)
                self.pending.append((work, flat, s))
            else:
                work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
                self.pending.append((work, flat, None))
        return views

    def wait(self) -> None:
        import os, time
        timing = os.environ.get("AVAMD_DP_TIMING") == "1"
        for work, flat, s in self.pending:
            t0 = time.perf_counter()
            work.wait()
            if timing:
                import sys
                print(f"[dp] bucket {flat.numel() * flat.element_size() / 1e6:8.1f} MB waited {1e3 * (time.perf_counter() - t0):8.1f} ms", file=sys.stderr, flush=True)
            if s is not None:
                torch.cuda.current_stream(flat.device).wait_stream(s)
        self.pending.clear()


def shard_batch(batch: Dict[str, torch.Tensor], rank: int, world: int) -> Dict[str, torch.Tensor]:
    """Contiguous item shard of a collated batch (independent batch items)."""
    B = next(iter(batch.values())).shape[0]
    if B % world:
        raise ValueError(f"global batch {B} is not divisible by world size {world} (CTC mean needs equal shards)")
    per = B // world
    return {k: v[rank * per:(rank + 1) * per] for k, v in batch.items()}
