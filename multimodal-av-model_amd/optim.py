"""Adam with a fused HIP step kernel (torch.optim.Adam defaults: betas (0.9, 0.999), eps 1e-8, no weight decay —
model/trainer.py:34-39).  State keys (step / exp_avg / exp_avg_sq) match torch.optim.Adam, so the reference's
checkpoint dict (main.py:47-55) stays loadable.  ``grad_scale`` folds the data-parallel 1/world_size (or a loss
un-scale) into the step so that no extra pass over the gradients is needed."""
from __future__ import annotations

import torch

from . import _lib as L
from . import ops


class AvAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        self.grad_scale = 1.0

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        st = ops.stream()
        fn = L.lib().av_adam_step
        for group in self.param_groups:
            b1, b2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                if not p.is_cuda or p.dtype != torch.float32:
                    raise RuntimeError("AvAdam: parameters must be float32 CUDA tensors (no CPU fallback)")
                state = self.state[p]
                if len(state) == 0:
                    state["step"] = 0
                    state["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    state["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                state["step"] = int(state["step"]) + 1
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                L.check(fn(ops.ptr(p), ops.ptr(g), ops.ptr(state["exp_avg"]), ops.ptr(state["exp_avg_sq"]), p.numel(), float(group["lr"]),
                           float(b1), float(b2), float(group["eps"]), state["step"], float(self.grad_scale), st), "av_adam_step")
                torch.autograd.graph.increment_version(p)   # updated in place by the kernel: refresh compute-dtype caches
        return loss
