"""Compute-dtype switch of the HIP path.

``fp32``  parity mode: exact-fp32 MFMA (v_mfma_f32_16x16x4_f32); meets the 1e-3 gate against the CPU oracle.
``bf16``  perf mode (default): bf16 MFMA operands, fp32 accumulation, fp32 residual stream / master weights.
``fp16``  the reference's GPU arithmetic (torch.cuda.amp fp16 autocast + GradScaler, model/trainer.py:9,40,65; BASELINE configs[4]
          "fp16 + fp32 master"): IEEE-half MFMA operands (same matrix rate), fp32 accumulation / residual stream / master weights, served by
          libavhip_f16.so (the same kernels compiled with the half operand type).  Use it with ``loss_scaling=True``.
"""
from __future__ import annotations

import os

import torch

_MODES = {"fp32": torch.float32, "bf16": torch.bfloat16, "fp16": torch.float16}
_mode = os.environ.get("AVAMD_PRECISION", "bf16")
if _mode not in _MODES:
    raise ValueError(f"AVAMD_PRECISION must be one of {list(_MODES)}, got {_mode!r}")


def set_precision(mode: str) -> None:
    global _mode
    if mode not in _MODES:
        raise ValueError(f"precision must be one of {list(_MODES)}, got {mode!r}")
    _mode = mode


def get_precision() -> str:
    return _mode


def compute_dtype() -> torch.dtype:
    return _MODES[_mode]


def is_lp(dtype: torch.dtype) -> bool:
    """A 16-bit operand type of the fast kernels (bfloat16 in the default library, float16 in the fp16 one)."""
    return dtype in (torch.bfloat16, torch.float16)
