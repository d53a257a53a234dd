"""VisualEncoder / AudioEncoder with the reference's constructor and forward signatures (model/encoder.py:57-100),
running on the HIP kernels of libavhip.so.  There is no PyTorch-op fallback: on a box without the library or
without a GPU the forward raises.
"""
from __future__ import annotations

import os
from typing import Optional, Union

import torch
import torch.nn as nn

from ..utils import init as _init
from .w2v2 import Wav2Vec2ModelHIP, load_local_config, w2v2_apply


class AudioEncoder(nn.Module):
    """model/encoder.py:80-100.  ``model_name`` is a LOCAL HF wav2vec2 directory (config.json [+ model.safetensors]);
    a dict is accepted as an in-memory config (random init).  Returns (last_hidden_state, mean(hidden_states[6:10]))."""

    def __init__(self, model_name: Union[str, dict] = "kresnik/wav2vec2-large-xlsr-korean", freeze: bool = True, seed: int = 2):
        super().__init__()
        weights = None
        if isinstance(model_name, dict):
            cfg = model_name
        elif os.path.isdir(model_name):
            cfg = load_local_config(model_name)
            st = os.path.join(model_name, "model.safetensors")
            if os.path.exists(st):
                from safetensors.torch import load_file
                weights = load_file(st)
        else:
            raise FileNotFoundError(
                f"AudioEncoder: {model_name!r} is not a local directory; pretrained checkpoints cannot be fetched "
                "(no network) — pass a local HF wav2vec2 directory or a config dict")
        self.model = Wav2Vec2ModelHIP(cfg)
        sd = _init.w2v2_state_dict(cfg, seed=seed, prefix="") if weights is None else weights
        missing = self.model.load_state_dict(sd, strict=False)
        if weights is not None and [k for k in missing.missing_keys if k != "masked_spec_embed"]:
            raise KeyError(f"AudioEncoder: checkpoint lacks {missing.missing_keys[:5]}")
        self.output_dim = self.model.config.hidden_size
        if freeze:
            for p in self.model.parameters():
                p.requires_grad = False

    def forward(self, x: torch.Tensor, attention_mask: Optional[torch.Tensor] = None):
        if not x.is_cuda:
            raise RuntimeError("AudioEncoder (HIP): input must be on the GPU; there is no CPU fallback")
        return w2v2_apply(self.model, x, attention_mask)
