"""Checkpoint helpers with the reference's dict layout (main.py:47-64): ``epoch`` + four module state_dicts (the reference's key
sets: visual 129, audio 422, fusion 30, decoder 2) + the optimizer state (torch-Adam keys), so a file written by either side
loads on the other.  Loading never unpickles code: ``torch.load(..., weights_only=True)``."""
from __future__ import annotations

import torch

KEYS = ("epoch", "visual_encoder", "audio_encoder", "fusion", "decoder1", "optimizer")


def checkpoint_dict(epoch: int, trainer) -> dict:
    sc = getattr(trainer, "scaler", None)
    if sc is not None and sc.enabled:
        trainer.optimizer.sync_steps(sc)             # steps actually taken (overflowing steps are skipped on the device)
    extra = {"scaler": sc.state_dict()} if sc is not None and sc.enabled else {}
    return {**extra,
        "epoch": int(epoch),
        "visual_encoder": trainer.visual_encoder.state_dict(),
        "audio_encoder": trainer.audio_encoder.state_dict(),
        "fusion": trainer.fusion_module.state_dict(),
        "decoder1": trainer.decoder1.state_dict(),
        "optimizer": trainer.optimizer.state_dict(),
    }


def save_checkpoint(epoch: int, trainer, path: str) -> None:
    """main.py:47-55."""
    torch.save(checkpoint_dict(epoch, trainer), path)


def load_checkpoint(trainer, path: str, *, audio_encoder: bool = False, optimizer: bool = False) -> int:
    """main.py:57-64: restores visual encoder, fusion and decoder; the reference keeps the audio encoder and the optimizer
    lines commented out, so they are opt-in here.  Returns the epoch to resume at."""
    ck = torch.load(path, map_location=getattr(trainer, "device", "cpu"), weights_only=True)
    missing = [k for k in KEYS if k not in ck]
    if missing:
        raise KeyError(f"load_checkpoint: {path!r} lacks {missing}")
    trainer.visual_encoder.load_state_dict(ck["visual_encoder"])
    if audio_encoder:
        trainer.audio_encoder.load_state_dict(ck["audio_encoder"])
    trainer.fusion_module.load_state_dict(ck["fusion"])
    trainer.decoder1.load_state_dict(ck["decoder1"])
    if optimizer:
        trainer.optimizer.load_state_dict(ck["optimizer"])
        sc = getattr(trainer, "scaler", None)
        if sc is not None and sc.enabled and "scaler" in ck:
            sc.load_state_dict(ck["scaler"])                     # (AvAdam.load_state_dict re-seeds the device-side per-parameter step table)
    return int(ck["epoch"]) + 1
