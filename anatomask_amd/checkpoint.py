"""Checkpoint writer / reader for the pretraining run and the hand-off to STUNet finetuning.

The reference saves, once per epoch (P/pretrain_AntoMask.py:472-479):
    {'network_weights': model.state_dict()   # LocalDDP/DDP-wrapped -> keys prefixed 'module.'
     'optimizer_state', 'grad_scaler_state': None, 'train_loss': [...], 'current_epoch': i}
and the finetuning loader (nnunetv2/run/load_pretrained_weights.py:66-106, load_stunet_ssl_weights) keeps the entries whose key
contains 'encoder', strips everything up to and including 'sp_cnn.', and load_state_dict(strict=False)s them into the full
STUNet.  We write exactly that, plus what the reference forgot and a resume needs (teacher/EMA weights, AdamW moments of the
fused optimizer, step counter, RNG state) under extra keys that the reference loader ignores.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch


def reference_state(trainer, train_losses: List[float], epoch: int) -> Dict:
    m = trainer.model
    sd = {"module." + k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    n = m._live_end
    names = [k for k in m._pnames if k not in m._dead]
    opt = {"state": {}, "param_names": names, "step": trainer.step_count, "lr": trainer.lr, "betas": trainer.betas, "eps": trainer.eps,
           "weight_decay": trainer.wd}
    for k in names:
        o, num = m._offs[k], m._W[k].numel()
        opt["state"][k] = {"step": trainer.step_count, "exp_avg": trainer.m[o:o + num].view(m._W[k].shape).cpu().clone(),
                           "exp_avg_sq": trainer.v[o:o + num].view(m._W[k].shape).cpu().clone()}
    assert n <= trainer.m.numel()
    return {"network_weights": sd, "optimizer_state": opt, "grad_scaler_state": None, "train_loss": list(train_losses),
            "current_epoch": epoch,
            # --- resume extras (not in the reference) ---
            "ema_weights": {k: v.detach().cpu().clone() for k, v in trainer.teacher.ema.state_dict().items()},
            "ema_decay": trainer.teacher.decay, "rng_state": trainer.gen.get_state().cpu(), "anatomask_amd_version": 1}


def save_checkpoint(path: str, trainer, train_losses: List[float], epoch: int) -> None:
    torch.save(reference_state(trainer, train_losses, epoch), path)


def load_checkpoint(path: str, trainer) -> int:
    """Resume: returns the next epoch index."""
    ck = torch.load(path, map_location="cpu", weights_only=False)
    m = trainer.model
    m.load_state_dict({k[len("module."):]: v for k, v in ck["network_weights"].items()})
    if "ema_weights" in ck:
        trainer.teacher.ema.load_state_dict(ck["ema_weights"])
        trainer.teacher.decay = ck.get("ema_decay", trainer.teacher.decay)
    m._ensure_flat(); trainer.teacher.ema._ensure_flat()
    opt = ck.get("optimizer_state") or {}
    for k, st in opt.get("state", {}).items():
        o, num = m._offs[k], m._W[k].numel()
        trainer.m[o:o + num].copy_(st["exp_avg"].reshape(-1)); trainer.v[o:o + num].copy_(st["exp_avg_sq"].reshape(-1))
    trainer.step_count = int(opt.get("step", 0))
    if "rng_state" in ck:
        trainer.gen.set_state(ck["rng_state"])
    return int(ck["current_epoch"]) + 1


def encoder_weights_for_finetuning(network_weights: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """What load_stunet_ssl_weights extracts (nnunetv2/run/load_pretrained_weights.py:66-106): keys containing 'encoder',
    stripped up to 'sp_cnn.' -> 'conv_blocks_context.{s}.{b}.{conv1,...}' of the full STUNet."""
    out = {}
    for k, v in network_weights.items():
        if "encoder" in k and "sp_cnn." in k:
            out[k.split("sp_cnn.", 1)[1]] = v
    return out
