"""Checkpoint writer / reader for the pretraining run and the hand-off to STUNet finetuning.

The reference saves, once per epoch (P/pretrain_AntoMask.py:472-479; plain SparK additionally '_head_best.pt' and 'val_loss',
P/pretrain.py:445-493):
    {'network_weights': model.state_dict()   # LocalDDP/DDP-wrapped -> keys prefixed 'module.'
     'optimizer_state': optimizer.state_dict(), 'grad_scaler_state': None, 'train_loss': [...], 'current_epoch': i}
and the finetuning loader (nnunetv2/run/load_pretrained_weights.py:66-106, load_stunet_ssl_weights) keeps the entries whose key
contains 'encoder', strips everything up to and including 'sp_cnn.', and load_state_dict(strict=False)s them into the full STUNet.

`optimizer_state` is written in torch.optim.AdamW.state_dict() layout -- index-keyed 'state' {i: {'step', 'exp_avg', 'exp_avg_sq'}}
and 'param_groups' whose 'params' are index lists, in the parameter order of get_param_groups (P/utils/lr_control.py:32-53: 'decay'
then 'no_decay', each in named_parameters order) -- so the reference's `optimizer.load_state_dict(ck['optimizer_state'])` accepts it.
What the reference forgot and a faithful resume needs goes under extra keys its loaders ignore: EMA teacher weights and decay, the
per-rank sampler RNG states, the data loaders' RandomState, the epoch-EMA of the loss.  Everything in the file is a tensor or a python
scalar / str / list / dict (no numpy objects): the reference's plain `torch.load(fname)` is weights_only=True from torch 2.6 on and would
reject the whole file otherwise (tests/test_host_api.py).  Files are written atomically (temp + rename).
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional

import torch


def _group_names(model) -> List[List[str]]:
    """parameter names per optimizer group, as get_param_groups orders them (dead densify[4] tensors included: the reference's
    optimizer holds them too, they simply never receive a gradient)."""
    decay, no_decay = [], []
    for name, p in model.named_parameters():
        (no_decay if (len(p.shape) == 1 or name.endswith(".bias") or any(k in name for k in ("cls_token", "pos_embed", "mask_token", "gamma")))
         else decay).append(name)
    return [decay, no_decay]


def optimizer_state_dict(trainer) -> Dict:
    """torch.optim.AdamW.state_dict() of the fused optimizer (flat exp_avg / exp_avg_sq buffers -> per-parameter tensors)."""
    m = trainer.model
    groups = _group_names(m)
    state, pg, idx = {}, [], 0
    for gi, names in enumerate(groups):
        ids = []
        for k in names:
            if k not in m._dead and trainer.step_count > 0:            # torch creates state lazily: parameters without a gradient have none
                o, num = m._offs[k], m._W[k].numel()
                state[idx] = {"step": torch.tensor(float(trainer.step_count)),
                              "exp_avg": trainer.m[o:o + num].view(m._W[k].shape).detach().cpu().clone(),
                              "exp_avg_sq": trainer.v[o:o + num].view(m._W[k].shape).detach().cpu().clone()}
            ids.append(idx)
            idx += 1
        pg.append({"lr": trainer.lr, "betas": tuple(trainer.betas), "eps": trainer.eps, "weight_decay": trainer.wd, "amsgrad": False,
                   "maximize": False, "foreach": None, "capturable": False, "differentiable": False, "fused": None,
                   "weight_decay_scale": 1.0 if gi == 0 else 0.0, "lr_scale": 1.0, "params": ids})
    return {"state": state, "param_groups": pg, "param_names": [n for g in groups for n in g]}


def reference_state(trainer, train_losses: List[float], epoch: int, val_losses: Optional[List[float]] = None, extra: Optional[Dict] = None) -> Dict:
    m = trainer.model
    ck = {"network_weights": {"module." + k: v.detach().cpu().clone() for k, v in m.state_dict().items()},
          "optimizer_state": optimizer_state_dict(trainer), "grad_scaler_state": None, "train_loss": list(train_losses),
          "current_epoch": epoch,
          # --- resume extras (not in the reference) ---
          "ema_weights": {k: v.detach().cpu().clone() for k, v in trainer.teacher.ema.state_dict().items()},
          "ema_decay": trainer.teacher.decay, "rng_state": trainer.gen.get_state().cpu(), "rng_rank": getattr(trainer, "rank", 0),
          "step_count": trainer.step_count, "anatomask_amd_version": FORMAT_VERSION}
    if val_losses is not None:
        ck["val_loss"] = list(val_losses)
    if extra:
        ck.update(extra)
    return ck


def save_checkpoint(path: str, trainer, train_losses: List[float], epoch: int, val_losses: Optional[List[float]] = None,
                    extra: Optional[Dict] = None) -> None:
    tmp = path + ".tmp"
    torch.save(reference_state(trainer, train_losses, epoch, val_losses, extra), tmp)
    os.replace(tmp, path)                              # a crash mid-write never destroys the previous 'latest'


FORMAT_VERSION = 3      # 3: tensors and python scalars only (loader / augmenter RandomStates as plain tuples); 2 pickled numpy RandomState tuples


def read_checkpoint(path: str) -> Dict:
    """The file as a dict, read ONCE (pretrain.py hands the same dict to the feed and to load_checkpoint).  ALWAYS through torch's
    restricted unpickler (`weights_only=True`): a file is data, never code.  Files of format 3 hold tensors and python scalars only and
    load as they are.  An older file of ours (format 2: numpy RandomState tuples inside `feed_state`) is re-read with the restricted
    unpickler allowed to rebuild plain numpy arrays (`_reconstruct`, `ndarray`, `dtype`, the uint32 dtype class -- nothing callable beyond
    array construction) and its feed state is dropped (the loaders restart from their seeds).  Anything else -- a pickle that names
    any other global -- is refused with torch's own UnpicklingError; there is no `weights_only=False` path."""
    import pickle
    try:
        return torch.load(path, map_location="cpu", weights_only=True)
    except pickle.UnpicklingError as first:
        import numpy as np
        import warnings
        # numpy 2.x spells the module `numpy._core`, numpy 1.x `numpy.core`: a format-2 file names whichever its WRITER had, so the
        # function is registered under both spellings ((callable, "module.name") tuples are torch's form for that)
        core = getattr(np, "_core", None) or getattr(np, "core", None)
        rec = getattr(getattr(core, "multiarray", None), "_reconstruct", None)
        if rec is None:
            raise first
        allow = [(rec, "numpy._core.multiarray._reconstruct"), (rec, "numpy.core.multiarray._reconstruct"), np.ndarray, np.dtype,
                 type(np.dtype(np.uint32))]
        try:
            with torch.serialization.safe_globals(allow):
                ck = torch.load(path, map_location="cpu", weights_only=True)
        except Exception:                                  # (whatever the second attempt dies of, the file's real problem is the first error)
            raise first
        if not isinstance(ck, dict) or int(ck.get("anatomask_amd_version", 0)) >= FORMAT_VERSION:
            raise first
        warnings.warn(f"{path}: a format-{ck.get('anatomask_amd_version', 0)} checkpoint of this package (numpy objects in its loader state); "
                      "its loader state is dropped, the loaders restart from their seeds")
        ck.pop("feed_state", None)
        return ck


def load_checkpoint(path, trainer, rank: int = 0) -> Dict:
    """Resume.  Returns the checkpoint dict (its 'current_epoch' + 1 is the next epoch).  The sampler RNG is restored only on the rank
    that saved it; every other rank re-derives its stream from (seed, rank, epoch) -- restoring rank 0's state everywhere would make
    all ranks draw identical masks, which a fresh run (seed 4321 + rank) never does."""
    ck = path if isinstance(path, dict) else read_checkpoint(path)      # (a dict: the caller has read the file already)
    m = trainer.model
    m.load_state_dict({k[len("module."):]: v for k, v in ck["network_weights"].items()})
    if "ema_weights" in ck:
        trainer.teacher.ema.load_state_dict(ck["ema_weights"])
        trainer.teacher.decay = ck.get("ema_decay", trainer.teacher.decay)
    m._ensure_flat(); trainer.teacher.ema._ensure_flat()
    opt = ck.get("optimizer_state") or {}
    names = opt.get("param_names") or [n for g in _group_names(m) for n in g]
    for i, st in opt.get("state", {}).items():
        k = names[int(i)] if not isinstance(i, str) or i.isdigit() else i
        if k in m._dead:
            continue
        o, num = m._offs[k], m._W[k].numel()
        trainer.m[o:o + num].copy_(st["exp_avg"].reshape(-1)); trainer.v[o:o + num].copy_(st["exp_avg_sq"].reshape(-1))
    trainer.step_count = int(ck.get("step_count", max([int(float(s["step"])) for s in opt.get("state", {}).values()], default=0)))
    trainer.reset_guard()                              # (the non-finite guard counts optimizer calls from here)
    if "rng_state" in ck:
        if rank == int(ck.get("rng_rank", 0)):
            trainer.gen.set_state(ck["rng_state"])
        else:
            trainer.gen.manual_seed(4321 + rank + 7919 * (int(ck["current_epoch"]) + 1))
    return ck


def peek_extra(path, key: str):
    """one extra entry of a checkpoint (e.g. 'feed_state') without touching a trainer; `path` may be the dict read_checkpoint returned."""
    return (path if isinstance(path, dict) else read_checkpoint(path)).get(key)


def encoder_weights_for_finetuning(network_weights: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """What load_stunet_ssl_weights extracts (nnunetv2/run/load_pretrained_weights.py:66-106): keys containing 'encoder',
    stripped up to the LAST 'sp_cnn.' -> 'conv_blocks_context.{s}.{b}.{conv1,...}' of the full STUNet."""
    out = {}
    for k, v in network_weights.items():
        if "encoder" in k:
            out[k.split("sp_cnn.")[-1]] = v
    return out
