#!/usr/bin/env python
"""AnatoMask pretraining driver on the HIP engine: the loop of P/pretrain_AntoMask.py:371-479 (single GPU) and
P/pretrain_AnatoMask_DDP.py:421-513 (torchrun, one process per GPU, RCCL) with the reference's hyper-parameters as defaults,
lifted from in-script literals into flags (SURVEY.md 5 "Config").

    python -m anatomask_amd.pretrain --model B --input-size 112 112 128 --batch-size 4 --data /path/with/npy --out run1
    python -m torch.distributed.run --nproc-per-node 8 -m anatomask_amd.pretrain ...

Data: any iterator yielding nnU-Net style batches {'data': float32 (B,1,H,W,D)} works (that is all the step consumes,
P/pretrain_AntoMask.py:390-392).  `--data DIR`: an nnU-Net v2 preprocessed folder (<case>.npy|.npz + <case>.pkl) goes through
`anatomask_amd.data` (foreground oversampling 0.33, mirroring, as the reference's loader); a folder of bare .npy volumes gets
random crops; without --data the batches are synthetic.
"""
import argparse
import glob
import math
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

from . import checkpoint
from .data import PatchLoader3D, PreprocessedDataset
from .modules import STUNET_CONFIGS, build_spark, ema_decay_for_epoch, linear_warmup_cosine_lrs
from .trainer import AnatoMaskTrainer


def npy_crop_batches(files, batch, size, iters, seed):
    """Random crops of preprocessed volumes (zero-padded when a volume is smaller), the role nnUNetDataLoader3D plays."""
    rs = np.random.RandomState(seed)
    for _ in range(iters):
        out = np.zeros((batch, 1, *size), dtype=np.float32)
        for b in range(batch):
            v = np.load(files[rs.randint(len(files))], mmap_mode="r")
            v = v[0] if v.ndim == 4 else v
            lo = [rs.randint(0, max(s - c, 0) + 1) for s, c in zip(v.shape, size)]
            crop = np.asarray(v[lo[0]:lo[0] + size[0], lo[1]:lo[1] + size[1], lo[2]:lo[2] + size[2]], dtype=np.float32)
            out[b, 0, :crop.shape[0], :crop.shape[1], :crop.shape[2]] = crop
        yield {"data": torch.from_numpy(out).pin_memory()}


def synthetic_batches(batch, size, iters, seed):
    g = torch.Generator().manual_seed(seed)
    for _ in range(iters):
        yield {"data": torch.randn(batch, 1, *size, generator=g)}


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="B", choices=list(STUNET_CONFIGS))
    ap.add_argument("--input-size", type=int, nargs=3, default=[112, 112, 128])       # P/pretrain_AntoMask.py:209
    ap.add_argument("--mask-ratio", type=float, default=0.6)                          # :215
    ap.add_argument("--epochs", type=int, default=1000)                               # :228
    ap.add_argument("--iters-per-epoch", type=int, default=250)
    ap.add_argument("--batch-size", type=int, default=4, help="per GPU (:229)")
    ap.add_argument("--lr", type=float, default=1e-4); ap.add_argument("--weight-decay", type=float, default=1e-5)
    ap.add_argument("--clip", type=float, default=12.0); ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--data", default=None); ap.add_argument("--out", default="anatomask_run")
    ap.add_argument("--resume", default=None)
    ap.add_argument("--plain-spark", action="store_true", help="plain SparK baseline (P/pretrain.py): random mask, no teacher")
    a = ap.parse_args(argv)

    world = int(os.environ.get("WORLD_SIZE", "1")); rank = int(os.environ.get("RANK", "0")); local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)
    os.makedirs(a.out, exist_ok=True)
    torch.manual_seed(0)
    kw = STUNET_CONFIGS[a.model]
    model = build_spark(kw["dims"], kw["depth"], kw["width"], tuple(a.input_size), a.mask_ratio,
                        compute_dtype=torch.bfloat16 if a.dtype == "bf16" else torch.float32).to(dev)
    trainer = AnatoMaskTrainer(model, lr=a.lr, weight_decay=a.weight_decay, clip=a.clip, total_epochs=a.epochs, seed=4321 + rank,
                               self_distill=not a.plain_spark)
    lrs = linear_warmup_cosine_lrs(a.epochs, a.lr, a.warmup, 1e-6)                    # :359
    start = checkpoint.load_checkpoint(a.resume, trainer) if a.resume else 0
    files = sorted(glob.glob(os.path.join(a.data, "*.npy"))) if a.data else None
    nnunet_feed = None
    if a.data and glob.glob(os.path.join(a.data, "*.pkl")):          # nnU-Net v2 preprocessed folder
        nnunet_feed = PatchLoader3D(PreprocessedDataset(a.data), a.batch_size, a.input_size, 0.33, seed=1000 + rank)
    epoch_loss, ema_loss = [], None
    for i in range(start, a.epochs):
        trainer.set_epoch(i); trainer.lr = lrs[i]                                     # :383-386, :452
        if nnunet_feed is not None:
            it = (next(nnunet_feed) for _ in range(a.iters_per_epoch))
        else:
            it = (npy_crop_batches(files, a.batch_size, a.input_size, a.iters_per_epoch, 1000 * i + rank) if files
                  else synthetic_batches(a.batch_size, a.input_size, a.iters_per_epoch, 1000 * i + rank))
        t0, acc = time.time(), torch.zeros(1, device=dev)
        for batch in it:
            out = trainer.step(batch["data"].to(dev, non_blocking=True), epoch=i)
            acc += out["loss"]
        loss = acc.item() / a.iters_per_epoch                                          # ONE host sync per epoch
        if not math.isfinite(loss):                                                    # :443-446
            print(f"[rk{rank:02d}] Loss is {loss}, stopping training!", flush=True)
            sys.exit(-1)
        epoch_loss.append(loss)
        ema_loss = loss if ema_loss is None else 0.9 * ema_loss + 0.1 * loss           # :456-461
        if rank == 0:
            print(f"Epoch {i} lr {lrs[i]:.2e} ema_decay {trainer.teacher.decay:.5f} train loss {loss:.4f} (ema {ema_loss:.4f}) "
                  f"{time.time() - t0:.1f} s, {a.iters_per_epoch * a.batch_size * world / (time.time() - t0):.1f} volumes/s", flush=True)
            checkpoint.save_checkpoint(os.path.join(a.out, f"STUNet_{a.model}_head_latest.pt"), trainer, epoch_loss, i)   # :472-479
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
