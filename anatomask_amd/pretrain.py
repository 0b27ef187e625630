#!/usr/bin/env python
"""Pretraining driver on the HIP engine: the loops of P/pretrain_AntoMask.py:371-479 (AnatoMask, single GPU),
P/pretrain_AnatoMask_DDP.py:421-513 (one process per GPU, RCCL) and P/pretrain.py:376-493 (plain SparK: per-epoch validation pass
in eval mode, best + latest checkpoints) with the reference's hyper-parameters as defaults, lifted from in-script literals into
flags (SURVEY.md 5 "Config").

    python -m anatomask_amd.pretrain --model B --input-size 112 112 128 --batch-size 4 --data /path/to/preprocessed --out run1
    python -m anatomask_amd.pretrain --gpus 8 ...          # launches itself: one rank per GPU (anatomask_amd.launch), or run under torchrun

Data (`--data DIR`, an nnU-Net v2 preprocessed folder: <case>.npy|.npz + <case>.pkl): loader threads crop the ENLARGED patch
(get_patch_size) with foreground oversampling 0.33 straight into pinned buffers (anatomask_amd.data), a copy stream moves them to
the GPU one batch ahead, and the spatial augmentation of the reference's train transforms (rotation / scaling p 0.2, order-3
interpolation; mirroring p 0.5 per axis) runs on the device.  Without --data the batches are synthetic N(0,1).
"""
import argparse
import math
import os
import sys
import time

import torch
import torch.distributed as dist

from . import checkpoint, launch
from .data import (DeviceAugmenter, DeviceFeed, PatchLoader3D, PinnedPool, PrefetchLoader, PreprocessedDataset, ROTATION_FOR_DA, SpatialAugmenter,
                   get_patch_size)
from .modules import STUNET_CONFIGS, build_spark, linear_warmup_cosine_lrs
from .trainer import AnatoMaskTrainer


def synthetic_batches(batch, size, seed):
    g = torch.Generator().manual_seed(seed)
    while True:
        yield {"data": torch.randn(batch, 1, *size, generator=g)}


def split_cases(keys, val_fraction=0.2, seed=12345):
    """train / validation split of the case list (the reference: sklearn train_test_split(test_size 0.2, random_state 12345) on the
    sorted keys, P/pretrain.py:283-285 -- restated as a seeded permutation; the exact sklearn shuffle is not reproduced)."""
    import numpy as np
    keys = sorted(keys)
    perm = np.random.RandomState(seed).permutation(len(keys))
    n_val = max(1, int(round(len(keys) * val_fraction))) if len(keys) > 1 else 0
    val = sorted(keys[i] for i in perm[:n_val])
    return [k for k in keys if k not in set(val)], val


def rng_state_to_plain(rs) -> dict:
    """numpy RandomState (MT19937) state as torch tensors / python scalars only: the checkpoint must stay loadable by the
    reference's hand-off, a plain `torch.load(fname)` (nnunetv2/run/load_pretrained_weights.py), which is weights_only=True
    from torch 2.6 on and rejects numpy objects anywhere in the file."""
    name, keys, pos, has_gauss, cached = rs if isinstance(rs, tuple) else rs.get_state()      # (a tuple: a snapshot taken under the loader's lock)
    return {"name": str(name), "keys": torch.from_numpy(keys.astype("int64")), "pos": int(pos), "has_gauss": int(has_gauss),
            "cached_gaussian": float(cached)}


def rng_state_from_plain(rs, st: dict):
    import numpy as np
    rs.set_state((st["name"], np.asarray(st["keys"].cpu().numpy(), dtype=np.uint32), int(st["pos"]), int(st["has_gauss"]),
                  float(st["cached_gaussian"])))


class Feed:
    """loader threads -> pinned pool -> copy stream -> device augmentation; next(feed) is a (B,1,*input_size) device tensor.
    `state` (what state() returned at a checkpoint) is applied to the loaders' generators BEFORE the prefetch threads start; the
    batches that were prefetched but not consumed when the checkpoint was written are not replayed (the workers run up to
    num_cached + workers batches ahead of the step, and state() samples their generators while they run)."""

    def __init__(self, folder, cases, batch, input_size, dev, rank, workers, augment: bool, seed: int, state=None):
        ds = PreprocessedDataset(folder, cases)
        rot = ROTATION_FOR_DA
        enl = tuple(int(v) for v in get_patch_size(tuple(input_size), rot, rot, rot, (0.85, 1.25))) if augment else tuple(input_size)
        self.pool = PinnedPool((batch, 1, *enl), n=6 + 2 + workers)
        self.loaders = {w: PatchLoader3D(ds, batch, enl, 0.33, seed=seed + 1000 * rank + w, final_patch_size=tuple(input_size), pool=self.pool)
                        for w in range(workers)}
        self.aug = DeviceAugmenter(SpatialAugmenter(tuple(input_size), seed=seed + 77 + rank,
                                                    p_rot=0.2 if augment else 0.0, p_scale=0.2 if augment else 0.0,
                                                    mirror_axes=(0, 1, 2) if augment else ()))
        self.load_state(state)
        # loader threads next to this rank's GPU (its NUMA node's CPUs), when the platform says which those are
        from .data import gpu_numa_cpus
        cpus = gpu_numa_cpus(dev.index if getattr(dev, "index", None) is not None else 0) if getattr(dev, "type", "cpu") == "cuda" else None
        self.pf = PrefetchLoader(lambda w: self.loaders[w], n_workers=workers, num_cached=6, cpus=cpus)
        self.feed = DeviceFeed(self.pf, dev)

    def __next__(self):
        return self.aug(next(self.feed))

    def state(self):
        # (each loader's generator is sampled under that loader's lock, i.e. between two batches of its prefetch thread)
        return {"loader_rng": {int(w): rng_state_to_plain(ld.rng_snapshot()) for w, ld in self.loaders.items()}, "aug_rng": rng_state_to_plain(self.aug.aug.rs)}

    def load_state(self, st):
        for w, s_ in (st or {}).get("loader_rng", {}).items():
            if int(w) in self.loaders:
                rng_state_from_plain(self.loaders[int(w)].rs, s_)
        if st and "aug_rng" in st:
            rng_state_from_plain(self.aug.aug.rs, st["aug_rng"])

    def close(self):
        self.pf.close()


def gather_feed_states(feed, rank: int, world: int) -> dict:
    """{rank: feed.state()} of EVERY rank on every rank (only rank 0 writes the checkpoint; a resumed run must not restart ranks
    1..N-1 from their initial seeds and replay epoch 0's crops)."""
    mine = feed.state()
    if world <= 1:
        return {rank: mine}
    states = [None] * world
    dist.all_gather_object(states, mine)
    return {r: s_ for r, s_ in enumerate(states)}


def default_workers(ranks_on_node: int) -> int:
    """loader threads per rank: the host's cores shared by the ranks of this node (one core per rank stays with the launch thread)."""
    return max(1, min(8, (os.cpu_count() or 8) // max(1, ranks_on_node) - 1))


def all_ranks_finite(value: float, dev, world: int) -> bool:
    """every rank stops together: a rank that exits alone leaves its peers hanging in the next all-reduce."""
    ok = torch.tensor([1.0 if math.isfinite(value) else 0.0], device=dev)
    if world > 1:
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    return bool(ok.item() > 0)


def first_nonfinite_step(trainer, world: int):
    """P/pretrain_AntoMask.py:441-446 looks at `loss.item()` after EVERY step; here the look is on the device (am_adamw_ema's guard: a
    step whose loss / gradient norm is not finite changes nothing, and neither does any step after it) and the host reads the latch once
    per epoch.  Returns None or the step; all ranks agree (the all-reduced gradient is non-finite everywhere; MIN over ranks for safety)."""
    s = trainer.nonfinite_step()
    if world > 1:
        t = torch.tensor([float(s) if s is not None else float("inf")], device=trainer.guard.device)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        v = float(t.item())
        s = None if math.isinf(v) else int(v)
    return s


class RunLog:
    """The run's text log and per-epoch series (P/pretrain_AntoMask.py:244-275 `print_to_log_file`: a `training_log_<Y>_<M>_<D>_<h>_<m>_<s>.txt`
    in the output folder, every line stamped, writes retried on IOError; `nnUNetLogger`'s dict of per-epoch lists,
    nnunetv2/training/logging/nnunet_logger.py:9-52, logged at :381,453,467).  Rank 0 owns the file; every rank may `say` (errors)."""
    SERIES = ("train_losses", "val_losses", "lrs", "epoch_start_timestamps", "epoch_end_timestamps")

    def __init__(self, folder: str, rank: int, series=None):
        from datetime import datetime
        t = datetime.now()
        self.rank = rank
        self.path = os.path.join(folder, "training_log_%d_%d_%d_%02d_%02d_%02d.txt" % (t.year, t.month, t.day, t.hour, t.minute, t.second))
        self.series = {k: list((series or {}).get(k, [])) for k in self.SERIES}

    def say(self, *args, to_file=None):
        from datetime import datetime
        line = " ".join(str(a) for a in (f"{datetime.now()}:", *args))
        if self.rank == 0 if to_file is None else to_file:
            for attempt in range(5):
                try:
                    with open(self.path, "a+") as f:
                        f.write(line + "\n")
                    break
                except IOError as e:
                    print(f"{datetime.now()}: failed to log: {e}", flush=True)
                    time.sleep(0.5)
        print(line, flush=True)

    def log(self, key: str, value, epoch: int):
        """nnUNetLogger.log: one value per epoch; a repeated epoch overwrites its entry."""
        s = self.series[key]
        if len(s) == epoch:
            s.append(value)
        elif len(s) > epoch:
            s[epoch] = value
        else:
            s.extend([None] * (epoch - len(s)) + [value])


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1, help="> 1 without a launcher: start one rank per GPU (anatomask_amd.launch)")
    ap.add_argument("--model", default="B", choices=list(STUNET_CONFIGS))
    ap.add_argument("--input-size", type=int, nargs=3, default=[112, 112, 128])       # P/pretrain_AntoMask.py:209
    ap.add_argument("--mask-ratio", type=float, default=0.6)                          # :215
    ap.add_argument("--epochs", type=int, default=1000)                               # :228
    ap.add_argument("--iters-per-epoch", type=int, default=250)
    ap.add_argument("--val-iters", type=int, default=50)
    ap.add_argument("--batch-size", type=int, default=4, help="per GPU (:229)")
    ap.add_argument("--lr", type=float, default=1e-4); ap.add_argument("--weight-decay", type=float, default=1e-5)
    ap.add_argument("--clip", type=float, default=12.0); ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32", "f32s"],
                    help="storage / matrix-core arithmetic: bf16; f32 = exact fp32 products (the reference's AMP = False, P/pretrain_AntoMask.py:239, at "
                         "1/16 of the bf16 matrix rate); f32s = fp32 storage, products from bf16 hi / lo splits (16 significant bits per operand, 2.6x f32)")
    ap.add_argument("--data", default=None); ap.add_argument("--out", default="anatomask_run")
    ap.add_argument("--workers", type=int, default=0,
                    help="loader threads per rank; 0 = min(8, host cores // ranks on this node - 1): 8 ranks x 8 threads on one host is not free")
    ap.add_argument("--no-augment", action="store_true")
    ap.add_argument("--resume", default=None)
    ap.add_argument("--plain-spark", action="store_true", help="plain SparK baseline (P/pretrain.py): random mask, no teacher, validation + best ckpt")
    ap.add_argument("--sync-bn", action="store_true", help="SyncBatchNorm statistics in the decoder (P/pretrain_DDP.py:225)")
    ap.add_argument("--deterministic", action="store_true",
                    help="bit-reproducible steps: ordered folds instead of fp32 atomics in the weight-gradient reductions (+2 %% step time)")
    a = ap.parse_args(argv)
    if a.gpus > 1 and not launch.launched():
        sys.exit(launch.self_launch(a.gpus, "-m", ["anatomask_amd.pretrain", *(argv if argv is not None else sys.argv[1:])]))

    world = int(os.environ.get("WORLD_SIZE", "1")); rank = int(os.environ.get("RANK", "0")); local = int(os.environ.get("LOCAL_RANK", "0"))
    workers = a.workers if a.workers > 0 else default_workers(int(os.environ.get("LOCAL_WORLD_SIZE", str(world))))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)
    os.makedirs(a.out, exist_ok=True)
    torch.manual_seed(0)
    kw = STUNET_CONFIGS[a.model]
    model = build_spark(kw["dims"], kw["depth"], kw["width"], tuple(a.input_size), a.mask_ratio,
                        compute_dtype=torch.bfloat16 if a.dtype == "bf16" else torch.float32, sbn=a.sync_bn).to(dev)
    trainer = AnatoMaskTrainer(model, lr=a.lr, weight_decay=a.weight_decay, clip=a.clip, total_epochs=a.epochs, seed=4321 + rank,
                               self_distill=not a.plain_spark, deterministic_wgrad=a.deterministic, f32_split=a.dtype == "f32s")
    trainer.rank = rank
    lrs = linear_warmup_cosine_lrs(a.epochs, a.lr, a.warmup, 1e-6)                    # :359
    feed = val_feed = None
    if a.data:
        if rank == 0:                                  # nnU-Net's unpack_dataset: .npz -> .npy once, then memory-mapped reads
            n_unpacked = PreprocessedDataset(a.data).unpack()
            if n_unpacked:
                print(f"unpacked {n_unpacked} .npz cases to .npy", flush=True)
        if world > 1:
            dist.barrier()
        tr_keys, val_keys = split_cases(PreprocessedDataset(a.data).keys()) if a.plain_spark else (PreprocessedDataset(a.data).keys(), [])
        resumed = None
        if a.resume:                                   # the loaders' generators are restored before their threads start
            ck_resume = checkpoint.read_checkpoint(a.resume)       # the file is read ONCE per rank (weights, EMA and Adam state ride in it)
            resumed = (checkpoint.peek_extra(ck_resume, "feed_state") or {}).get(rank)
        feed = Feed(a.data, tr_keys, a.batch_size, a.input_size, dev, rank, workers, not a.no_augment, seed=1000, state=resumed)
        if a.plain_spark and val_keys:
            val_feed = Feed(a.data, val_keys, a.batch_size, a.input_size, dev, rank, max(1, workers // 4), False, seed=5000)
    else:
        gen = synthetic_batches(a.batch_size, a.input_size, 1000 + rank)
        feed = type("Syn", (), {"__next__": lambda s: next(gen)["data"].to(dev, non_blocking=True), "state": lambda s: {}, "load_state": lambda s, st: None,
                                "close": lambda s: None})()
    start, epoch_loss, val_loss, ema_loss, best_val = 0, [], [], None, 1e9
    ck_resume = locals().get("ck_resume")
    series = None
    if a.resume:
        ck = checkpoint.load_checkpoint(ck_resume if ck_resume is not None else a.resume, trainer, rank)
        ck_resume = None
        start = int(ck["current_epoch"]) + 1
        epoch_loss, val_loss, ema_loss = list(ck.get("train_loss", [])), list(ck.get("val_loss", [])), ck.get("ema_loss")
        best_val = ck.get("best_val_loss", best_val)
        series = ck.get("logging")
    log = RunLog(a.out, rank, series)
    log.say(f"anatomask_amd pretraining: STUNet-{a.model} {tuple(a.input_size)} batch {a.batch_size} x {world} rank(s), mask ratio {a.mask_ratio}, "
            f"{a.dtype}, {'plain SparK' if a.plain_spark else 'AnatoMask'}, epochs {start}..{a.epochs - 1}, {a.iters_per_epoch} iterations each"
            + (f", resumed from {a.resume}" if a.resume else ""))
    for i in range(start, a.epochs):
        trainer.set_epoch(i); trainer.lr = lrs[i]                                     # :383-386, :452
        t0, acc = time.time(), torch.zeros(1, device=dev)
        log.log("epoch_start_timestamps", t0, i); log.log("lrs", lrs[i], i)            # :381
        for _ in range(a.iters_per_epoch):
            out = trainer.step(next(feed), epoch=i)
            acc += out["loss"]
        loss = acc.item() / a.iters_per_epoch                                          # ONE host sync per epoch
        bad = first_nonfinite_step(trainer, world)                                     # :441-446: the per-step check, made on the device
        if bad is not None or not all_ranks_finite(loss, dev, world):                  # on every rank together
            log.say(f"[rk{rank:02d}] Loss is {loss}" + (f" (first non-finite step: {bad}; weights, Adam state, teacher and BatchNorm buffers "
                    f"are those before it)" if bad is not None else "") + ", stopping training!", to_file=True)
            sys.exit(-1)
        epoch_loss.append(loss)
        ema_loss = loss if ema_loss is None else 0.9 * ema_loss + 0.1 * loss           # :456-461
        log.log("train_losses", loss, i); log.log("epoch_end_timestamps", time.time(), i)          # :453, :467
        extra = {"ema_loss": ema_loss, "feed_state": gather_feed_states(feed, rank, world), "logging": log.series}
        if a.plain_spark and val_feed is not None:                                     # P/pretrain.py:426-463: eval() pass, no grad, BN on running stats
            vacc = torch.zeros(1, device=dev)
            for _ in range(a.val_iters):
                vacc += trainer.eval_loss(next(val_feed))
            v = vacc.item() / a.val_iters
            val_loss.append(v)
            log.log("val_losses", v, i)
            if v < best_val and rank == 0:
                best_val = v
                checkpoint.save_checkpoint(os.path.join(a.out, f"STUNet_{a.model}_head_best.pt"), trainer, epoch_loss, i, val_loss,
                                           dict(extra, best_val_loss=best_val))
        if rank == 0:
            dt = time.time() - t0
            log.say(f"Epoch {i} lr {lrs[i]:.2e} ema_decay {trainer.teacher.decay:.5f} train_loss {loss:.4f} (ema {ema_loss:.4f})"
                    + (f" val_loss {val_loss[-1]:.4f} (best {best_val:.4f})" if val_loss else "")
                    + f" Epoch time: {dt:.2f} s, {a.iters_per_epoch * a.batch_size * world / dt:.1f} volumes/s")          # :468-470
            checkpoint.save_checkpoint(os.path.join(a.out, f"STUNet_{a.model}_head_latest.pt"), trainer, epoch_loss, i,
                                       val_loss if a.plain_spark else None, dict(extra, best_val_loss=best_val))   # :472-479
    feed.close()
    if val_feed is not None:
        val_feed.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
