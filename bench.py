#!/usr/bin/env python
"""Benchmark of the AnatoMask pretraining step on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W
  (N > 1: launched by torch.distributed.run, one rank per GPU, RCCL gradient all-reduce)

Workload (BASELINE.json configs[1]/[2]): STUNet-B AnatoMask, 128^3 patch, mask_ratio 0.6, bf16 storage /
bf16 MFMA with fp32 accumulation, fp32 master weights + AdamW + EMA; one "step" = the full iteration of
P/pretrain_AntoMask.py:418-441 (teacher fwd, sampler, student fwd, loss, bwd, clip, AdamW, EMA) on a batch
of B synthetic N(0,1) volumes already resident in HBM.  value = volumes/s over all ranks (weak scaling).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

# SURVEY.md 8(d) / BASELINE.md 3: algorithmic bytes of the student sparse-encoder forward, bf16, per volume
ENC_FWD_ALGO_BYTES = {"B": 1105.3e6}
HBM_PEAK = 8.0e12          # MI355X_MICROARCH.md: 8 TB/s spec (6.29 TB/s measured copy ceiling)
MFMA_BF16_PEAK = 2.5e15    # dense bf16


def time_kernel(fn, iters=20, warm=3):
    """average duration (s) of one launch, HIP events on the stream the kernels run on (torch's current stream)."""
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / iters


def dominant_kernel_roofline(B, dev):
    """The dominant kernel of the step (profiles/: conv_igemm bf16 4x8x8 brick, 64 output channels) on its largest
    instance: decoder level-3 conv 64->64 at 128^3 (P/decoder3D.py:20).  Bound: MFMA (AI ~ 1700 flop/B)."""
    from anatomask_amd import ops
    C = 64
    x = torch.randn(B, 128, 128, 128, C, device=dev).to(torch.bfloat16)
    w = (torch.randn(C, C, 3, 3, 3, device=dev) * 0.02)
    wp = ops.pack_weight(w, torch.bfloat16, False, False)
    y = torch.empty_like(x)
    t = time_kernel(lambda: ops.conv3d(ops.CONV_FWD, x, wp, None, (128, 128, 128), 3, 1, out=y))
    flops = 2.0 * B * 128 ** 3 * C * C * 27
    achieved = flops / t / 1e12
    # HBM bytes per launch from the PMC counters of the SAME launch shape (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate
    # passes, FETCH_SIZE doubled per the gfx950 correction): profiles/r01_pmc_traffic.md, measured at B=2 -> linear in B
    traffic = (605.5e6 + 536.9e6) * B / 2
    return {"bound": "mfma", "kernel": "conv_igemm_kernel<bf16,4,4,16,4,11> (decoder conv3 64->64 @128^3)", "achieved": round(achieved, 2),
            "peak": MFMA_BF16_PEAK / 1e12, "unit": "TFLOP/s", "frac": round(achieved * 1e12 / MFMA_BF16_PEAK, 4),
            "traffic": traffic, "traffic_source": "profiles/r01_pmc_traffic.md (rocprofv3 --pmc, scaled from B=2)",
            "algorithmic_bytes": (2 * 128 ** 3 * C * 2 * B) + 27 * C * C * 2, "launch_ms": round(t * 1e3, 4), "flop_per_launch": flops}


def encoder_forward_hbm(model, x, dev):
    """North-star figure: achieved algorithmic HBM GB/s of the STUDENT SPARSE-ENCODER FORWARD at 128^3 bf16
    (algorithmic bytes per SURVEY.md 8d) from HIP events around that part of the forward."""
    from anatomask_amd import engine, ops
    B = x.shape[0]
    L = model.spec.fmap[0] * model.spec.fmap[1] * model.spec.fmap[2]
    k = torch.rand(B, L, device=dev)
    mk = ops.mask_sampler(torch.zeros(B, L, device=dev), k, model.len_keep, 0)
    mi = ops.MaskInfo(mk.view(B, *model.spec.fmap))
    xs = x[:, 0].contiguous()

    def enc_only():
        engine.forward(model.spec, model._W, model._pack, xs, mi, True, None, encoder_only=True)
    t = time_kernel(enc_only, iters=10, warm=2)
    algo = ENC_FWD_ALGO_BYTES["B"] * B
    gbs = algo / t / 1e9
    return {"what": "student sparse-encoder forward, STUNet-B 128^3 bf16, mask 0.6", "ms": round(t * 1e3, 3),
            "algorithmic_bytes": algo, "achieved_GBps": round(gbs, 1), "frac_of_hbm_peak": round(gbs * 1e9 / HBM_PEAK, 4),
            "flop": 114.0e9 * B, "achieved_TFLOPs": round(114.0e9 * B / t / 1e12, 2)}


def cpu_baseline(state_dict_cpu, spec_kw):
    """The CPU oracle (port of the reference step, pinned against it by tests/golden) on the host cores:
    ONE full AnatoMask step of the same workload at B=1 (teacher fwd + sampler + student fwd/bwd + clip + AdamW + EMA)."""
    from oracle import anatomask_oracle as O
    n = min(32, os.cpu_count() or 1)      # measured on the MI355X host: 8/16/32/64 threads -> 5.4/4.8/4.1/5.8 s per forward; 256 oversubscribes 10x
    torch.set_num_threads(n)
    cfg = O.Config(spec_kw["dims"], spec_kw["depth"], spec_kw["width"], (128, 128, 128), 0.6)
    st = O.StepState(cfg, state_dict_cpu)
    x = torch.randn(1, 1, 128, 128, 128, generator=torch.Generator().manual_seed(1234))
    mask1 = O.random_mask(cfg, 1, torch.Generator().manual_seed(4321))
    keys = torch.rand(1, cfg.L, generator=torch.Generator().manual_seed(4322))
    t0 = time.time()
    o = O.train_step(st, x, mask1, keys, 500, 999, 1e-4, 0.9995)
    dt = time.time() - t0
    return {"value": round(1.0 / dt, 5), "unit": "volumes/s", "cores": n, "kind": "port",
            "sample": f"1 full step, B=1, STUNet-B 128^3 fp32, torch-CPU oracle, {dt:.1f} s, loss {o['loss']:.4f}"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=int(os.environ.get("AM_BENCH_BATCH", "8")),
                    help="volumes per GPU per step (SURVEY.md 8d C2: chosen to fill the GPU; the reference default is 4: 5 %% slower)")
    ap.add_argument("--size", default="B")
    ap.add_argument("--patch", type=int, default=128)
    ap.add_argument("--mask-ratio", type=float, default=0.6)
    ap.add_argument("--recompute", action="store_true")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    a = ap.parse_args()

    import torch.distributed as dist
    from anatomask_amd import modules as M
    from anatomask_amd.trainer import AnatoMaskTrainer

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    assert a.gpus == world, f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run"

    kw = M.STUNET_CONFIGS[a.size]
    dtype = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    torch.manual_seed(0)                                            # identical init on every rank (+ broadcast in the trainer)
    model = M.build_spark(kw["dims"], kw["depth"], kw["width"], (a.patch,) * 3, a.mask_ratio, compute_dtype=dtype, recompute=a.recompute)
    sd_cpu = ({k: v.clone() for k, v in model.state_dict().items()}
              if (rank == 0 and world == 1 and not a.no_cpu_baseline and a.size == "B" and a.patch == 128) else None)
    model = model.to(dev)
    tr = AnatoMaskTrainer(model, lr=1e-4, total_epochs=1000, seed=4321 + rank)
    tr.set_epoch(500)
    x = torch.randn(a.batch, 1, a.patch, a.patch, a.patch, device=dev, generator=torch.Generator(device=dev).manual_seed(1234 + rank))

    for _ in range(a.warmup):
        out = tr.step(x, epoch=500)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        out = tr.step(x, epoch=500)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = tt.item()
    loss = out["loss"].item()
    assert loss == loss and abs(loss) < 1e6, f"non-finite loss {loss}"   # the reference's finite-loss guard (:443-446)

    res = None
    if rank == 0:
        res = {"metric": f"pretrain volumes/sec @{a.patch}^3 patch mask={a.mask_ratio}", "value": round(a.batch * world * a.steps / dt, 4),
               "unit": "volumes/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 3),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
               "config": {"workload": f"STUNet-{a.size} AnatoMask step, {a.patch}^3 patch, mask_ratio {a.mask_ratio}, {a.dtype} storage/MFMA + fp32 master",
                          "per_gpu_batch": a.batch, "global_batch": a.batch * world, "parallelism": f"dp{world}",
                          "step": "teacher fwd + sampler + student fwd + loss + bwd + clip + AdamW + EMA"},
               "final_loss": round(loss, 5)}
        if not a.no_roofline and a.size == "B" and a.patch == 128:
            res["roofline"] = dominant_kernel_roofline(a.batch, dev)
            res["encoder_fwd_hbm"] = encoder_forward_hbm(model, x, dev)
    if world > 1:
        dist.barrier()
    if rank == 0:
        if sd_cpu is not None:
            del tr, model, x
            torch.cuda.empty_cache()
            try:
                res["cpu_baseline"] = cpu_baseline(sd_cpu, kw)
            except Exception as e:                                   # e.g. host OOM: report, do not fail the bench
                res["cpu_baseline"] = {"value": None, "unit": "volumes/s", "cores": os.cpu_count(), "kind": "port", "sample": f"failed: {e!r}"}
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
