#!/usr/bin/env python
"""Benchmark of the AnatoMask pretraining step on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W
  N > 1 without a launcher: this process (which has not touched the GPU) starts
      python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py <same flags>
  as a child, one rank per GPU, RCCL gradient all-reduce; under a launcher (RANK/WORLD_SIZE set) it is a rank.

Workload (BASELINE.json configs[1]/[2]): STUNet-B AnatoMask, 128^3 patch, mask_ratio 0.6, bf16 storage /
bf16 MFMA with fp32 accumulation, fp32 master weights + AdamW + EMA; one "step" = the full iteration of
P/pretrain_AntoMask.py:418-441 (teacher fwd, sampler, student fwd, loss, bwd, clip, AdamW, EMA) on a batch
of B synthetic N(0,1) volumes already resident in HBM.  value = volumes/s over all ranks (weak scaling).

Extra fields of the JSON line (besides the driver's contract):
  step_ms_median / step_ms_min      per-step durations from HIP events recorded after every step of the timed window
  value_with_h2d                    the same K steps fed from pinned host memory through the double-buffered copy stream
                                    (anatomask_amd.data.DeviceFeed; the reference's `inp.to(device, non_blocking=True)`): NOT `value`
  exchange (N > 1)                  RCCL rank count, gradient bytes per step, step time with the exchange disabled, exposed ms
  roofline                          dominant kernel (profiles/): isolated launch AND the same launch timed inside the step
  encoder_fwd_hbm                   north-star figure: student sparse-encoder forward vs the 8 TB/s HBM peak
  cpu_baseline                      the CPU oracle taking one B=1 step on the host cores (rank 0, N = 1 only)
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

# SURVEY.md 8(d) / BASELINE.md 3: algorithmic (bytes, flop) of the student sparse-encoder forward, bf16, per volume, at the patch /
# mask ratio the figure was derived for (size -> (patch, mask_ratio, bytes, flop))
ENC_FWD_ALGO = {"S": (48, 0.6, 36.1e6, 1.5e9), "B": (128, 0.6, 1105.3e6, 114.0e9), "L": (160, 0.7, 6135e6, 1716e9), "H": (192, 0.6, 30622e6, 14345e9)}
HBM_PEAK = 8.0e12          # MI355X_MICROARCH.md: 8 TB/s spec (6.29 TB/s measured copy ceiling)
# counter evidence of this round's tree, written by tools/pmc_k3.sh / tools/enc_traffic.py on the GPU box and committed under profiles/
PMC_K3_JSON = os.path.join("profiles", "r06_pmc_k3.json")
ENC_TRAFFIC_JSON = os.path.join("profiles", "r06_encoder_fwd_traffic.json")
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


def dominant_kernel_roofline(B, dev, tr, x, C=64, S=128, size="B"):
    """The dominant kernel of the step (profiles/r06_final_step_kernel_stats_b16_isolated.csv: conv_k3_kernel, the persistent 8-wave LDS-DMA kernel of
    the dense k3 s1 convolutions) on its largest instance: the first conv of the last decoder block, C -> C at S^3 (P/decoder3D.py:20;
    C = width / 8: 64 for STUNet-B, 128 for L, 192 for H), launched as the training step launches it: WITH the statistics epilogue
    (`want_partials=True`: the student's BatchNorm reads its sums from the conv).  Bound: MFMA (AI ~ 1700 flop/B at C = 64).
    `launch_ms`: the launch alone (20 back-to-back launches, HIP events on the launch stream); `launch_ms_in_step`: the SAME launch
    timed by HIP events where it sits inside the training step (student decoder, last block, first conv), median over 3 steps;
    `launch_ms_no_statistics`: the variant without the epilogue sums (the teacher's / the data gradient's launch)."""
    from anatomask_amd import ops
    xx = torch.randn(B, S, S, S, C, device=dev).to(torch.bfloat16)
    w = (torch.randn(C, C, 3, 3, 3, device=dev) * 0.02)
    wp = ops.pack_weight(w, torch.bfloat16, False, False)
    y = torch.empty_like(xx)
    it = 20 if size == "B" else 8
    t = time_kernel(lambda: ops.conv3d(ops.CONV_FWD, xx, wp, None, (S, S, S), 3, 1, out=y, want_partials=True), iters=it)
    t_plain = time_kernel(lambda: ops.conv3d(ops.CONV_FWD, xx, wp, None, (S, S, S), 3, 1, out=y), iters=it)
    del xx, y
    flops = 2.0 * B * S ** 3 * C * C * 27
    # the same launch inside the step: wrap ops.conv3d for three steps (events only, no synchronisation inside the step)
    evs, orig = [], ops.conv3d

    def timed_conv3d(mode, x_, w_packed, bias, out_spatial, ksize, stride, *a, **k):
        hit = (mode == ops.CONV_FWD and ksize == 3 and stride == 1 and tuple(x_.shape[1:]) == (S, S, S, C)
               and w_packed.logical == (C, C) and k.get("ep_scale") is None and k.get("want_partials"))
        if not hit:
            return orig(mode, x_, w_packed, bias, out_spatial, ksize, stride, *a, **k)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = orig(mode, x_, w_packed, bias, out_spatial, ksize, stride, *a, **k)
        e1.record()
        evs.append((e0, e1))
        return r
    in_step = None
    if tr is not None:
        ops.conv3d = timed_conv3d
        try:
            for _ in range(3):
                tr.step(x, epoch=500)
            torch.cuda.synchronize()
        finally:
            ops.conv3d = orig
        if evs:
            ds = sorted(a.elapsed_time(b) for a, b in evs)
            in_step = ds[len(ds) // 2]
    achieved = flops / t / 1e12
    algo = (2 * S ** 3 * C * 2 * B) + 27 * C * C * 2
    # HBM bytes per launch: the PMC counters of the SAME launch (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, FETCH_SIZE
    # doubled per the gfx950 correction, counters in KB = 1024 B), as tools/pmc_k3.sh left them in profiles/r06_pmc_k3.json: measured at
    # B = 16 for 64 -> 64 @128^3 and linear in B.  No counter pass exists for the other shapes -> null; a missing file is an error, not a constant.
    traffic = src = None
    if (C, S) == (64, 128):
        pj = os.path.join(ROOT, PMC_K3_JSON)
        rec = json.load(open(pj))["conv_k3_kernel"]            # (raises if the profile was not committed)
        traffic = (rec["fetch_bytes"] + rec["write_bytes"]) * B / rec["batch"]
        src = (f"{PMC_K3_JSON} (tools/pmc_k3.sh: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE of this launch at B={rec['batch']}: "
               f"{(rec['fetch_bytes'] + rec['write_bytes']) / rec['algorithmic_bytes']:.2f} x algorithmic, {rec['launch_us']:.0f} us under the profiler; scaled to the bench batch)")
    return {"bound": "mfma", "kernel": f"conv_k3_kernel<4,false,true> (decoder conv3 {C}->{C} @{S}^3, statistics epilogue on)", "achieved": round(achieved, 2),
            "peak": MFMA_BF16_PEAK / 1e12, "unit": "TFLOP/s", "frac": round(achieved * 1e12 / MFMA_BF16_PEAK, 4),
            "traffic": traffic, "traffic_source": src,
            "algorithmic_bytes": algo, "launch_ms": round(t * 1e3, 4), "flop_per_launch": flops,
            "launch_ms_no_statistics": round(t_plain * 1e3, 4), "frac_no_statistics": round(flops / t_plain / MFMA_BF16_PEAK, 4),
            "launch_ms_in_step": None if in_step is None else round(in_step, 4),
            "frac_in_step": None if in_step is None else round(flops / (in_step * 1e-3) / MFMA_BF16_PEAK, 4)}


def encoder_forward_hbm(model, x, dev, size, patch, mask_ratio):
    """North-star figure: achieved algorithmic HBM GB/s of the STUDENT SPARSE-ENCODER FORWARD (bf16; algorithmic bytes per
    SURVEY.md 8d / BASELINE.md 3) from HIP events around that part of the forward."""
    from anatomask_amd import engine, ops
    p0, m0, bytes_pv, flop_pv = ENC_FWD_ALGO[size]
    if (p0, m0) != (patch, mask_ratio):
        return None
    B = x.shape[0]
    L = model.spec.fmap[0] * model.spec.fmap[1] * model.spec.fmap[2]
    k = torch.rand(B, L, device=dev)
    mk = ops.mask_sampler(torch.zeros(B, L, device=dev), k, model.len_keep, 0)
    mi = ops.MaskInfo(mk.view(B, *model.spec.fmap), n_active=B * model.len_keep)
    xs = x[:, 0].contiguous()

    def enc_only():
        engine.forward(model.spec, model._W, model._pack, xs, mi, True, None, encoder_only=True)
    t = time_kernel(enc_only, iters=10, warm=2)
    algo = bytes_pv * B
    gbs = algo / t / 1e9
    out = {"what": f"student sparse-encoder forward, STUNet-{size} {patch}^3 bf16, mask {mask_ratio}", "ms": round(t * 1e3, 3),
           "algorithmic_bytes": algo, "achieved_GBps": round(gbs, 1), "frac_of_hbm_peak": round(gbs * 1e9 / HBM_PEAK, 4),
           "flop": flop_pv * B, "achieved_TFLOPs": round(flop_pv * B / t / 1e12, 2)}
    if size == "B":
        # the same forward in COUNTED bytes: rocprofv3 FETCH_SIZE + WRITE_SIZE over its launches (tools/enc_traffic.py on this round's tree,
        # committed under profiles/; linear in the batch).  A missing file is an error, not a constant.
        rec = json.load(open(os.path.join(ROOT, ENC_TRAFFIC_JSON)))
        counted = rec["counted_bytes"] * B / rec["batch"]
        out.update(counted_bytes=counted, counted_GBps=round(counted / t / 1e9, 1), counted_frac=round(counted / t / HBM_PEAK, 4),
                   counted_source=f"{ENC_TRAFFIC_JSON} ({rec['launches']} launches at B={rec['batch']}: {rec['counted_bytes'] / rec['algorithmic_bytes']:.2f} x the algorithmic bytes)")
    return out


def cpu_baseline(state_dict_cpu, spec_kw):
    """the CPU oracle (port of the reference step, pinned against it by tests/golden) on the host cores, in this run.
      value            C2: ONE full AnatoMask step of the bench workload at B=1 (teacher fwd + sampler + student fwd/bwd + clip + AdamW +
                       EMA), n = min(32, host cores) threads  (measured on the MI355X host: 8/16/32/64 threads -> 5.4/4.8/4.1/5.8 s per
                       forward; 256 oversubscribes 10x)
      c2_8_threads     the same step on 8 threads (comparability with the survey container's 0.029-0.037 volumes/s)
      c1 / c1_8_threads  STUNet-small, 48^3, B=2, plain SparK step (P/pretrain.py): 2 warm-up + 10 timed steps, median"""
    from oracle import anatomask_oracle as O
    n = min(32, os.cpu_count() or 1)

    def c2(threads):
        torch.set_num_threads(threads)
        cfg = O.Config(spec_kw["dims"], spec_kw["depth"], spec_kw["width"], (128, 128, 128), 0.6)
        st = O.StepState(cfg, state_dict_cpu)
        x = torch.randn(1, 1, 128, 128, 128, generator=torch.Generator().manual_seed(1234))
        mask1 = O.random_mask(cfg, 1, torch.Generator().manual_seed(4321))
        keys = torch.rand(1, cfg.L, generator=torch.Generator().manual_seed(4322))
        t0 = time.time()
        o = O.train_step(st, x, mask1, keys, 500, 999, 1e-4, 0.9995)
        return time.time() - t0, o["loss"]

    def c1(threads):
        torch.set_num_threads(threads)
        cfg = O.Config.stunet_s((48, 48, 48), 0.6)
        st = O.StepState(cfg, O.seeded_state(cfg, 1))
        x = torch.randn(2, 1, 48, 48, 48, generator=torch.Generator().manual_seed(1234))
        ts = []
        for i in range(12):
            m = O.random_mask(cfg, 2, torch.Generator().manual_seed(4321 + i))
            t0 = time.time()
            o = O.plain_spark_step(st, x, m, 1e-4)
            ts.append(time.time() - t0)
        ts = sorted(ts[2:])
        return ts[len(ts) // 2], o["loss"]
    dt, loss = c2(n)
    res = {"value": round(1.0 / dt, 5), "unit": "volumes/s", "cores": n, "kind": "port",
           "sample": f"1 full step, B=1, STUNet-B 128^3 fp32, torch-CPU oracle, {dt:.1f} s, loss {loss:.4f}"}
    d1, l1 = c1(n)
    res["c1"] = {"value": round(2.0 / d1, 3), "unit": "volumes/s", "cores": n,
                 "sample": f"STUNet-S 48^3 B=2 plain SparK step, median of 10 steps after 2 warm-up, {d1 * 1e3:.0f} ms/step, loss after 12 steps {l1:.4f}"}
    if n != 8:
        d8, l8 = c1(8)
        res["c1_8_threads"] = {"value": round(2.0 / d8, 3), "unit": "volumes/s", "cores": 8, "sample": f"{d8 * 1e3:.0f} ms/step"}
        dt8, _ = c2(8)
        res["c2_8_threads"] = {"value": round(1.0 / dt8, 5), "unit": "volumes/s", "cores": 8, "sample": f"1 full step, {dt8:.1f} s"}
    return res


def secondary_lines(a, kw, dev, dtype_main):
    """Beside the headline (never instead of it): the same step at the reference's batch 4, and the reference RECIPE
    (P/pretrain_AntoMask.py:209,229: input 112 x 112 x 128, batch 4, fp32 -- here fp32 storage / exact-f32 MFMA)."""
    from anatomask_amd import modules as M
    from anatomask_amd.trainer import AnatoMaskTrainer
    out = []
    torch.cuda.synchronize()
    from anatomask_amd import ops as _ops
    for what, size3, batch, dt, warm, steps, split, size, mr, rc in [
            ("same workload at the reference's batch size 4", (a.patch,) * 3, 4, dtype_main, 5, 11, False, None, a.mask_ratio, False),
            ("reference recipe: 112x112x128, batch 4, fp32 storage (AMP=False), split-bf16 products (AM_DT_F32S: hi hi + hi lo + lo hi (+ lo lo), fp32 accumulation)",
             (112, 112, 128), 4, torch.float32, 3, 7, True, None, a.mask_ratio, False),
            ("reference recipe: 112x112x128, batch 4, fp32 storage (AMP=False), exact-f32 MFMA (the parity mode)", (112, 112, 128), 4, torch.float32, 2, 5, False,
             None, a.mask_ratio, False),
            # BASELINE.json configs[3] / configs[4] (single-GPU rows): the larger backbones, full AnatoMask step, bf16
            ("BASELINE configs[3]: STUNet-L AnatoMask step, 160^3, mask_ratio 0.7, batch 4, bf16", (160,) * 3, 4, torch.bfloat16, 2, 5, False, "L", 0.7, False),
            ("BASELINE configs[4]: STUNet-H AnatoMask step, 192^3, mask_ratio 0.6, batch 2, bf16, activation recomputation (P/GC.py)", (192,) * 3, 2, torch.bfloat16,
             2, 3, False, "H", 0.6, True)]:
        torch.manual_seed(0)
        kw_ = kw if size is None else M.STUNET_CONFIGS[size]
        model = M.build_spark(kw_["dims"], kw_["depth"], kw_["width"], size3, mr, compute_dtype=dt, recompute=rc).to(dev)
        tr = AnatoMaskTrainer(model, lr=1e-4, total_epochs=1000, seed=4321, distributed=False, f32_split=split)
        tr.set_epoch(500)
        x = torch.randn(batch, 1, *size3, device=dev, generator=torch.Generator(device=dev).manual_seed(1234))
        for _ in range(warm):
            tr.step(x, epoch=500)
        torch.cuda.synchronize()
        # per-step HIP events, MEDIAN reported: right after the headline run's 60 GiB went back to the driver (empty_cache) a window's
        # first steps can stall on allocator traffic for hundreds of ms -- a property of the process state, not of the step
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
        evs[0].record()
        for i in range(steps):
            o = tr.step(x, epoch=500)
            evs[i + 1].record()
        torch.cuda.synchronize()
        per = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(steps))
        d = per[len(per) // 2] * 1e-3
        out.append({"what": what, "per_gpu_batch": batch, "dtype": "bf16" if dt == torch.bfloat16 else "f32", "ms_per_step": round(d * 1e3, 3),
                    "ms_per_step_max": round(per[-1], 3), "value": round(batch / d, 3), "unit": "volumes/s", "steps": steps,
                    "final_loss": round(o["loss"].item(), 5)})
        del tr, model, x
        torch.cuda.empty_cache()
    return out


def exchange_report(tr, grad_elems, world, dt, dtn, steps, backend):
    """`exchange` of the JSON line (N > 1): what was sent per step and what it cost (step time with the exchange off -> exposed ms)."""
    sizes = [(b - a) * 4 for a, b in tr.exchange_log]
    tl = tr.exchange_timeline()                              # the LAST step's collectives: [issue_ms, done_ms] per piece from the start of backward
    return {"timeline": tl, "exposed_ms_last_step": tl["exposed_ms"], "backend": backend, "ranks": world, "gradient_bytes_per_step": int(grad_elems) * 4, "collectives_per_step": len(sizes),
            "largest_collective_bytes": max(sizes) if sizes else 0, "first_collective_after_tag": getattr(tr, "first_sent_tag", None),
            "ms_per_step_without_exchange": round(dtn / steps * 1e3, 3), "exposed_ms": round((dt - dtn) / steps * 1e3, 3)}


def ranks_report(world, rank, dist, dev, per_ms, dt_local):
    """N > 1: what makes the line self-proving -- every rank's device identity (uuid / PCI bus id), its own wall time of the timed window and
    the median of its per-step HIP-event times, gathered on rank 0.  `ranks_seen` must equal N and the devices must be N DIFFERENT ones."""
    if dev is not None:
        pr = torch.cuda.get_device_properties(dev)
        ident = str(getattr(pr, "uuid", "")) or f"{pr.name}:{getattr(pr, 'pci_bus_id', dev.index)}"
        ident = f"{ident}|pci{getattr(pr, 'pci_bus_id', '?')}|local{dev.index}"
    else:
        ident = f"host-pid{os.getpid()}"
    mine = {"rank": rank, "device": ident, "window_s": round(dt_local, 4), "step_ms_median": round(sorted(per_ms)[len(per_ms) // 2], 3),
            "step_ms_min": round(min(per_ms), 3), "step_ms_max": round(max(per_ms), 3)}
    if dist is None:
        return [mine]
    got = [None] * world
    dist.all_gather_object(got, mine)
    return got


def timed_window(step, steps, world, dist, dev):
    """EXACTLY `steps` steps bracketed by barrier + synchronize on both sides (wall clock, MAX over ranks), with a HIP event
    after every step for the per-step distribution.  dev None: host-only ranks (the gloo dry run), no device calls.
    dist None: no process group (the plain N = 1 run); a process group of ONE rank (AM_BENCH_FORCE_DIST) runs every collective."""
    gpu = dev is not None
    if dist is not None:
        dist.barrier()
    if gpu:
        torch.cuda.synchronize()
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)] if gpu else []
    t0 = time.perf_counter()
    if gpu:
        evs[0].record()
    out = None
    for i in range(steps):
        out = step()
        if gpu:
            evs[i + 1].record()
    if dist is not None:
        dist.barrier()
    if gpu:
        torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    per = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(steps)) if gpu else [dt / steps * 1e3] * steps
    timed_window.local_s = dt                               # this rank's own clock (ranks_report)
    if dist is not None:
        tt = torch.tensor([dt], device=dev if gpu else "cpu", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = tt.item()
    return dt, per, out


def dry_run_launch(a):
    """`--dry-run-launch` (CPU, gloo; tests/test_launch.py): the N-rank self-launch AND the rank body's N > 1 plumbing without a GPU.
    Every rank builds the real model (STUNet-S) and the real trainer's gradient-exchange state over host flat buffers; a step is
    engine.backward's hook sequence over known per-rank gradients (the HIP compute is what is left out); the timed window, the
    MAX-over-ranks clock, the exchange-off window and the `exchange` record are the real ones.  Rank 0 prints one JSON line."""
    import torch.distributed as dist
    from anatomask_amd import modules as M
    from anatomask_amd.engine import Spec
    from anatomask_amd.trainer import AnatoMaskTrainer
    world, rank = int(os.environ["WORLD_SIZE"]), int(os.environ["RANK"])
    dist.init_process_group("gloo")
    kw = M.STUNET_CONFIGS["S"]
    torch.manual_seed(0)
    m = M.build_spark(kw["dims"], kw["depth"], kw["width"], (48, 48, 48), 0.6)
    named = list(m.named_parameters())
    order = [(n, p) for n, p in named if n not in m._dead] + [(n, p) for n, p in named if n in m._dead]
    offs, tot = {}, 0
    for n, p in order:
        if n in m._dead and "live_end" not in offs:
            offs["live_end"] = tot
        offs[n] = tot
        tot += (p.numel() + 3) // 4 * 4
    m._offs, m._live_end, m._pnames, m._gflat = offs, offs["live_end"], [n for n, _ in named], torch.zeros(tot)
    m.spec = Spec(kw["dims"], kw["depth"], kw["width"], (48, 48, 48))
    tr = AnatoMaskTrainer.__new__(AnatoMaskTrainer)
    tr.model, tr.distributed, tr.pg, tr.world, tr._works, tr.exchange_log = m, True, None, world, [], []
    tr._pieces = tr._t_begin = tr._t_end = None
    tr.BUCKET_BYTES, tr.FLUSH_BYTES = 8 << 20, 4 << 20
    tr._build_ranges()
    tags = ["proj"] + [f"dec{i}" for i in reversed(range(4))] + ["densify"] + [f"stage{s}.0" for s in reversed(range(5))]

    def step():
        m._gflat[:m._live_end] = float(rank + 1)
        tr._exchange_begin()
        for t in tags:
            tr._after_group(t)
        tr._finish_exchange()
        return m._gflat[:1] * tr.grad_scale
    for _ in range(a.warmup):
        step()
    dt, per0, out = timed_window(step, a.steps, world, dist, None)
    rr = ranks_report(world, rank, dist, None, per0, timed_window.local_s)
    ok = abs(float(out) - sum(range(1, world + 1)) / world) < 1e-6
    tr.distributed = False
    dtn, _, _ = timed_window(step, a.steps, world, dist, None)
    tr.distributed = True
    step()
    t = torch.tensor([float(rank)])
    dist.all_reduce(t)
    if rank == 0:
        print(json.dumps({"dry_run": True, "n_gpus": a.gpus, "world": world, "rank_sum": t.item(), "backend": "gloo", "mean_gradient_ok": ok,
                          "ranks_seen": len({r["device"] for r in rr}), "ranks": rr,
                          "exchange": exchange_report(tr, m._live_end, world, dt, dtn, a.steps, "gloo")}), flush=True)
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=int(os.environ.get("AM_BENCH_BATCH", "16")),
                    help="volumes per GPU per step (SURVEY.md 8d C2: chosen to fill the GPU and reported; measured 4 / 8 / 16 / 24: 92 / 102 / 106 / 107 volumes/s, 16 = 62 GiB reserved; the reference default is 4)")
    ap.add_argument("--size", default="B")
    ap.add_argument("--patch", type=int, default=128)
    ap.add_argument("--mask-ratio", type=float, default=0.6)
    ap.add_argument("--recompute", action="store_true")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-h2d", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary lines (batch 4; the reference recipe in fp32 storage)")
    ap.add_argument("--dry-run-launch", action="store_true", help="CPU/gloo check of the N-rank self-launch (no GPU work)")
    a = ap.parse_args()

    from anatomask_amd import launch
    if a.gpus > 1 and not launch.launched():
        # no launcher around us: become one.  Nothing in this process has touched the GPU (importing torch does not).
        sys.exit(launch.self_launch(a.gpus, os.path.abspath(__file__), sys.argv[1:]))
    if a.dry_run_launch:
        return dry_run_launch(a)

    import torch.distributed as dist
    from anatomask_amd import modules as M
    from anatomask_amd.data import DeviceFeed
    from anatomask_amd.trainer import AnatoMaskTrainer

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    # AM_BENCH_FORCE_DIST=1 (set by a launcher BEFORE this process touches the GPU: tests/test_e2e_gpu.py::test_bench_rank_body_rccl_world1): the
    # N > 1 rank body -- RCCL process group, barriers, MAX-over-ranks clock, the gradient exchange, the exchange-off window, exchange_report,
    # ranks_report's all_gather_object -- with a world of ONE rank, so that every line of it has run on a GPU before an 8-GPU node runs it
    dist_on = world > 1 or (os.environ.get("AM_BENCH_FORCE_DIST") == "1" and launch.launched())
    if dist_on:
        # a collective that hangs (a rank that died, a link that never came up) must END the job: RCCL's watchdog aborts the process once a
        # collective is older than `timeout` (TORCH_NCCL_ASYNC_ERROR_HANDLING=1, torch's default, tears the process down), the launcher
        # then kills the other ranks and exits non-zero -- instead of ten silent minutes at a barrier
        import datetime
        os.environ.setdefault("TORCH_NCCL_ASYNC_ERROR_HANDLING", "1")
        dist.init_process_group("nccl", device_id=dev, timeout=datetime.timedelta(seconds=int(os.environ.get("AM_BENCH_COLLECTIVE_TIMEOUT_S", "240"))))
    else:
        dist = None

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
    # first use of timing-enabled HIP events costs ~2 s of one-off runtime set-up on this stack: pay it before the timed window
    w0, w1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    w0.record(); w1.record(); w1.synchronize(); w0.elapsed_time(w1)
    dt, per, out = timed_window(lambda: tr.step(x, epoch=500), a.steps, world, dist, dev)
    rr = ranks_report(world, rank, dist, dev, per, timed_window.local_s)
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
               "final_loss": round(loss, 5), "step_ms_median": round(per[len(per) // 2], 3), "step_ms_min": round(per[0], 3),
               # N > 1: the devices that took part (must be N distinct ones), every rank's own clock, and the per-GPU rate to hold against the N = 1 line
               "ranks_seen": len({r["device"] for r in rr}), "ranks": rr,
               "value_per_gpu": round(a.batch * a.steps / dt, 4),
               "rank_step_ms": {"min": min(r["step_ms_median"] for r in rr), "median": sorted(r["step_ms_median"] for r in rr)[len(rr) // 2],
                                "max": max(r["step_ms_median"] for r in rr)}}

    # ---- the same steps fed from pinned host memory through the copy stream (reported beside `value`, never as `value`)
    if not a.no_h2d:
        host = x.cpu().pin_memory()
        feed = DeviceFeed(({"data": host} for _ in range(a.steps + 4)), dev)
        for _ in range(2):
            tr.step(next(feed), epoch=500)
        dth, _, _ = timed_window(lambda: tr.step(next(feed), epoch=500), a.steps, world, dist, dev)
        if rank == 0:
            res["value_with_h2d"] = round(a.batch * world * a.steps / dth, 4)
            res["h2d"] = {"bytes_per_step": host.numel() * 4, "how": "pinned host batch -> double-buffered device slots on a copy stream, overlapped with the previous step"}
        del feed, host

    # ---- exposed communication: the same window with the gradient exchange switched off (N > 1)
    if dist_on:
        tr.distributed = False
        for _ in range(2):
            tr.step(x, epoch=500)
        dtn, _, _ = timed_window(lambda: tr.step(x, epoch=500), a.steps, world, dist, dev)
        tr.distributed = True
        tr.step(x, epoch=500)                                        # (refills exchange_log)
        if rank == 0:
            res["exchange"] = exchange_report(tr, model._live_end, world, dt, dtn, a.steps, "rccl")

    # every collective of the job is behind us: the ranks part HERE (barrier, then the process group goes), and rank 0 measures the
    # roofline / encoder figures alone -- no rank waits at a barrier while another one benchmarks kernels
    if dist_on:
        dist.barrier()
        tr.distributed = False
        dist.destroy_process_group()
    if rank == 0 and not a.no_roofline and a.dtype == "bf16":
        res["roofline"] = dominant_kernel_roofline(a.batch, dev, tr if world == 1 else None, x, C=kw["width"] // 8, S=a.patch, size=a.size)
        enc = encoder_forward_hbm(model, x, dev, a.size, a.patch, a.mask_ratio)
        if enc is not None:
            res["encoder_fwd_hbm"] = enc
    if rank == 0:
        del tr, model, x
        torch.cuda.empty_cache()
        if world == 1 and not a.no_secondary and a.size == "B" and a.patch == 128:
            res["secondary"] = secondary_lines(a, kw, dev, dtype)
        if sd_cpu is not None:
            try:
                res["cpu_baseline"] = cpu_baseline(sd_cpu, kw)
            except Exception as e:                                   # e.g. host OOM: report, do not fail the bench
                res["cpu_baseline"] = {"value": None, "unit": "volumes/s", "cores": os.cpu_count(), "kind": "port", "sample": f"failed: {e!r}"}
        print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
