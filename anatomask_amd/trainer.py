"""The fused AnatoMask training step (the hot loop body, P/pretrain_AntoMask.py:418-441) on the HIP engine.

teacher fwd (eval, no grad) -> per-patch raw l2 -> hard-mask sampler -> student fwd -> normalised masked MSE
-> backward -> [RCCL gradient all-reduce, overlapped] -> clip(12) + AdamW + EMA in one pass.
Everything stays on the device: no .item(), no numpy round trip, no host synchronisation inside a step
(the reference has ~3 D2H syncs per step plus one per sample in generate_mask).
"""
from __future__ import annotations

from typing import Dict, Optional

import torch
import torch.distributed as dist

from . import engine, ops
from .modules import ModelEma, SparK, ema_decay_for_epoch


class _Range:
    """roctx range around a phase of the step (torch.cuda.nvtx IS roctx on ROCm): `rocprofv3 --marker-trace` shows teacher / sampler /
    student / backward / exchange / optimizer as named spans.  Off unless AnatoMaskTrainer.trace_ranges is set (tools/): a push / pop pair
    per phase is host work the launch-bound configurations should not pay by default."""
    __slots__ = ("on",)

    def __init__(self, on: bool, name: str):
        self.on = on
        if on:
            torch.cuda.nvtx.range_push(name)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        if self.on:
            torch.cuda.nvtx.range_pop()
        return False


class AnatoMaskTrainer:
    trace_ranges = False          # roctx ranges around the phases of step() (see _Range)

    def __init__(self, model: SparK, lr: float = 1e-4, weight_decay: float = 1e-5, betas=(0.9, 0.999), eps: float = 1e-8,
                 clip: float = 12.0, ema_decay: float = 0.999, total_epochs: int = 1000, guide: bool = True, seed: int = 4321,
                 process_group=None, distributed: Optional[bool] = None, self_distill: bool = True, deterministic_wgrad: bool = False,
                 f32_split: bool = False):
        self.model = model
        if model.spec.dec_inorm:
            raise NotImplementedError("the fused trainer keeps the drivers' BatchNorm decoder (P/pretrain_AntoMask.py:212); a "
                                      "LightDecoder(use_IN=True) model trains through the module API (SparK.forward / forward_loss / a torch optimizer)")
        # fp32-storage models only: matrix-core products from bf16 hi / lo splits (AM_DT_F32S) instead of the exact fp32 matrix
        # instruction -- the fast reference-precision mode (the reference recipe is AMP = False, P/pretrain_AntoMask.py:239).  A property of
        # the MODEL (SparK.f32_split, inherited by the teacher's deep copy): no process-wide switch
        self.f32_split = bool(f32_split) or bool(getattr(model, "f32_split", False))
        model.set_f32_split(self.f32_split)
        self._capturing, self._graph, self._graph_key = False, None, None
        # convolution weight gradients as per-slot partial sums folded in a fixed order instead of fp32 atomics (am_conv3d_wgrad's
        # det_workspace; +3 % step time).  A property of the MODEL (SparK.deterministic_wgrad -> its PackCache), like f32_split: nothing
        # process-wide is touched.  The per-channel norm statistics keep their fp64 atomics (order-dependent at the 1e-16 level).
        self.deterministic_wgrad = bool(deterministic_wgrad) or bool(getattr(model, "deterministic_wgrad", False))
        model.set_deterministic_wgrad(self.deterministic_wgrad)
        self.self_distill = self_distill      # False: plain SparK step (P/spark3D.py:98-146, P/pretrain.py): random mask, no teacher
        model._ensure_flat()
        model.train()
        self.teacher = ModelEma(model, decay=ema_decay)
        self.teacher.ema._ensure_flat()
        self.lr, self.wd, self.betas, self.eps, self.clip = lr, weight_decay, betas, eps, clip
        self.total_epochs, self.guide = total_epochs, guide
        dev = model._flat.device
        n = model._live_end
        self.m = torch.zeros(n, device=dev)
        self.v = torch.zeros(n, device=dev)
        self.sumsq = torch.zeros(1, device=dev, dtype=torch.float64)
        self.gnorm = torch.zeros(1, device=dev)
        self.step_count = 0
        # per-step non-finite stop on the device (am_adamw_ema `guard`): {latched, first bad call, calls, 0}; the BatchNorm buffers the
        # student's forward updates are snapshotted per step and put back by am_guard_restore when the guard latches
        self.guard = torch.zeros(4, device=dev, dtype=torch.int32)
        self._guard_step0 = 0
        self._snap_b = torch.empty_like(model._bflat)
        self._snap_i = torch.empty_like(model._iflat)
        self.gen = torch.Generator(device=dev)
        self.gen.manual_seed(seed)
        self.distributed = dist.is_available() and dist.is_initialized() if distributed is None else distributed
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if self.distributed else 1
        self._works = []
        self.exchange_log = []                     # (a, b) of every collective of the current step, in issue order
        if self.distributed:                      # DDP start-up broadcast of parameters AND buffers (P/pretrain_AnatoMask_DDP.py:239-240)
            dist.broadcast(model._flat, 0, group=process_group)
            dist.broadcast(model._bflat, 0, group=process_group)
            self.teacher.ema._flat.copy_(model._flat)
            self.teacher.ema._bflat.copy_(model._bflat)
            model.weights_changed(); self.teacher.ema.weights_changed()   # packed MFMA copies made before the broadcast are stale
        self._build_ranges()

    # ------------------------------------------------------------------ gradient exchange (a18)
    BUCKET_BYTES = 64 << 20       # no collective larger than this (STUNet-H's first decoder block alone is 0.9 GB of fp32 gradients)
    FLUSH_BYTES = 24 << 20        # ready gradients are held back until this much has accumulated (DDP's 25 MB bucket_cap_mb)

    def _build_ranges(self):
        """Flat-buffer ranges of the gradient groups in the order backward finishes them: projection, decoder block 3..0, densify,
        encoder blocks deep -> shallow (engine.backward's `after_group` tags).  The flat buffer is in named_parameters order
        [encoder | decoder | densify], so the tags tile the live region; pieces that become ready one after the other are usually
        adjacent in memory and are merged before they are sent."""
        model = self.model
        o, n = model._offs, model._live_end
        live = [k for k in model._pnames if k not in model._dead]
        ends = {k: (o[live[i + 1]] if i + 1 < len(live) else n) for i, k in enumerate(live)}

        def span(pred):
            ks = [k for k in live if pred(k)]
            if not ks:
                return None
            a, b = min(o[k] for k in ks), max(ends[k] for k in ks)
            assert sum(ends[k] - o[k] for k in ks) == b - a, "a gradient group must be one contiguous range of the flat buffer"
            return (a, b)
        r: Dict[str, tuple] = {}
        r["proj"] = span(lambda k: k.startswith("dense_decoder.proj."))
        for i in range(len(model.spec.dec_chs) - 1):
            r[f"dec{i}"] = span(lambda k, i=i: k.startswith(f"{engine.DEC}.{i}."))
        r["densify"] = span(lambda k: k.startswith(("densify_norms.", "densify_projs.", "mask_tokens.")))
        for s in range(model.spec.n_stage):
            for b in range(model.spec.depth[s]):
                r[f"stage{s}.{b}"] = span(lambda k, s=s, b=b: k.startswith(f"{engine.ENC}.{s}.{b}."))
        self._ranges = {t: v for t, v in r.items() if v is not None}
        self._pending = []                         # merged (a, b) ranges whose gradients are final but not yet sent

    def _after_group(self, tag: str):
        """Gradient group `tag` is final.  Groups are merged with adjacent ready ones and sent once FLUSH_BYTES have accumulated, so the
        exchange starts with the first decoder blocks and overlaps the rest of backward."""
        if not self.distributed or tag not in self._ranges:
            return
        a, b = self._ranges[tag]
        merged = []
        for (c, d) in self._pending:
            if d == a:
                a = c
            elif c == b:
                b = d
            else:
                merged.append((c, d))
        merged.append((a, b))
        self._pending, self._last_tag = merged, tag
        if sum(d - c for c, d in self._pending) * 4 >= self.FLUSH_BYTES:
            self._flush()

    _COMM: Dict[int, "torch.cuda.Stream"] = {}     # per device: the stream the collectives are issued from (nothing else ever runs there)
    _pieces = None                                  # timeline of the current step's collectives (exchange_timeline)
    _t_begin = None

    @classmethod
    def _comm_stream(cls, dev) -> "torch.cuda.Stream":
        i = dev.index if dev.index is not None else torch.cuda.current_device()
        if i not in cls._COMM:
            cls._COMM[i] = torch.cuda.Stream(device=dev)
        return cls._COMM[i]

    def _exchange_begin(self):
        """start of a step's backward: the origin of the exchange timeline."""
        import time
        self.exchange_log.clear()
        self._pieces, self._t_end = [], None
        g = self.model._gflat
        if g.is_cuda:
            self._t_begin = torch.cuda.Event(enable_timing=True)
            self._t_begin.record()
        else:
            self._t_begin = time.perf_counter()

    def _flush(self):
        """async all-reduce(SUM) of every pending range, in pieces of at most BUCKET_BYTES.  On the GPU the collectives are issued from a
        DEDICATED third stream that waits for (an event on the main stream, an event on the weight-gradient side stream) -- what has
        been enqueued on both up to this point is exactly what the pending ranges need -- so RCCL's kernels never queue behind the
        persistent weight-gradient grids that are launched on the side stream AFTER this point, and the main stream goes on with backward
        un-joined.  Every piece carries an event pair (eligible to start, finished) on that stream: exchange_timeline()."""
        if not self._pending:
            return
        import time
        if not self.exchange_log:
            self.first_sent_tag = getattr(self, "_last_tag", None)        # how early in backward the exchange starts (bench.py `exchange`)
        if self._pieces is None:
            self._exchange_begin()
        g = self.model._gflat
        cuda = g.is_cuda
        stream_wait = cuda and dist.get_backend(self.pg) == "nccl"       # RCCL: Work.wait() is a stream-level wait (the host does not block)
        ctx = None
        if cuda:
            comm = self._comm_stream(g.device)
            ev = torch.cuda.Event()
            ev.record()
            comm.wait_event(ev)
            if engine._USE_SIDE:
                ev2 = torch.cuda.Event()
                ev2.record(engine._side_stream(g.device))
                comm.wait_event(ev2)
            ctx = torch.cuda.stream(comm)
            ctx.__enter__()
        try:
            for a, b in sorted(self._pending, reverse=True):
                npiece = max(1, -(-(b - a) * 4 // self.BUCKET_BYTES))
                step = ((b - a + npiece - 1) // npiece + 3) // 4 * 4
                for c in range(a, b, step):
                    d = min(c + step, b)
                    rec = {"range": (c, d), "bytes": (d - c) * 4}
                    if cuda:
                        rec["e0"] = torch.cuda.Event(enable_timing=True)
                        rec["e0"].record()
                    else:
                        rec["t0"] = time.perf_counter()
                    w = dist.all_reduce(g[c:d], op=dist.ReduceOp.SUM, group=self.pg, async_op=True)
                    if stream_wait:
                        w.wait()                                         # the comm stream waits for RCCL's stream; the main stream does not
                        rec["e1"] = torch.cuda.Event(enable_timing=True)
                        rec["e1"].record()
                    else:
                        rec["work"] = w
                        self._works.append(w)
                    self._pieces.append(rec)
                    self.exchange_log.append((c, d))
        finally:
            if ctx is not None:
                ctx.__exit__(None, None, None)
        self._pending = []

    def _finish_exchange(self):
        if not self.distributed:
            return
        import time
        self._flush()
        g = self.model._gflat
        if g.is_cuda:
            self._t_end = torch.cuda.Event(enable_timing=True)            # the main stream has enqueued all of backward: what it waits for
            self._t_end.record()                                          # from here on is exposed communication
        else:
            self._t_end = time.perf_counter()
        for rec in self._pieces or []:                                    # gloo (tests): Work.wait() blocks the host
            if "work" in rec:
                rec.pop("work").wait()
                rec["t1"] = time.perf_counter()
        self._works.clear()
        if g.is_cuda:                                                     # RCCL: the main stream joins the comm stream once
            ev = torch.cuda.Event()
            ev.record(self._comm_stream(g.device))
            torch.cuda.current_stream().wait_event(ev)
        # the buffer now holds the SUM over ranks; 1/world is folded into am_adamw_ema

    def exchange_timeline(self) -> Dict:
        """The last step's collectives: per piece [issue_ms, done_ms] from the start of backward (issue = the piece's gradients were final
        on both streams and it was handed to RCCL; done = its result was in the buffer), where backward ended on the main stream, and the
        communication left exposed behind it.  Synchronises the device (bench / tests only)."""
        if not self._pieces:
            return {"pieces": [], "backward_end_ms": None, "exposed_ms": 0.0}
        cuda = self.model._gflat.is_cuda
        if cuda:
            torch.cuda.synchronize()
        out = []
        for rec in self._pieces:
            if "e1" in rec:
                t0, t1 = self._t_begin.elapsed_time(rec["e0"]), self._t_begin.elapsed_time(rec["e1"])
            elif cuda:                                                    # gloo over device tensors: issue on the device clock, done on the host's
                t0 = self._t_begin.elapsed_time(rec["e0"]); t1 = None
            else:
                t0, t1 = (rec["t0"] - self._t_begin) * 1e3, (rec["t1"] - self._t_begin) * 1e3
            out.append({"bytes": rec["bytes"], "issue_ms": round(t0, 3), "done_ms": None if t1 is None else round(t1, 3)})
        end = self._t_begin.elapsed_time(self._t_end) if cuda else (self._t_end - self._t_begin) * 1e3
        dones = [p["done_ms"] for p in out if p["done_ms"] is not None]
        return {"pieces": out, "backward_end_ms": round(end, 3), "exposed_ms": round(max(0.0, max(dones) - end), 3) if dones else None}

    @property
    def grad_scale(self) -> float:
        """what the flat gradient buffer must be multiplied by to be DDP's mean gradient (folded into am_adamw_ema)."""
        return 1.0 / self.world

    # ------------------------------------------------------------------ one step
    @torch.no_grad()
    def step(self, inp_bchwd: torch.Tensor, epoch: int = 0, mask1: Optional[torch.Tensor] = None,
             keys: Optional[torch.Tensor] = None, lr: Optional[float] = None, ema_decay: Optional[float] = None):
        """inp_bchwd: (B,1,H,W,D) fp32 on the device.  mask1/keys teacher-force the two random draws.
        Returns device tensors only: {'loss','grad_norm','mask','recon_loss','rec_loss'}."""
        m, t = self.model, self.teacher.ema
        spec = m.spec
        x = inp_bchwd[:, 0].float().contiguous()
        B = x.shape[0]
        L = spec.fmap[0] * spec.fmap[1] * spec.fmap[2]
        dev = x.device
        # 1. first mask (SparK.mask :419)
        if mask1 is None:
            k1 = torch.rand(B, L, device=dev, generator=self.gen)
            m1 = ops.mask_sampler(torch.zeros(B, L, device=dev), k1, m.len_keep, 0)
        else:
            m1 = mask1.reshape(B, L).to(device=dev, dtype=torch.uint8).contiguous()
        mi1 = ops.MaskInfo(m1.view(B, *spec.fmap), n_active=B * m.len_keep if mask1 is None else None)
        tr_ = self.trace_ranges
        if self.self_distill:
            # 2. teacher pass + raw per-patch loss (:421-425)
            # (only the masked patches' teacher loss is used: the last decoder conv skips the visible 40 % of the volume)
            with _Range(tr_, "anatomask.teacher_forward"):
                need = ops.MaskInfo((1 - m1).view(B, *spec.fmap), n_active=B * (L - m.len_keep) if mask1 is None else None) if spec.input_size[0] // spec.fmap[0] == 16 else None
                # (STUNet-B / L / H: the decoder's eval-mode tail and the per-patch l2 are ONE stencil kernel, ops.head_stencil -- rec1 never exists)
                recon = torch.zeros(B, L, device=dev) if need is not None else None
                rec1 = engine.forward(spec, t._W, t._pack, x, mi1, train=False, needed_patches=need, l2_out=recon)
                if rec1 is not None:
                    recon, _, _, _ = ops.patch_loss_fwd(x, rec1, mi1, normalized=False, want_loss=False)
                del rec1
            # 3. hard-mask sampler (:427)
            with _Range(tr_, "anatomask.mask_sampler"):
                ll = m.len_loss_for(L, m.len_keep, epoch, self.total_epochs - 1, self.guide)
                if keys is None:
                    keys = torch.rand(B, L, device=dev, generator=self.gen)
                mk = ops.mask_sampler(recon, keys.to(dev).float().contiguous(), m.len_keep, ll)
                mi = ops.MaskInfo(mk.view(B, *spec.fmap), n_active=B * m.len_keep)      # the sampler leaves exactly len_keep visible per sample
        else:                                                     # plain SparK: the random mask IS the student mask
            recon, mk, mi = None, m1, mi1
        # 4. student forward + loss (:429-430)
        self._snap_b.copy_(m._bflat); self._snap_i.copy_(m._iflat)   # (what a step that turns out non-finite must not have changed)
        tape = engine.Tape()
        with _Range(tr_, "anatomask.student_forward"):
            rec = engine.forward(spec, m._W, m._pack, x, mi, train=True, tape=tape, recompute=m.recompute)
            l2m, pm, pr, info = ops.patch_loss_fwd(x, rec, mi, normalized=True)
            drec = ops.patch_loss_bwd(x, rec, mi, pm, pr, info, None)
        # 5. backward (:435) with overlapped gradient exchange
        with _Range(tr_, "anatomask.backward"):
            m._gflat.zero_()
            if self.distributed and not self._capturing:          # (the timeline's timing events exist only where an exchange does; never under capture)
                self._exchange_begin()
            engine.backward(spec, m._W, m._G, m._pack, x, mi, tape, drec, self._after_group if self.distributed else None, join_before_hook=False)
            del tape
        with _Range(tr_, "anatomask.exchange_wait"):
            self._finish_exchange()
        rng_opt = _Range(tr_, "anatomask.optimizer_ema")
        # 6. clip + AdamW + EMA (:437-440), one pass over the live parameters
        n = m._live_end
        decay = self.teacher.decay if ema_decay is None else ema_decay
        ops.sumsq(m._gflat[:n], self.sumsq)
        dyn = None
        if self._capturing:                                       # graphed_step: step count / lr arrive through device memory, written by a
            dyn = self._dyn_dev                                   # stream-ordered copy in FRONT of every replay (nothing in the graph reads host memory)
        else:
            self.step_count += 1
        ops.adamw_ema(m._flat, m._gflat, self.m, self.v, t._flat if self.self_distill else None, n, self.lr if lr is None else lr,
                      self.betas, self.eps, self.wd, max(self.step_count, 1), self.sumsq, self.clip, decay, self.gnorm,
                      grad_scale=self.grad_scale, dyn=dyn, guard=self.guard)
        ops.guard_restore(m._bflat, self._snap_b, self.guard)
        ops.guard_restore(m._iflat, self._snap_i, self.guard)
        if not self.self_distill:
            m.weights_changed()
            rng_opt.__exit__()
            return {"loss": info[0:1], "grad_norm": self.gnorm, "mask": mk, "recon_loss": None, "rec_loss": l2m}
        if m._flat.numel() > n:                                   # dead densify[4] tensors: EMA only (no optimizer step)
            ops.ema(t._flat[n:], m._flat[n:], decay, self.guard)
        ops.ema(t._bflat, m._bflat, decay, self.guard)            # BN running stats are EMA'd too (timm: every state_dict entry)
        if m._n_ibuf:                                             # ... and the int64 num_batches_tracked, with timm's float32 promotion
            ops.ema_i64(t._iflat[:m._n_ibuf], m._iflat[:m._n_ibuf], decay, self.guard)
        m.weights_changed(); t.weights_changed()
        rng_opt.__exit__()
        return {"loss": info[0:1], "grad_norm": self.gnorm, "mask": mk, "recon_loss": recon, "rec_loss": l2m}

    # ------------------------------------------------------------------ hipGraph replay of the step (launch-bound configurations)
    def graphed_step(self, inp_bchwd: torch.Tensor, epoch: int = 0):
        """step() captured once into a hipGraph (torch.cuda.CUDAGraph over the HIP stream capture: every am_* launch, the side-stream
        weight gradients and torch's few glue ops become graph nodes) and replayed: one host call per step instead of ~600 launches.
        For small volumes the step is launch-bound (STUNet-S 48^3: 5.9 ms eager, all of it host time).  The shapes, the epoch-dependent
        host scalars (hard-mask quota, EMA decay) and the learning rate are baked into the graph: a change re-captures.  AdamW's step
        count and lr reach the optimizer kernel through device memory (am_adamw_ema dyn_scalars), the mask draws through the
        graph-registered generator.  Single-process only (no gradient exchange inside a graph).  The first two calls run eagerly
        (one-time kernel attribute calls and workspace allocations must not happen under capture)."""
        if self.distributed and self.world > 1:
            raise RuntimeError("graphed_step: the gradient exchange is not captured; use step() under DDP")
        m = self.model
        L = m.spec.fmap[0] * m.spec.fmap[1] * m.spec.fmap[2]
        key = (tuple(inp_bchwd.shape), m.len_loss_for(L, m.len_keep, epoch, self.total_epochs - 1, self.guide), self.teacher.decay, self.lr)
        self._eager_calls = getattr(self, "_eager_calls", 0)
        if self._eager_calls < 2:
            self._eager_calls += 1
            return self.step(inp_bchwd, epoch=epoch)
        if self._graph_key != key:
            dev = inp_bchwd.device
            self._g_inp = inp_bchwd.clone()
            self._dyn_dev = torch.zeros(4, device=dev, dtype=torch.float32)
            self._dyn_ring = [torch.zeros(4, dtype=torch.float32).pin_memory() for _ in range(16)]
            self._dyn_evs = [None] * 16
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            g.register_generator_state(self.gen)
            self._capturing = True
            try:
                with torch.cuda.graph(g):
                    self._g_out = self.step(self._g_inp, epoch=epoch)
            finally:
                self._capturing = False
            self._graph, self._graph_key = g, key
        self._g_inp.copy_(inp_bchwd)
        self.step_count += 1
        # per-step scalars: an async copy out of a RING of pinned slots, enqueued on the replay's stream in front of the replay.  (A
        # captured memcpy node out of ONE reused pinned buffer is read when the GPU gets there: a host that runs ahead -- one sync
        # per epoch -- would have overwritten it with a later step's bias corrections.)  A slot is rewritten only after the copy
        # that read it has completed (event), 16 steps later.
        i = self.step_count % len(self._dyn_ring)
        if self._dyn_evs[i] is not None:
            self._dyn_evs[i].synchronize()
        self._dyn_ring[i].copy_(torch.tensor(ops.adam_dyn_scalars(self.lr, self.betas, self.step_count, self.teacher.decay), dtype=torch.float32))
        self._dyn_dev.copy_(self._dyn_ring[i], non_blocking=True)
        self._dyn_evs[i] = torch.cuda.Event()
        self._dyn_evs[i].record()
        self._graph.replay()
        return self._g_out

    @torch.no_grad()
    def eval_loss(self, inp_bchwd: torch.Tensor, mask: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Validation pass of the plain-SparK driver (P/pretrain.py:426-441: model.eval(), no grad, random mask, the normalised
        masked MSE of P/spark3D.py:130-146): student in eval mode -- decoder BatchNorm on running statistics.  Returns loss[1]."""
        m, spec = self.model, self.model.spec
        x = inp_bchwd[:, 0].float().contiguous()
        B, L = x.shape[0], spec.fmap[0] * spec.fmap[1] * spec.fmap[2]
        if mask is None:
            mk = ops.mask_sampler(torch.zeros(B, L, device=x.device), torch.rand(B, L, device=x.device, generator=self.gen), m.len_keep, 0)
            mi = ops.MaskInfo(mk.view(B, *spec.fmap), n_active=B * m.len_keep)
        else:
            mi = ops.MaskInfo(mask.reshape(B, *spec.fmap).to(device=x.device, dtype=torch.uint8).contiguous())
        # (the normalised loss reads rec on the masked patches only: the eval-mode tail is evaluated there)
        need = ops.MaskInfo((1 - mi.t).contiguous(), n_active=(B * (L - m.len_keep)) if mask is None else None) if spec.input_size[0] // spec.fmap[0] == 16 else None
        rec = engine.forward(spec, m._W, m._pack, x, mi, train=False, needed_patches=need)
        _, _, _, info = ops.patch_loss_fwd(x, rec, mi, normalized=True)
        return info[0:1]

    def nonfinite_step(self) -> Optional[int]:
        """None, or the optimizer step (1-based, as `step_count` counts them) whose loss / gradient norm was not finite: that step and
        every later one left weights, Adam moments, teacher and BatchNorm buffers untouched (am_adamw_ema `guard`).  ONE host
        synchronisation -- the driver calls it where the reference's per-step `loss.item()` check would sit, once per epoch."""
        g = self.guard.tolist()
        return self._guard_step0 + g[1] if g[0] else None

    def reset_guard(self):
        self.guard.zero_()
        self._guard_step0 = self.step_count

    def set_epoch(self, i: int):
        """per-epoch EMA decay ramp (P/pretrain_AntoMask.py:383-386)."""
        self.teacher.decay = ema_decay_for_epoch(i, self.total_epochs)
