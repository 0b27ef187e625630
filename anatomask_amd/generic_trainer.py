"""AnatoMask / plain-SparK training step for a SparK around ANY zoo-converted backbone (MedNeXt, ConvNeXt, ... --
`SparseEncoder(<dense backbone>)`, P/MedNeXt_head.py:172-188 + P/encoder3D.py:236-276), the step body of P/pretrain_AntoMask.py:418-441:

    mask1 -> teacher fwd (EMA model, eval, no grad) -> raw per-patch l2 of the masked patches -> hard-mask sampler
          -> student fwd -> normalised masked MSE -> backward -> clip -> AdamW -> EMA update

Forward and backward run under torch autograd through the sparse layer zoo's HIP Functions, the engine's pooled norm / matrix-core
convolutions and LightDecoder's engine node (SparK._forward_generic).  Everything after backward is what the fused STUNet trainer
(trainer.AnatoMaskTrainer) does, on flat fp32 buffers the parameters / gradients / EMA weights are views of:

    ||g||^2 (am_sumsq) -> clip + AdamW + EMA in ONE pass (am_adamw_ema) -> EMA of the float buffers (am_ema) -> integer buffers as timm

and, with a process group, ONE bucketed gradient exchange (<= 64 MB per collective, 1 / world folded into the optimizer pass) between
backward and the optimizer.  (The STUNet trainer overlaps its exchange with its hand-written backward; autograd's backward is opaque,
so this one sends after it.)  Same step semantics and the same returned dictionary as AnatoMaskTrainer.step."""
from typing import Dict, List, Optional

import torch

from . import ops
from .modules import ModelEma, SparK


class GenericTrainer:
    BUCKET_BYTES = 64 << 20

    def __init__(self, model: SparK, lr: float = 1e-4, weight_decay: float = 1e-5, betas=(0.9, 0.999), eps: float = 1e-8,
                 clip: float = 12.0, ema_decay: float = 0.999, total_epochs: int = 1000, guide: bool = True, seed: int = 4321,
                 process_group=None, distributed: Optional[bool] = None, self_distill: bool = True):
        if not getattr(model, "_generic", False):
            raise TypeError("GenericTrainer drives SparK models around zoo-converted backbones; STUNet models take trainer.AnatoMaskTrainer")
        self.model = model
        model.train()
        dev = next(model.parameters()).device
        if dev.type != "cuda":
            raise RuntimeError("GenericTrainer runs on the HIP engine: move the model to a cuda (ROCm) device first")
        self.self_distill = self_distill
        self.lr, self.wd, self.betas, self.eps, self.clip = lr, weight_decay, betas, eps, clip
        self.total_epochs, self.guide = total_epochs, guide
        import torch.distributed as dist
        self.pg = process_group
        self.distributed = (dist.is_available() and dist.is_initialized()) if distributed is None else distributed
        self.world = dist.get_world_size(process_group) if self.distributed else 1
        # ---- flat storage: trainable parameters -> views of `flat`, their gradients -> views of `gflat`
        self._params = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
        self.flat, self._pviews = self._flatten([p for _, p in self._params], dev)
        self.gflat = torch.zeros_like(self.flat)
        self._gviews, o = [], 0
        for _, p in self._params:
            self._gviews.append(self.gflat[o:o + p.numel()].view(p.shape))
            o += (p.numel() + 3) // 4 * 4
        self.n = self.flat.numel()
        if self.distributed and self.world > 1:                      # DDP's construction-time broadcast of rank 0's weights and buffers
            dist.broadcast(self.flat, 0, group=process_group)
            for b in model.buffers():
                dist.broadcast(b, 0, group=process_group)
        model.weights_changed()
        # ---- EMA teacher (timm.utils.ModelEma: deepcopy, eval, requires_grad False), same flat layout
        self.teacher = ModelEma(model, decay=ema_decay)
        t = self.teacher.ema
        tp = dict(t.named_parameters())
        self.tflat, _ = self._flatten([tp[n] for n, _ in self._params], dev)
        self._fbufs = [(e, s) for (_, e), (_, s) in zip(t.named_buffers(), model.named_buffers()) if e.is_floating_point()]
        self._ibufs = [(e, s) for (_, e), (_, s) in zip(t.named_buffers(), model.named_buffers()) if not e.is_floating_point()]
        self.m = torch.zeros(self.n, device=dev)
        self.v = torch.zeros(self.n, device=dev)
        self.sumsq = torch.zeros(1, device=dev, dtype=torch.float64)
        self.gnorm = torch.zeros(1, device=dev)
        self.step_count = 0
        self.gen = torch.Generator(device=dev)
        self.gen.manual_seed(seed + (dist.get_rank(process_group) if self.distributed else 0))
        self.exchange_log: List[int] = []                             # bytes of every collective of the last step
        # parameters that never receive a gradient (levels the decoder never reads, unused mask tokens): torch.optim.AdamW skips a parameter
        # whose .grad is None -- no weight decay, no state (P/pretrain_AntoMask.py:437-440) -- and so does this trainer: the first backward
        # runs with .grad = None everywhere, what is still None afterwards is dead for good (the graph is static), and the fused pass covers
        # the live ranges of the flat buffer only; the teacher's dead entries still take timm's EMA step (decay * t + (1 - decay) * s).
        # Buffers: broadcast once at construction and then per rank, as the reference's DDP(broadcast_buffers=False) leaves them.
        self._live_ranges: Optional[List[tuple]] = None
        self._dead_ranges: List[tuple] = []
        self.dead_parameters: List[str] = []

    @staticmethod
    def _flatten(params, dev):
        tot = sum((p.numel() + 3) // 4 * 4 for p in params)
        flat = torch.zeros(tot, device=dev, dtype=torch.float32)
        views, o = [], 0
        for p in params:
            v = flat[o:o + p.numel()].view(p.shape)
            v.copy_(p.data.float())
            p.data = v
            p.grad = None
            views.append(v)
            o += (p.numel() + 3) // 4 * 4
        return flat, views

    def _rebind(self):
        """.to() / load_state_dict(assign=True) may have replaced parameter storage: the step needs the views back"""
        for (_, p), v in zip(self._params, self._pviews):
            if p.data_ptr() != v.data_ptr():
                v.copy_(p.data.float())
                p.data = v

    @property
    def grad_scale(self) -> float:
        return 1.0 / self.world

    def _exchange(self):
        self.exchange_log.clear()
        if not (self.distributed and self.world > 1):
            return
        import torch.distributed as dist
        step = self.BUCKET_BYTES // 4
        works = []
        for a in range(0, self.n, step):
            piece = self.gflat[a:min(a + step, self.n)]
            works.append(dist.all_reduce(piece, group=self.pg, async_op=True))
            self.exchange_log.append(piece.numel() * 4)
        for w in works:
            w.wait()                                                  # the buffer holds the SUM over ranks; 1 / world is folded into am_adamw_ema

    # ------------------------------------------------------------------ one step
    def step(self, inp_bchwd: torch.Tensor, epoch: int = 0, mask1: Optional[torch.Tensor] = None,
             keys: Optional[torch.Tensor] = None, lr: Optional[float] = None, ema_decay: Optional[float] = None) -> Dict[str, torch.Tensor]:
        """inp_bchwd: (B,1,H,W,D) fp32 on the device.  mask1 (B,1,f,f,f) bool / keys (B,L) teacher-force the two random draws.
        Returns device tensors only: {'loss','grad_norm','mask','recon_loss','rec_loss'}."""
        m, t = self.model, self.teacher.ema
        self._rebind()
        B, dev = inp_bchwd.shape[0], inp_bchwd.device
        L = m.fmap_h * m.fmap_w * m.fmap_d
        # 1. first mask (SparK.mask, P/pretrain_AntoMask.py:419) -- drawn on the device, like the fused trainer
        if mask1 is None:
            k1 = torch.rand(B, L, device=dev, generator=self.gen)
            mask1 = ops.mask_sampler(torch.zeros(B, L, device=dev), k1, m.len_keep, 0).bool().view(B, 1, m.fmap_h, m.fmap_w, m.fmap_d)
        mask1 = mask1.to(dev)
        recon = None
        if self.self_distill:
            # 2. teacher pass + raw per-patch l2 of the masked patches (:421-425), 3. hard-mask sampler (:427)
            with torch.no_grad():
                inp1, rec1 = t(inp_bchwd, active_b1ff=mask1)
                recon = ((rec1 - inp1) ** 2).mean(dim=2) * mask1.logical_not().view(B, -1).to(rec1.dtype)
                del inp1, rec1
                if keys is None:
                    keys = torch.rand(B, L, device=dev, generator=self.gen)
                mask, _ = t.generate_mask(recon, guide=self.guide, epoch=epoch, total_epoch=self.total_epochs - 1, keys=keys)
        else:
            mask = mask1
        # 4. student forward + loss (:429-430), 5. backward (:435) into the flat gradient buffer
        self.gflat.zero_()
        first = self._live_ranges is None
        for (_, p), g in zip(self._params, self._gviews):
            p.grad = None if first else g
        inpp, recc = m(inp_bchwd, active_b1ff=mask)
        loss, l2m = m.forward_loss(inpp, recc, mask)
        loss.backward()
        del inpp, recc
        if first:
            self._discover_live()
        self._exchange()
        # 6. clip + AdamW + EMA (:437-440) in one pass over the flat buffers
        with torch.no_grad():
            decay = self.teacher.decay if ema_decay is None else ema_decay
            ops.sumsq(self.gflat, self.sumsq)
            self.step_count += 1
            for a, b in self._live_ranges:
                ops.adamw_ema(self.flat[a:b], self.gflat[a:b], self.m[a:b], self.v[a:b], self.tflat[a:b] if self.self_distill else None, b - a,
                              self.lr if lr is None else lr, self.betas, self.eps, self.wd, self.step_count, self.sumsq, self.clip, decay, self.gnorm,
                              grad_scale=self.grad_scale)
            if self.self_distill:
                for a, b in self._dead_ranges:
                    ops.ema(self.tflat[a:b], self.flat[a:b], decay)
            if self.self_distill:                                      # timm's ModelEma: every state_dict entry, buffers included
                for e, s in self._fbufs:
                    if e.is_contiguous() and s.is_contiguous() and e.dtype == s.dtype == torch.float32:
                        ops.ema(e, s, decay)
                    else:
                        e.copy_(e * decay + (1. - decay) * s)
                for e, s in self._ibufs:
                    e.copy_(e * decay + (1. - decay) * s)
                t.weights_changed()
            m.weights_changed()
        return {"loss": loss.detach().reshape(1), "grad_norm": self.gnorm, "mask": mask, "recon_loss": recon, "rec_loss": l2m.detach()}

    def _discover_live(self):
        """after the FIRST backward (run with .grad = None): parameters autograd left without a gradient are dead; the others' gradients move
        into their views of the flat buffer, which autograd accumulates into from now on"""
        spans, o = [], 0
        for (n, p), g in zip(self._params, self._gviews):
            num = (p.numel() + 3) // 4 * 4
            live = p.grad is not None
            if live:
                g.copy_(p.grad)
            else:
                self.dead_parameters.append(n)
            p.grad = g
            if spans and spans[-1][2] == live:
                spans[-1][1] = o + num
            else:
                spans.append([o, o + num, live])
            o += num
        self._live_ranges = [(a, b) for a, b, live in spans if live]
        self._dead_ranges = [(a, b) for a, b, live in spans if not live]

    def set_epoch(self, i: int):
        """per-epoch EMA decay ramp (P/pretrain_AntoMask.py:383-386)."""
        from .modules import ema_decay_for_epoch
        self.teacher.decay = ema_decay_for_epoch(i, self.total_epochs)

    # ------------------------------------------------------------------ checkpoint (same keys as the reference's torch.save, P/pretrain_AntoMask.py:472-479)
    def state_dict(self) -> dict:
        return {"model": self.model.state_dict(), "state_dict_ema": self.teacher.ema.state_dict(), "exp_avg": self.m.clone(), "exp_avg_sq": self.v.clone(),
                "step": self.step_count, "generator": self.gen.get_state(), "dead_parameters": list(self.dead_parameters),
                "live_ranges": None if self._live_ranges is None else [list(r) for r in self._live_ranges]}

    def save(self, path: str, epoch: int = 0, extra: Optional[dict] = None):
        """one file, tensors + plain containers only: `load` reads it with torch's restricted unpickler (weights_only=True)"""
        sd = self.state_dict()
        sd = {k: ({n: t.detach().cpu() for n, t in v.items()} if isinstance(v, dict) else (v.detach().cpu() if torch.is_tensor(v) else v)) for k, v in sd.items()}
        sd["current_epoch"] = int(epoch)
        sd["extra"] = dict(extra or {})
        torch.save(sd, path)

    def load(self, path: str) -> dict:
        sd = torch.load(path, map_location="cpu", weights_only=True)      # never unpickles code
        self.load_state_dict(sd)
        return {"current_epoch": int(sd.get("current_epoch", 0)), "extra": sd.get("extra", {})}

    def load_state_dict(self, sd: dict):
        self.model.load_state_dict(sd["model"]); self.teacher.ema.load_state_dict(sd["state_dict_ema"])
        self._rebind()
        tp = dict(self.teacher.ema.named_parameters())
        o = 0
        for n, p in self._params:                                      # the teacher's parameters are views of tflat as well
            self.tflat[o:o + p.numel()].view(p.shape).copy_(tp[n].data)
            tp[n].data = self.tflat[o:o + p.numel()].view(p.shape)
            o += (p.numel() + 3) // 4 * 4
        self.m.copy_(sd["exp_avg"]); self.v.copy_(sd["exp_avg_sq"])
        self.step_count = int(sd["step"])
        self.gen.set_state(sd["generator"].cpu() if torch.is_tensor(sd["generator"]) else sd["generator"])
        if sd.get("live_ranges") is not None:                          # (a file written before the first step has none: the next step discovers them)
            self.dead_parameters = list(sd.get("dead_parameters", []))
            self._live_ranges = [tuple(r) for r in sd["live_ranges"]]
            live, self._dead_ranges, o = sorted(self._live_ranges), [], 0
            for a, b in live:
                if a > o:
                    self._dead_ranges.append((o, a))
                o = b
            if o < self.n:
                self._dead_ranges.append((o, self.n))
        self.model.weights_changed(); self.teacher.ema.weights_changed()
