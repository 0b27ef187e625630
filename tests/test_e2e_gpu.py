"""GPU: the whole HIP path (engine + module API + fused trainer) against the reference-generated fixtures
(tests/golden/*.npz: reference-initialiser weights, CT-like volumes) and the CPU oracle.  Tolerances (floors measured by
tests/calibrate_tolerances.py, stated next to the constants in tests/test_oracle_golden.py):
  fp32 forward quantities (stage maps, rec, per-patch l2, loss): <= 2e-4 relative (fixture-vs-HIP), fp32 floor ~2e-5
  fp32 gradients end-to-end, EVERY parameter tensor vs the reference's gradient: relative L2 <= 1.5e-2 (median over tensors
      <= 6e-3), cosine >= 0.9998 -- the reference's own fp32-vs-fp64 floor is 6e-3 / 2.6e-3 / 0.99998
  one optimizer step: update error median <= 1e-3, <= 0.3 % sign-flipped elements; N steps: statistical (chaotic in the reference too)
  bf16 storage: loss within 2e-3; per-tensor gradient error bounded by what an IDEAL bf16-storage evaluation of the same graph
      (oracle.storage("bf16")) shows against the same reference gradient, times a stated factor
  learning: 120 steps on one fixed batch follow the reference's loss curve and end below 0.1 (fp32 and bf16)
"""
import os

import numpy as np
import pytest
import torch

from oracle import anatomask_oracle as O
from tests.helpers import assert_checks, fixture_weights, grad_errors, load, np_volume, rel_err, sample, tiny_cfg
from tests.test_oracle_golden import (GRAD_COS_MIN, GRAD_RTOL, GRAD_RTOL_MEDIAN, NSTEP_UPDATE_MEDIAN, NSTEP_WEIGHT_MAX, NSTEP_WEIGHT_MEDIAN,
                                      STEP1_FLIPPED, STEP1_UPDATE_MEDIAN, _zero_grad_bias, delta_metrics, mismatch_fraction)

pytestmark = pytest.mark.gpu


def _free_port():
    """a port the OS hands out now (fixed rendezvous ports collide when two test runs share a host)."""
    from anatomask_amd.launch import free_port
    return free_port()

DEV = "cuda:0"


@pytest.fixture(scope="module")
def fwd():
    return load("forward_tiny.npz")


def make_model(cfg, W, dtype=torch.float32):
    from anatomask_amd import modules as M
    m = M.build_spark(cfg.dims, cfg.depth, cfg.width, cfg.input_size, cfg.mask_ratio, compute_dtype=dtype)
    m.load_state_dict({k: v.clone() for k, v in W.items()})
    return m.to(DEV)


def feats_ncdhw(f, mask, cfg):
    out = []
    for s, t in enumerate(f):
        v = t.float().cpu().permute(0, 4, 1, 2, 3)
        out.append(torch.where(O.upsample_mask(mask, v.shape[2:]), v, torch.zeros(())))
    return out


def test_forward_matches_reference_fixture(fwd):
    from anatomask_amd import engine, ops
    cfg = tiny_cfg(fwd)
    W = fixture_weights(cfg, fwd)
    x = np_volume(int(fwd["B"]), cfg.input_size, fwd["x_seed"])
    mask = torch.from_numpy(fwd["fwd_mask"])
    m = make_model(cfg, W).train()
    m._ensure_flat()
    mi = ops.MaskInfo.from_bool(mask, DEV)
    with torch.no_grad():
        rec, feats = engine.forward(m.spec, m._W, m._pack, x[:, 0].contiguous().to(DEV), mi, True, None, want_feats=True)
    for i, f in enumerate(feats_ncdhw(feats, mask, cfg)):
        assert_checks(f, fwd[f"enc{i}_checks"], 2e-4, f"enc{i}")
        assert rel_err(sample(f), fwd[f"enc{i}_sample"]) < 2e-4
    rp = O.patchify(cfg, rec.cpu().unsqueeze(1))
    assert_checks(rp, fwd["fwd_rec_checks"], 2e-4, "rec")
    assert rel_err(sample(rp, 256), fwd["fwd_rec_sample"]) < 2e-4
    # BN running stats after one train forward
    sd = m.state_dict()
    for k in fwd:
        if k.startswith("bn1::"):
            np.testing.assert_allclose(sd[k[5:]].cpu().numpy(), fwd[k], rtol=1e-4, atol=1e-6)
    # eval-mode (teacher) forward on fresh weights
    m2 = make_model(cfg, W).eval()
    with torch.no_grad():
        ip, rp2 = m2(x.to(DEV), active_b1ff=mask.to(DEV))
    assert_checks(rp2.cpu(), fwd["eval_rec_checks"], 2e-4, "eval rec")
    tl2 = (((rp2 - ip) ** 2).mean(dim=2) * mask.to(DEV).logical_not().int().view(ip.shape[0], -1)).cpu().numpy()
    assert rel_err(tl2, fwd["eval_teacher_l2"]) < 2e-4


def test_module_api_loss_and_grads(fwd):
    """Reference calling convention: inp,rec = model(x, active_b1ff=mask); loss,_ = model.forward_loss(...); loss.backward()."""
    cfg = tiny_cfg(fwd)
    W = fixture_weights(cfg, fwd)
    x = np_volume(int(fwd["B"]), cfg.input_size, fwd["x_seed"]).to(DEV)
    mask = torch.from_numpy(fwd["fwd_mask"]).to(DEV)
    m = make_model(cfg, W).train()
    inp, rec = m(x, active_b1ff=mask)
    assert inp.shape == rec.shape == (x.shape[0], cfg.L, 4096)
    loss, rec_loss = m.forward_loss(inp, rec, mask)
    assert abs(loss.item() - float(fwd["fwd_loss"])) < 2e-4 * abs(float(fwd["fwd_loss"]))
    assert rel_err(rec_loss.detach().cpu().numpy(), fwd["fwd_l2"]) < 2e-4
    loss.backward()
    grads = {k: p.grad for k, p in m.named_parameters()}
    for k, gn in zip(fwd["grad_keys"], fwd["grad_norms"]):
        g = grads[str(k)]
        if gn < 0:
            assert g is None, k                       # dead densify[4] tensors
            continue
        if gn < 1e-6:
            assert float(g.norm()) < 1e-4, k          # analytically zero (conv bias under a norm)
            continue
        assert abs(float(g.norm()) - gn) < GRAD_RTOL * gn, (k, float(g.norm()), gn)
    for k in fwd:
        if k.startswith("grad::"):
            scale = np.abs(fwd[k]).max()
            if scale > 1e-6:
                assert np.abs(grads[k[6:]].cpu().numpy() - fwd[k]).max() < GRAD_RTOL * scale, k
    errs = grad_errors(grads, fwd)                       # every live tensor vs the reference's gradient
    assert len(errs) == 102
    live = {k: v for k, v in errs.items() if v[1] is not None}
    worst = max(live.items(), key=lambda kv: kv[1][0])
    print("HIP fp32 gradient vs reference: median rel %.2e  max %.2e (%s)  min cos %.6f"
          % (np.median([v[0] for v in live.values()]), worst[1][0], worst[0], min(v[1] for v in live.values())))
    for k, (e, c) in live.items():
        assert e < GRAD_RTOL and c > GRAD_COS_MIN, (k, e, c)
    assert np.median([v[0] for v in live.values()]) < GRAD_RTOL_MEDIAN


def test_gradients_vs_oracle_same_box(fwd):
    """Same check against the CPU oracle evaluated here (cosine + norm per tensor)."""
    cfg = tiny_cfg(fwd)
    W = O.seeded_state(cfg, 3)
    x = np_volume(2, cfg.input_size, 77)
    mask = O.random_mask(cfg, 2, torch.Generator().manual_seed(5))
    loss_o, _, g_o, _ = O.student_loss_and_grads(cfg, W, x, mask)
    m = make_model(cfg, W).train()
    inp, rec = m(x.to(DEV), active_b1ff=mask.to(DEV))
    loss, _ = m.forward_loss(inp, rec, mask.to(DEV))
    loss.backward()
    assert abs(loss.item() - float(loss_o)) < 2e-4 * abs(float(loss_o))
    for k, p in m.named_parameters():
        if g_o[k] is None or float(g_o[k].norm()) < 1e-6:
            continue
        a, b = p.grad.cpu().double().flatten(), g_o[k].double().flatten()
        cos = float((a * b).sum() / (a.norm() * b.norm()))
        assert cos > GRAD_COS_MIN and float((a - b).norm() / b.norm()) < GRAD_RTOL, (k, cos, float(a.norm()), float(b.norm()))


def _run_trainer(dtype, r, f, teacher_force_student_mask=False, f32_split=False):
    from anatomask_amd.trainer import AnatoMaskTrainer
    cfg = tiny_cfg(f)
    W0 = fixture_weights(cfg, f)
    m = make_model(cfg, W0, dtype)
    ep, tot = (int(v) for v in r["epoch"])
    tr = AnatoMaskTrainer(m, lr=float(r["lr"]), ema_decay=float(r["ema_decay"]), total_epochs=tot + 1, distributed=False, f32_split=f32_split)
    out, snap = [], None
    for s in range(int(r["N"])):
        x = np_volume(int(f["B"]), cfg.input_size, r["x_seeds"][s]).to(DEV)
        keys = torch.from_numpy(r["keys"][s])
        if teacher_force_student_mask:               # keys = 0 for the reference's visible set, ties by id
            keys = torch.from_numpy(1.0 - r["mask"][s].reshape(keys.shape).astype(np.float32))
        o = tr.step(x, epoch=ep, mask1=torch.from_numpy(r["mask1"][s]), keys=keys)
        out.append({k: v.clone().cpu() for k, v in o.items()})
        if s == 0:
            snap = ({k: v.detach().cpu().clone() for k, v in m.state_dict().items()},
                    {k: v.detach().cpu().clone() for k, v in tr.teacher.ema.state_dict().items()})
    return cfg, W0, m, tr, out, snap


@pytest.mark.parametrize("f32_split", [False, True])
def test_trainer_n_steps_fp32_matches_reference(f32_split):
    """f32_split=True: the same N reference steps with the matrix-core products taken from bf16 hi / lo splits (AM_DT_F32S, the fast
    reference-precision mode) -- SAME bounds as the exact-fp32 mode."""
    from anatomask_amd import ops
    r, f = load("train_tiny.npz"), load("forward_tiny.npz")
    cfg, W0, m, tr, out, snap = _run_trainer(torch.float32, r, f, f32_split=f32_split)
    names = [str(n) for n in r["names"]]
    for s, o in enumerate(out):
        assert np.array_equal(o["mask"].numpy().astype(bool).reshape(r["mask"][s].shape), r["mask"][s]), f"sampler mask diverged at step {s}"
        # the trajectories decorrelate step by step (chaos floor: the reference's own fp32-vs-fp64 run is 1.4e-4 off at step 4)
        assert abs(o["loss"].item() - r["losses"][s]) < (3e-4 if s < 3 else 1e-3) * abs(r["losses"][s]), (s, o["loss"].item(), r["losses"][s])
        # grad-norm: 1e-3 at the first step; later the CPU oracle itself is up to 4e-2 off the reference (same chaos), and a change of the
        # summation ORDER inside one kernel (the DPP row reduction of the conv statistics epilogue) moved step 2 from 1.5e-2 to 2.8e-2
        assert abs(o["grad_norm"].item() - r["grad_norms"][s]) < (1e-3 if s == 0 else 4e-2 if s < 3 else 0.1) * r["grad_norms"][s], (s, o["grad_norm"].item())
        assert rel_err(o["recon_loss"].numpy(), r["recon"][s]) < (1e-3 if s < 3 else 5e-3)
    errs, mfs, mfe = [], [], []
    for k in names:                                                   # strict after ONE step
        if "step1delta::" + k not in r or _zero_grad_bias(k) or np.linalg.norm(r["step1delta::" + k]) < 1e-9:
            continue
        e, c = delta_metrics(snap[0][k] - W0[k], r["step1delta::" + k])
        errs.append(e)
        n_el = min(1024, W0[k].numel())
        mfs.append((mismatch_fraction(snap[0][k] - W0[k], r["step1delta::" + k]) * n_el, n_el))
        mfe.append((mismatch_fraction(snap[1][k] - W0[k], r["step1ema::" + k]) * n_el, n_el))
        assert c > 0.7, ("step1", k, e, c)
    print("HIP fp32 first update vs reference: median rel %.2e, flipped %.2e" % (np.median(errs), sum(a for a, _ in mfs) / sum(b for _, b in mfs)))
    assert np.median(errs) < STEP1_UPDATE_MEDIAN, np.median(errs)
    for mm in (mfs, mfe):
        # (Adam's first update is lr * sign(g): an element flips when a perturbation exceeds |g|.  The split products perturb the
        # gradients at the 1e-5 level instead of the exact mode's 1e-7: measured 4.1e-3 of the elements against 2.0e-3, stated bound 2x)
        assert sum(a for a, _ in mm) / sum(b for _, b in mm) <= (2 * STEP1_FLIPPED if f32_split else STEP1_FLIPPED)
    fsd = {k: v.cpu() for k, v in m.state_dict().items()}
    esd = {k: v.cpu() for k, v in tr.teacher.ema.state_dict().items()}
    errs, werrs = [], []
    for k in names:
        if "final::" + k in r:
            assert np.array_equal(fsd[k].numpy(), r["final::" + k]), k
            assert np.array_equal(esd[k].numpy(), r["ema::" + k]), k
            continue
        if _zero_grad_bias(k) or np.linalg.norm(r["finaldelta::" + k]) < 1e-9:
            continue
        e, c = delta_metrics(fsd[k] - W0[k], r["finaldelta::" + k])
        e2, c2 = delta_metrics(esd[k] - W0[k], r["emadelta::" + k])
        errs.append(e)
        werrs.append(float((fsd[k] - W0[k]).norm()) * e / (float(W0[k].norm()) + 1e-30))
        if W0[k].numel() >= 64:
            assert e < 0.9 and c > 0.5, ("final", k, e, c)
            assert e2 < 0.9 and c2 > 0.5, ("ema", k, e2, c2)
    print("HIP fp32 vs reference, update L2 error after N steps: median %.3f max %.3f; weight-level median %.2e max %.2e"
          % (np.median(errs), max(errs), np.median(werrs), max(werrs)))
    assert np.median(errs) < NSTEP_UPDATE_MEDIAN and np.median(werrs) < NSTEP_WEIGHT_MEDIAN and max(werrs) < NSTEP_WEIGHT_MAX


def _emulated_bf16_errors(cfg, W, x, mask, fwd):
    """per-tensor (rel, cos) of an IDEAL bf16-storage evaluation of the reference graph (CPU oracle under storage("bf16"):
    fp32 arithmetic, every C > 1 activation / gradient / MFMA weight copy rounded to bf16 where the HIP path stores it)
    against the reference's fp32 gradient."""
    with O.storage("bf16"):
        loss, _, g, _ = O.student_loss_and_grads(cfg, W, x, mask)
    return float(loss), grad_errors({k: v for k, v in g.items() if v is not None}, fwd)


def test_bf16_gradients_no_worse_than_ideal_bf16_storage(fwd):
    """bf16 STORAGE (the benchmarked precision), one student forward + backward on the reference fixture.
    bf16 rounding (2^-9 relative) flips ~1000x more LeakyReLU(0.01)/ReLU6 gates than fp32 rounding does, so NO bf16-storage
    evaluation of this graph reproduces the fp32 gradient closely: the ideal emulation sits at a median relative L2 error of
    ~0.27 per tensor (cos 0.75..0.97), against 2.3e-2 / cos 0.999 when the activations are smooth.  The assertion is therefore
    relative to that emulation, tensor by tensor: the HIP path must not be further from the reference than BF16_FACTOR x the
    ideal bf16-storage evaluation (+ a small absolute slack for tensors the emulation happens to hit well) -- a wrong tap,
    a missing term or a wrong scale in any backward kernel fails it (errors >= 1, cos <= 0.5 on the affected tensors)."""
    BF16_FACTOR, SLACK = 1.6, 0.08
    cfg = tiny_cfg(fwd)
    W = fixture_weights(cfg, fwd)
    x = np_volume(int(fwd["B"]), cfg.input_size, fwd["x_seed"])
    mask = torch.from_numpy(fwd["fwd_mask"])
    loss_e, emu = _emulated_bf16_errors(cfg, W, x, mask, fwd)
    m = make_model(cfg, W, torch.bfloat16).train()
    inp, rec = m(x.to(DEV), active_b1ff=mask.to(DEV))
    loss, _ = m.forward_loss(inp, rec, mask.to(DEV))
    loss.backward()
    ref_loss = float(fwd["fwd_loss"])
    assert abs(loss.item() - ref_loss) < 2e-3 * ref_loss, (loss.item(), ref_loss, loss_e)
    errs = grad_errors({k: p.grad for k, p in m.named_parameters() if p.grad is not None}, fwd)
    live = [k for k, v in errs.items() if v[1] is not None]
    med_h, med_e = np.median([errs[k][0] for k in live]), np.median([emu[k][0] for k in live])
    print("bf16 gradient vs reference: HIP median rel %.3f (min cos %.3f) | ideal bf16-storage emulation median rel %.3f (min cos %.3f)"
          % (med_h, min(errs[k][1] for k in live), med_e, min(emu[k][1] for k in live)))
    assert med_h < BF16_FACTOR * med_e, (med_h, med_e)
    for k in live:
        assert errs[k][0] < BF16_FACTOR * emu[k][0] + SLACK, (k, errs[k], emu[k])
        assert errs[k][1] > min(emu[k][1] - 0.15, 0.9), (k, errs[k], emu[k])
    gh = float(torch.cat([p.grad.flatten() for p in m.parameters() if p.grad is not None]).norm())
    assert abs(gh - float(np.linalg.norm(fwd["grad_norms"][fwd["grad_norms"] > 0]))) < 0.1 * gh


def _nstep_update_errors(state, W0, r):
    errs, werrs = [], []
    for k in [str(n) for n in r["names"]]:
        if "finaldelta::" + k not in r or _zero_grad_bias(k) or np.linalg.norm(r["finaldelta::" + k]) < 1e-9:
            continue
        e, _ = delta_metrics(state[k] - W0[k], r["finaldelta::" + k])
        errs.append(e)
        werrs.append(float((state[k] - W0[k]).norm()) * e / (float(W0[k].norm()) + 1e-30))
    return np.median(errs), max(errs), np.median(werrs), max(werrs)


def test_trainer_n_steps_bf16_tracks_reference():
    """N teacher-forced steps in bf16 storage vs the reference's fp32 run: loss within 2e-3 at every step, grad-norm within 15 %,
    and the N-step weight update no further from the reference's than 1.3x (median) / 1.5x (max) what the IDEAL bf16-storage
    evaluation of the same N steps (CPU oracle under storage("bf16"), run here) shows -- measured: HIP 0.66 / ideal 0.70 median."""
    r, f = load("train_tiny.npz"), load("forward_tiny.npz")
    cfg, W0, m, tr, out, _ = _run_trainer(torch.bfloat16, r, f, teacher_force_student_mask=True)
    for s, o in enumerate(out):
        assert np.array_equal(o["mask"].numpy().astype(bool).reshape(r["mask"][s].shape), r["mask"][s])
        assert abs(o["loss"].item() - r["losses"][s]) < 2e-3 * abs(r["losses"][s]), (s, o["loss"].item(), r["losses"][s])
        assert abs(o["grad_norm"].item() / r["grad_norms"][s] - 1) < 0.15, (s, o["grad_norm"].item(), r["grad_norms"][s])
    fsd = {k: v.float().cpu() for k, v in m.state_dict().items()}
    assert all(torch.isfinite(v).all() for v in fsd.values())
    ep, tot = (int(v) for v in r["epoch"])
    st = O.StepState(cfg, W0)
    with O.storage("bf16"):
        for s in range(int(r["N"])):
            keys = torch.from_numpy(1.0 - r["mask"][s].reshape(r["keys"][s].shape).astype(np.float32))
            O.train_step(st, np_volume(int(f["B"]), cfg.input_size, r["x_seeds"][s]), torch.from_numpy(r["mask1"][s]), keys, ep, tot,
                         float(r["lr"]), float(r["ema_decay"]))
    h, e = _nstep_update_errors(fsd, W0, r), _nstep_update_errors(st.student, W0, r)
    print("bf16 N-step update error vs reference (median, max, weight-level median, max): HIP %.3f %.3f %.2e %.2e | ideal bf16-storage %.3f %.3f %.2e %.2e"
          % (*h, *e))
    assert h[0] < 1.3 * e[0] and h[1] < 1.5 * e[1] and h[2] < 1.3 * e[2] and h[3] < 1.5 * e[3], (h, e)


@pytest.mark.parametrize("dtype,track,final", [(torch.float32, 2e-3, 0.1), (torch.bfloat16, 2e-2, 0.12)])
def test_overfit_one_batch_follows_reference_curve(dtype, track, final):
    """Does the step LEARN?  120 full AnatoMask steps (teacher, sampler, student, backward, clip, AdamW, EMA) on one fixed CT-like
    batch, the random draws teacher-forced from the reference's run of the same thing (tests/golden/overfit_tiny.npz: loss
    1.005 -> 0.044).  The first 5 steps must follow the reference's loss curve within `track` (relative); after that the
    sampler's hard-patch set flips with 1e-7 differences and the trajectories decorrelate (the CPU oracle itself is 1e-2..4e-2
    off the reference from step 7 on), so the curve is bounded instead: every 10-step mean within 25 % (+ 0.01) of the reference's, and
    the last 10 steps average below `final`."""
    from anatomask_amd.trainer import AnatoMaskTrainer
    ov, f = load("overfit_tiny.npz"), load("forward_tiny.npz")
    cfg = tiny_cfg(f)
    m = make_model(cfg, fixture_weights(cfg, f), dtype)
    ep, tot = (int(v) for v in ov["epoch"])
    tr = AnatoMaskTrainer(m, lr=float(ov["lr"]), ema_decay=float(ov["ema_decay"]), total_epochs=tot + 1, distributed=False)
    x = np_volume(int(f["B"]), cfg.input_size, ov["x_seed"]).to(DEV)
    losses = []
    for s in range(int(ov["N"])):
        o = tr.step(x, epoch=ep, mask1=torch.from_numpy(ov["mask1"][s]), keys=torch.from_numpy(ov["keys"][s]))
        losses.append(o["loss"])
    losses = torch.cat(losses).cpu().numpy()
    ref = ov["losses"]
    assert np.isfinite(losses).all()
    assert np.abs(losses[:5] / ref[:5] - 1).max() < track, (losses[:5], ref[:5])
    bl, rf = losses.reshape(-1, 10).mean(1), ref.reshape(-1, 10).mean(1)
    blocks = bl / rf
    print("overfit", dtype, "loss[::10]", np.round(losses[::10], 4), "block ratio to reference", np.round(blocks, 3))
    # within 25 % of the reference's block mean, plus 0.01 absolute for the late blocks (loss 0.04-0.06, where the decorrelated
    # trajectories wander by more than a quarter of so small a value once in ~8 runs)
    assert (np.abs(bl - rf) <= 0.25 * rf + 0.01).all(), blocks
    assert losses[-10:].mean() < final, losses[-10:]


def test_reference_style_driver_loop(fwd):
    """The unfused drop-in route: torch.optim.AdamW + clip_grad_norm_ + ModelEma.update on our modules,
    exactly as P/pretrain_AntoMask.py:418-441 drives them; one step compared with the fused trainer."""
    from anatomask_amd import modules as M
    from anatomask_amd.trainer import AnatoMaskTrainer
    r = load("train_tiny.npz")
    cfg = tiny_cfg(fwd)
    W0 = fixture_weights(cfg, load("forward_tiny.npz"))
    ep, tot = (int(v) for v in r["epoch"])
    x = np_volume(int(fwd["B"]), cfg.input_size, r["x_seeds"][0]).to(DEV)
    mask1 = torch.from_numpy(r["mask1"][0]).to(DEV)
    model_without_ddp = make_model(cfg, W0)
    model_ema = M.ModelEma(model_without_ddp, decay=float(r["ema_decay"]), device=DEV, resume="")
    model = M.LocalDDP(model_without_ddp)
    optimizer = torch.optim.AdamW(M.get_param_groups(model_without_ddp, nowd_keys={"cls_token", "pos_embed", "mask_token", "gamma"}),
                                  lr=float(r["lr"]), weight_decay=1e-5, betas=(0.9, 0.999))
    model.train()
    with torch.no_grad():
        inp1, rec1 = model_ema.ema(x, active_b1ff=mask1)
        l2_loss = ((rec1 - inp1) ** 2).mean(dim=2, keepdim=False)
        recon_loss = l2_loss * mask1.logical_not().int().view(mask1.shape[0], -1)
    mask, _ = model_ema.ema.generate_mask(recon_loss, guide=True, epoch=ep, total_epoch=tot, keys=torch.from_numpy(r["keys"][0]).to(DEV))
    assert np.array_equal(mask.cpu().numpy(), r["mask"][0])
    inpp, recc = model(x, active_b1ff=mask, vis=False)
    loss, _ = model.module.forward_loss(inpp, recc, mask)
    optimizer.zero_grad()
    loss.backward()
    gn = torch.nn.utils.clip_grad_norm_(model.parameters(), 12).item()
    optimizer.step()
    model_without_ddp.weights_changed()
    model_ema.update(model)
    assert abs(loss.item() - r["losses"][0]) < 3e-4 * abs(r["losses"][0])
    assert abs(gn - r["grad_norms"][0]) < 5e-2 * r["grad_norms"][0]
    # the fused trainer takes the same step
    m2 = make_model(cfg, W0)
    tr = AnatoMaskTrainer(m2, lr=float(r["lr"]), ema_decay=float(r["ema_decay"]), total_epochs=tot + 1, distributed=False)
    tr.step(x, epoch=ep, mask1=mask1, keys=torch.from_numpy(r["keys"][0]))
    flips = tot_el = 0
    for (k, a), (_, b) in zip(model_without_ddp.state_dict().items(), m2.state_dict().items()):
        if a.is_floating_point() and not _zero_grad_bias(k) and k not in model_without_ddp._dead:
            d = (a - b).abs().flatten()
            flips += int((d > 1e-4).sum()); tot_el += d.numel()       # |update| = lr = 1e-3: count sign flips
    assert flips / tot_el < 5e-3, flips / tot_el
    # checkpoint keys as the reference writes them (P/pretrain_AntoMask.py:472-479)
    keys = list(model.state_dict().keys())
    assert all(k.startswith("module.") for k in keys) and len(keys) == 131
    assert "module.sparse_encoder.sp_cnn.conv_blocks_context.0.0.conv1.weight" in keys


def test_depth2_blocks_identity_shortcut_vs_oracle():
    """STUNet-L/H style stages (depth > 1): blocks after the first have no 1x1 conv, the residual is the block input
    (P/STUNet_head.py:99-103).  Forward + gradients against the CPU oracle on the same box."""
    from anatomask_amd import modules as M
    cfg = O.Config([8, 16, 32, 64, 128, 128], [2, 2, 2, 2, 2, 2], 128, (32, 32, 48), 0.6)
    W = O.seeded_state(cfg, 11)
    assert len(W) == 131 + 5 * 8                       # 8 more entries per extra block
    x = np_volume(2, cfg.input_size, 91)
    mask = O.random_mask(cfg, 2, torch.Generator().manual_seed(9))
    loss_o, rl_o, g_o, _ = O.student_loss_and_grads(cfg, W, x, mask)
    m = M.build_spark(cfg.dims, cfg.depth, cfg.width, cfg.input_size, cfg.mask_ratio)
    assert list(m.state_dict().keys()) == list(W.keys())
    m.load_state_dict({k: v.clone() for k, v in W.items()})
    m = m.to(DEV).train()
    inp, rec = m(x.to(DEV), active_b1ff=mask.to(DEV))
    loss, rl = m.forward_loss(inp, rec, mask.to(DEV))
    assert abs(loss.item() - float(loss_o)) < 2e-4 * abs(float(loss_o)), (loss.item(), float(loss_o))
    assert rel_err(rl.detach().cpu().numpy(), rl_o.numpy()) < 2e-4
    loss.backward()
    # 12 residual blocks: twice the LeakyReLU gates of the depth-1 nets, and any change of the fp32 summation order inside a
    # conv (tap order, tiling) flips a different handful of them.  Direction must hold for every tensor; the norm
    # must be within GRAD_RTOL for at least 90 % of the large tensors and within 2x GRAD_RTOL for all of them.
    n_big = n_tight = 0
    for k, p in m.named_parameters():
        if g_o[k] is None:
            assert p.grad is None, k
            continue
        if float(g_o[k].norm()) < 1e-6:
            continue
        a, b = p.grad.cpu().double().flatten(), g_o[k].double().flatten()
        cos = float((a * b).sum() / (a.norm() * b.norm()))
        dev_ = abs(float(a.norm() / b.norm()) - 1)
        big = a.numel() >= 256
        # 12 residual blocks: the reference's own fp32-vs-fp64 floor on this net is cos 0.99987, norm 7.7e-3, rel 1.6e-2
        assert cos > 0.999 and dev_ < 2 * GRAD_RTOL, (k, cos, float(a.norm()), float(b.norm()))
        if big:
            n_big += 1
            n_tight += dev_ < GRAD_RTOL
    assert n_tight >= 0.9 * n_big, (n_tight, n_big)


def test_trainer_distributed_path_single_rank_nccl():
    """The RCCL exchange path with world_size 1 (all that one GPU allows): broadcast, per-group async all-reduce from the
    backward hook, averaging; must give the same step as the non-distributed trainer."""
    import os
    import torch.distributed as dist
    from anatomask_amd.trainer import AnatoMaskTrainer
    r, f = load("train_tiny.npz"), load("forward_tiny.npz")
    cfg = tiny_cfg(f)
    W0 = fixture_weights(cfg, load("forward_tiny.npz"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", str(_free_port()))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(DEV))
    try:
        ep, tot = (int(v) for v in r["epoch"])
        x = np_volume(int(f["B"]), cfg.input_size, r["x_seeds"][0]).to(DEV)
        outs = []
        for distributed in (False, True):
            m = make_model(cfg, W0)
            tr = AnatoMaskTrainer(m, lr=float(r["lr"]), ema_decay=float(r["ema_decay"]), total_epochs=tot + 1, distributed=distributed)
            o = tr.step(x, epoch=ep, mask1=torch.from_numpy(r["mask1"][0]), keys=torch.from_numpy(r["keys"][0]))
            outs.append((o["loss"].item(), o["grad_norm"].item()))
        assert abs(outs[0][0] - outs[1][0]) < 1e-6 and abs(outs[0][1] - outs[1][1]) < 1e-3 * outs[0][1], outs
        assert abs(outs[1][0] - r["losses"][0]) < 3e-4 * abs(r["losses"][0])
        # the collectives were issued from the dedicated stream; every piece carries its (issue, done) event pair on the DEVICE clock
        tl = tr.exchange_timeline()
        assert len(tl["pieces"]) == len(tr.exchange_log) >= 1 and sum(p["bytes"] for p in tl["pieces"]) == m._live_end * 4
        assert all(p["done_ms"] is not None and 0.0 <= p["issue_ms"] <= p["done_ms"] for p in tl["pieces"])
        assert [p["issue_ms"] for p in tl["pieces"]] == sorted(p["issue_ms"] for p in tl["pieces"])
        assert tl["backward_end_ms"] > 0 and tl["exposed_ms"] >= 0.0
        from anatomask_amd import engine
        comm = AnatoMaskTrainer._comm_stream(torch.device(DEV))
        assert comm.cuda_stream not in (torch.cuda.current_stream().cuda_stream, engine._side_stream(torch.device(DEV)).cuda_stream)
    finally:
        dist.destroy_process_group()


def test_checkpoint_roundtrip_and_finetune_handoff(tmp_path):
    import os
    """Reference checkpoint format (P/pretrain_AntoMask.py:472-479), resume, and the key contract of
    load_stunet_ssl_weights (nnunetv2/run/load_pretrained_weights.py:66-106)."""
    from anatomask_amd import checkpoint
    from anatomask_amd.trainer import AnatoMaskTrainer
    r, f = load("train_tiny.npz"), load("forward_tiny.npz")
    cfg = tiny_cfg(f)
    W0 = fixture_weights(cfg, load("forward_tiny.npz"))
    x = np_volume(int(f["B"]), cfg.input_size, 5).to(DEV)

    def fresh():
        return AnatoMaskTrainer(make_model(cfg, W0), lr=1e-3, ema_decay=0.99, total_epochs=1000, distributed=False, seed=1)
    a = fresh()
    a.step(x, epoch=500)
    p = str(tmp_path / "STUNet_B_head_latest.pt")
    # the data feed's generator states ride along as tensors / python scalars (anatomask_amd.pretrain.rng_state_to_plain) ...
    from anatomask_amd.pretrain import rng_state_to_plain
    fs = {0: {"loader_rng": {0: rng_state_to_plain(np.random.RandomState(3))}, "aug_rng": rng_state_to_plain(np.random.RandomState(4))}}
    checkpoint.save_checkpoint(p, a, [1.0], 0, extra={"ema_loss": 1.0, "feed_state": fs})
    # ... so that the reference's hand-off, a PLAIN torch.load(fname) (weights_only=True from torch 2.6 on), accepts the file
    ck = torch.load(p)
    assert checkpoint.peek_extra(p, "feed_state")[0]["loader_rng"][0]["pos"] == 624
    assert set(["network_weights", "optimizer_state", "grad_scaler_state", "train_loss", "current_epoch"]) <= set(ck)
    assert ck["grad_scaler_state"] is None and len(ck["network_weights"]) == 131
    enc = checkpoint.encoder_weights_for_finetuning(ck["network_weights"])
    assert len(enc) == 50 and "conv_blocks_context.0.0.conv1.weight" in enc and all(k.startswith("conv_blocks_context.") for k in enc)
    # the reference's resume hint (P/pretrain_AntoMask.py:360): torch.optim.AdamW built from get_param_groups accepts 'optimizer_state'
    from anatomask_amd import modules as M
    ref_opt = torch.optim.AdamW(M.get_param_groups(a.model, nowd_keys={"cls_token", "pos_embed", "mask_token", "gamma"}), lr=1e-3, weight_decay=1e-5)
    ref_opt.load_state_dict(ck["optimizer_state"])
    first = ref_opt.param_groups[0]["params"][0]
    assert ref_opt.state[first]["exp_avg"].shape == first.shape and float(ref_opt.state[first]["step"]) == 1.0
    assert torch.equal(ref_opt.state[first]["exp_avg"].cpu(), a.m[:first.numel()].view(first.shape).cpu())
    assert not os.path.exists(p + ".tmp")
    b = fresh()
    assert int(checkpoint.load_checkpoint(p, b)["current_epoch"]) + 1 == 1
    oa, ob = a.step(x, epoch=500), b.step(x, epoch=500)          # same RNG state, weights, moments, teacher -> same step
    assert abs(oa["loss"].item() - ob["loss"].item()) < 1e-6
    assert torch.equal(oa["mask"], ob["mask"])
    bad = tot = 0                                                # resumed and original runs differ by atomic-order noise only;
    for (k, u), (_, v) in zip(a.model.state_dict().items(), b.model.state_dict().items()):   # Adam (lr 1e-3) can flip noise-level elements
        if u.is_floating_point():
            bad += int(((u - v).abs() > 2e-4).sum()); tot += u.numel()
    assert bad / tot < 5e-3, bad / tot


def test_recompute_mode_matches_plain_backward():
    """P/GC.py policy (checkpoint per encoder stage / decoder block): same loss, same gradients, BN buffers updated once."""
    from anatomask_amd import modules as M
    cfg = O.Config([8, 16, 32, 64, 128, 128], [2, 1, 2, 1, 1, 1], 128, (32, 32, 48), 0.6)
    W = O.seeded_state(cfg, 12)
    x = np_volume(2, cfg.input_size, 3).to(DEV)
    mask = O.random_mask(cfg, 2, torch.Generator().manual_seed(4)).to(DEV)
    res = []
    for rc in (False, True):
        m = M.build_spark(cfg.dims, cfg.depth, cfg.width, cfg.input_size, cfg.mask_ratio, recompute=rc)
        m.load_state_dict({k: v.clone() for k, v in W.items()})
        m = m.to(DEV).train()
        inp, rec = m(x, active_b1ff=mask)
        loss, _ = m.forward_loss(inp, rec, mask)
        loss.backward()
        res.append((loss.item(), {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None},
                    {k: v.clone() for k, v in m.state_dict().items() if "running" in k or "num_batches" in k}))
    assert res[0][0] == res[1][0]
    for k in res[0][1]:
        a, b = res[0][1][k], res[1][1][k]
        if a.abs().max().item() < 1e-6:
            continue                                   # analytically-zero conv-bias gradients: float noise
        assert (a - b).abs().max().item() <= 1e-4 * a.abs().max().item(), k      # fp32 atomics order only
    for k in res[0][2]:
        assert torch.equal(res[0][2][k], res[1][2][k]), k


def test_plain_spark_step_vs_oracle():
    """Plain SparK step (P/spark3D.py:98-146 / P/pretrain.py: random mask, loss on the student only, no teacher)."""
    from anatomask_amd.trainer import AnatoMaskTrainer
    f = load("forward_tiny.npz")
    cfg = tiny_cfg(f)
    W0 = fixture_weights(cfg, load("forward_tiny.npz"))
    x = np_volume(2, cfg.input_size, 41)
    mask = O.random_mask(cfg, 2, torch.Generator().manual_seed(8))
    loss_o, _, grads, _ = O.student_loss_and_grads(cfg, {k: v.clone() for k, v in W0.items()}, x, mask)
    live = {k: g for k, g in grads.items() if g is not None}
    gn_o = float(O.clip_grad_norm(live, 12.0))
    tr = AnatoMaskTrainer(make_model(cfg, W0), lr=1e-3, total_epochs=1000, distributed=False, self_distill=False)
    o = tr.step(x.to(DEV), epoch=0, mask1=mask)
    assert torch.equal(o["mask"].cpu().bool().view(mask.shape), mask)
    assert abs(o["loss"].item() - float(loss_o)) < 2e-4 * abs(float(loss_o))
    assert abs(o["grad_norm"].item() - gn_o) < 5e-2 * gn_o


def test_forward_return_feat_and_vis():
    """SparK.forward(return_feat=True) / (vis=True) outputs (P/AnatoMask.py:172-185) vs the oracle's densify / unpatchify."""
    f = load("forward_tiny.npz")
    cfg = tiny_cfg(f)
    W0 = fixture_weights(cfg, load("forward_tiny.npz"))
    x = np_volume(2, cfg.input_size, 43)
    mask = O.random_mask(cfg, 2, torch.Generator().manual_seed(9))
    m = make_model(cfg, W0).eval()
    with torch.no_grad():
        inp, rec, feat = m(x.to(DEV), active_b1ff=mask.to(DEV), return_feat=True)
        vin, vmasked, vrec = m(x.to(DEV), active_b1ff=mask.to(DEV), vis=True)
    p = {k: v.clone() for k, v in W0.items()}
    act = O.upsample_mask(mask, cfg.input_size)
    feats = O.encoder_forward(cfg, p, x * act.float(), mask)
    to_dec = O.densify(cfg, p, feats, mask)
    want_feat = to_dec[0].flatten(start_dim=2).permute(0, 2, 1)
    assert feat.shape == want_feat.shape
    assert (feat.cpu() - want_feat).abs().max() <= 2e-4 * want_feat.abs().max()
    inp_o, rec_o = O.spark_forward(cfg, p, x, mask, False)
    mean, var = inp_o.mean(-1, keepdim=True), (inp_o.var(-1, keepdim=True) + 1e-6) ** .5
    want_rec = torch.where(act, x, O.unpatchify(cfg, rec_o * var + mean))
    assert torch.equal(vin.cpu(), x) and torch.equal(vmasked.cpu(), x * act)
    assert (vrec.cpu() - want_rec).abs().max() <= 2e-4 * want_rec.abs().max()


def test_config1_stunet_small_plain_spark_step():
    """BASELINE.json configs[0]: STUNet-small (dims 16..256, width 256), 48^3 patch, bs 2, plain SparK (random mask, no teacher):
    one fused HIP step vs the CPU oracle step on the same volumes, fp32."""
    from anatomask_amd.trainer import AnatoMaskTrainer
    cfg = O.Config([16, 32, 64, 128, 256, 256], [1] * 6, 256, (48, 48, 48), 0.6)
    W0 = fixture_weights(cfg, load("forward_tiny.npz"))
    x = np_volume(2, cfg.input_size, 51)
    mask = O.random_mask(cfg, 2, torch.Generator().manual_seed(10))
    loss_o, _, grads, _ = O.student_loss_and_grads(cfg, {k: v.clone() for k, v in W0.items()}, x, mask)
    gn_o = float(O.clip_grad_norm({k: g for k, g in grads.items() if g is not None}, 12.0))
    tr = AnatoMaskTrainer(make_model(cfg, W0), lr=1e-4, total_epochs=1000, distributed=False, self_distill=False)
    o = tr.step(x.to(DEV), epoch=0, mask1=mask)
    assert abs(o["loss"].item() - float(loss_o)) < 2e-4 * abs(float(loss_o))
    assert abs(o["grad_norm"].item() - gn_o) < 5e-2 * gn_o


def test_two_ranks_on_one_gpu_stay_in_sync():
    """world_size 2 with the REAL trainer (overlapped per-group exchange, side stream joins) -- both ranks on the one GPU a box has,
    over gloo with device tensors (RCCL refuses two ranks per device).  Different init per rank, different data per rank: after
    3 steps the students and the EMA teachers must be bit-identical across the ranks."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(_free_port()), os.path.join(root, "tools", "ddp_two_ranks_one_gpu.py")],
                       capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert r.stdout.count("weights identical across ranks: True; teacher identical: True") == 2, r.stdout[-2000:]
    # ... and what was exchanged IS the mean of the two ranks' local gradients (recomputed through the module API on each rank's
    # own input and mask from the common start weights), global norm included
    assert r.stdout.count("mean-of-gradients ok: True") == 2, r.stdout[-2000:]


def test_bench_rank_body_rccl_world1():
    """bench.py's N > 1 rank body on the GPU BEFORE an 8-GPU node runs it: `torchrun --nproc-per-node 1` with
    AM_BENCH_FORCE_DIST=1 (set by the launcher's environment, i.e. before the rank touches the GPU) makes the one rank take every
    world > 1 branch over RCCL: init_process_group("nccl") with the collective timeout, the broadcast of parameters and buffers, the
    barriers and the MAX-over-ranks clock of the timed window, the overlapped gradient exchange on the comm stream, the exchange-off
    window, exchange_report with the per-collective timeline, ranks_report's all_gather_object on the nccl backend, the barrier +
    destroy_process_group BEFORE the roofline section.  Ref: P/pretrain_AnatoMask_DDP.py:200,239-240,409-410,484."""
    import json
    import os
    import subprocess
    import sys
    from tests.test_launch import _check_timeline
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(HSA_ENABLE_IPC_MODE_LEGACY="0", AM_BENCH_FORCE_DIST="1", AM_BENCH_COLLECTIVE_TIMEOUT_S="120")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                        "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "2",
                        "--batch", "2", "--no-cpu-baseline", "--no-secondary", "--no-h2d"],
                       capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 1 and out["ranks_seen"] == 1 and out["value"] > 0 and out["config"]["parallelism"] == "dp1"
    ex = out["exchange"]
    assert ex["backend"] == "rccl" and ex["ranks"] == 1 and ex["gradient_bytes_per_step"] > 200e6            # STUNet-B: 53 M live parameters
    assert ex["collectives_per_step"] >= 4 and ex["largest_collective_bytes"] <= 64 << 20
    assert ex["first_collective_after_tag"].startswith("dec")                                              # the exchange starts inside backward
    _check_timeline(ex)
    print("bench.py rank body over RCCL (world 1): %.2f ms/step, %.2f without the exchange, %d collectives, last step exposed %.3f ms"
          % (out["ms_per_step"], ex["ms_per_step_without_exchange"], ex["collectives_per_step"], ex["exposed_ms_last_step"]))
    assert "roofline" in out and out["roofline"]["frac"] > 0.2                                             # (measured after the process group was destroyed)


def test_nonfinite_step_is_skipped_on_the_device_and_latched():
    """P/pretrain_AntoMask.py:441-446 checks `loss.item()` after every step.  Here the check is on the device (am_adamw_ema `guard`): a NaN
    volume at step k must leave weights, Adam moments, EMA teacher and BatchNorm buffers (running statistics AND num_batches_tracked) of
    student and teacher BIT-identical to what they were before step k, latch the flag with the step index, and keep later (clean) steps
    from changing anything either -- the driver finds the pre-step-k state when it looks once per epoch.  Also the overflow case: a
    loss of +inf with finite reconstructions elsewhere (am_patch_loss_bwd makes its gradient NaN)."""
    from anatomask_amd import modules as M
    from anatomask_amd.trainer import AnatoMaskTrainer
    for poison in (float("nan"), float("inf")):
        torch.manual_seed(0)
        model = M.build_spark([32, 32, 48, 64, 64, 64], [1] * 6, 128, (48, 48, 48), 0.6, compute_dtype=torch.bfloat16).to(DEV)
        tr = AnatoMaskTrainer(model, lr=1e-3, total_epochs=100, seed=3)
        t = tr.teacher.ema
        x = np_volume(2, (48, 48, 48), 77).to(DEV)

        def snap():
            return [v.detach().clone() for v in (model._flat, model._bflat, model._iflat, tr.m, tr.v, t._flat, t._bflat, t._iflat)]
        for _ in range(2):
            o = tr.step(x, epoch=50)
        assert tr.nonfinite_step() is None and tr.guard.tolist()[:3] == [0, 0, 2] and np.isfinite(o["loss"].item())
        assert int(model._iflat[0]) == 2 and int(t._iflat[0]) == 0      # (student counted 2 batches; timm's int64 EMA truncates 0.002 to 0)
        before = snap()
        xb = x.clone()
        xb[1, 0, 5:9, 17:21, 30:34] = poison
        o = tr.step(xb, epoch=50)                                       # step 3: non-finite
        assert not np.isfinite(o["loss"].item())
        after_bad = snap()
        for _ in range(2):
            tr.step(x, epoch=50)                                        # clean steps after the latch: skipped as well
        torch.cuda.synchronize()
        names = ("weights", "BN running stats", "num_batches_tracked", "exp_avg", "exp_avg_sq", "teacher", "teacher BN stats", "teacher counters")
        for n, a, b, c in zip(names, before, after_bad, snap()):
            assert torch.equal(a, b) and torch.equal(a, c), n
        assert all(torch.isfinite(v.float()).all() for v in before)
        assert tr.guard.tolist()[:3] == [1, 3, 5] and tr.nonfinite_step() == 3
        tr.reset_guard()                                               # (what a resume does) -> training continues from the clean state
        o = tr.step(x, epoch=50)
        assert np.isfinite(o["loss"].item()) and tr.nonfinite_step() is None and not torch.equal(model._flat, before[0])


def test_nonfinite_guard_two_ranks_latch_together():
    """a NaN volume on ONE rank: the all-reduced gradient is NaN on both, both latch at the same step and keep their pre-step state."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(_free_port()), os.path.join(root, "tools", "guard_two_ranks_one_gpu.py")],
                       capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert r.stdout.count("first non-finite step 3; state as before step 3: True") == 2, r.stdout[-2000:]


def test_pretrain_driver_exits_minus_one_on_a_nonfinite_step(tmp_path, monkeypatch):
    """the driver loop (anatomask_amd.pretrain): a non-finite volume in the feed -> exit code -1 at the end of that epoch (as
    P/pretrain_AntoMask.py:446), message naming the step, and NO checkpoint of the poisoned epoch."""
    from anatomask_amd import pretrain

    def poisoned(batch, size, seed):
        g, n = torch.Generator().manual_seed(seed), 0
        while True:
            v = torch.randn(batch, 1, *size, generator=g)
            n += 1
            if n == 5:                                                 # epoch 0 = steps 1..3 (saved), epoch 1 = steps 4..6
                v[0, 0, :4, :4, :4] = float("nan")
            yield {"data": v}
    monkeypatch.setattr(pretrain, "synthetic_batches", poisoned)
    out = str(tmp_path / "run")
    with pytest.raises(SystemExit) as e:
        pretrain.main(["--model", "S", "--input-size", "48", "48", "48", "--batch-size", "2", "--epochs", "3", "--iters-per-epoch", "3", "--out", out])
    assert e.value.code == -1
    ck = torch.load(os.path.join(out, "STUNet_S_head_latest.pt"))
    assert int(ck["current_epoch"]) == 0 and all(torch.isfinite(v.float()).all() for v in ck["network_weights"].values())
    log = [f for f in os.listdir(out) if f.startswith("training_log_")]
    assert log and "first non-finite step: 5" in open(os.path.join(out, log[0])).read()


# ------------------------------------------------------------------------------------------------------------------------
# L2 / L3 boundary: the sub-modules called on their own, as the reference's users call them
# ------------------------------------------------------------------------------------------------------------------------
def test_sparse_encoder_forward_matches_reference_fixture(fwd):
    """`model.sparse_encoder(masked)` with the side channel `_cur_active` set first -- exactly how tests/golden/make_fixtures.py
    drives the REFERENCE's SparseEncoder (P/encoder3D.py:5,366-367, P/STUNet_head.py:67-76) -- against the enc{i} fixtures;
    plus autograd through the stand-alone encoder against the CPU oracle."""
    from anatomask_amd import modules as M
    cfg = tiny_cfg(fwd)
    W = fixture_weights(cfg, fwd)
    x = np_volume(int(fwd["B"]), cfg.input_size, fwd["x_seed"])
    mask = torch.from_numpy(fwd["fwd_mask"])
    act = O.upsample_mask(mask, x.shape[2:])
    m = make_model(cfg, W).train()
    M._cur_active = mask.to(DEV)
    masked = (x * act.float()).to(DEV)
    feats = m.sparse_encoder(masked)
    assert isinstance(feats, list) and len(feats) == 5
    for i, f in enumerate(feats):
        assert tuple(f.shape) == (x.shape[0], cfg.dims[i], *(v >> i for v in cfg.input_size)) and f.dtype == torch.float32
        fc = f.detach().cpu()
        assert_checks(fc, fwd[f"enc{i}_checks"], 2e-4, f"enc{i}")
        assert rel_err(sample(fc), fwd[f"enc{i}_sample"]) < 2e-4
        assert float(fc[~O.upsample_mask(mask, fc.shape[2:]).expand_as(fc)].abs().max()) == 0.0      # exact zeros where the reference has zeros
    last = m.sparse_encoder.sp_cnn(masked, hierarchical=False)                # P/STUNet_head.py:75-76
    assert torch.equal(last, feats[-1])
    # gradients of a loss on the five maps, wrt every encoder parameter
    wts = [1.0, 0.5, 2.0, 1.5, 3.0]
    loss = sum(w_ * (f ** 2).mean() for w_, f in zip(wts, feats))
    loss.backward()
    keys = [k for k in O.trainable_keys(cfg) if k.startswith(O.ENC)]
    leaves = {k: W[k].clone().requires_grad_(True) for k in keys}
    p = dict(W); p.update(leaves)
    fo = O.encoder_forward(cfg, p, x * act.float(), mask)
    lo = sum(w_ * (f ** 2).mean() for w_, f in zip(wts, fo))
    lo.backward()
    assert abs(loss.item() - lo.item()) < 2e-4 * abs(lo.item())
    got = dict(m.named_parameters())
    for k in keys:
        a, b = got[k].grad.detach().cpu().double().flatten(), leaves[k].grad.double().flatten()
        if float(b.norm()) < 1e-9 or _zero_grad_bias(k):      # conv biases that feed a norm: analytically zero, float noise
            continue
        cos = float((a * b).sum() / (a.norm() * b.norm()))
        assert cos > 0.9995 and float((a - b).norm() / b.norm()) < 3e-2, (k, cos, float(a.norm()), float(b.norm()))
    assert all(q.grad is None for k, q in got.items() if not k.startswith(O.ENC))      # nothing outside the encoder was touched
    M._cur_active = None


def test_light_decoder_forward_and_grads_vs_oracle(fwd):
    """`model.dense_decoder(to_dec)` (P/decoder3D.py:55-63) on the engine: output, BN buffer update, gradients wrt the decoder
    parameters and wrt every to_dec[i], against the CPU oracle; a 5th list entry is ignored as in the reference."""
    cfg = tiny_cfg(fwd)
    W = fixture_weights(cfg, fwd)
    m = make_model(cfg, W).train()
    B = 2
    rs = np.random.RandomState(3)
    sizes = [tuple(v // 16 * 2 ** i for v in cfg.input_size) for i in range(5)]
    to_dec_cpu = [torch.from_numpy(rs.standard_normal((B, cfg.dec_chs[i] if i < 4 else 8, *sizes[i])).astype(np.float32)) for i in range(5)]
    to_dec = [t.to(DEV).requires_grad_(True) for t in to_dec_cpu]
    rec = m.dense_decoder(to_dec)
    assert tuple(rec.shape) == (B, 1, *cfg.input_size)
    (rec ** 2).mean().backward()
    keys = [k for k in O.trainable_keys(cfg) if k.startswith("dense_decoder")]
    leaves = {k: W[k].clone().requires_grad_(True) for k in keys}
    p = dict(W); p.update(leaves)
    tin = [t.clone().requires_grad_(True) for t in to_dec_cpu[:4]]
    nb = {}
    ro = O.decoder_forward(cfg, p, tin, True, nb)
    (ro ** 2).mean().backward()
    assert rel_err(rec.detach().cpu().numpy(), ro.detach().numpy()) < 2e-4
    for k, v in nb.items():
        np.testing.assert_allclose(m.state_dict()[k].cpu().numpy(), v.numpy(), rtol=1e-4, atol=1e-6)
    got = dict(m.named_parameters())
    for k in keys:
        a, b = got[k].grad.detach().cpu().double().flatten(), leaves[k].grad.double().flatten()
        cos = float((a * b).sum() / (a.norm() * b.norm()))
        assert cos > 0.9995 and float((a - b).norm() / b.norm()) < 3e-2, (k, cos)
    for i in range(4):
        a, b = to_dec[i].grad.cpu().double().flatten(), tin[i].grad.double().flatten()
        assert float((a - b).norm() / b.norm()) < 1e-2, i
    assert to_dec[4].grad is None


def test_light_decoder_instance_norm_matches_the_reference():
    """LightDecoder(sbn=False, use_IN=True) (P/decoder3D.py:44-45: nn.InstanceNorm3d, no affine parameters, per-sample statistics in train
    and eval mode) against the reference's own class evaluated in float64 (tests/golden/decoder_in_tiny.npz, made by
    tests/golden/make_decoder_in_fixture.py): state_dict keys, reconstruction, gradient wrt every input map and every parameter."""
    from anatomask_amd import modules as M
    from tests.helpers import seeded_params, decoder_in_inputs as inputs, DECODER_IN_WIDTH as WIDTH
    F_ = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "decoder_in_tiny.npz"))
    tol = 5e-4
    dec = M.LightDecoder(16, sbn=False, width=WIDTH, use_IN=True, out_channel=1)
    assert sorted(dec.state_dict().keys()) == list(F_["keys"])
    assert M.LightDecoder(16, width=WIDTH, use_IN=True).use_IN is False            # sbn (the default) wins, P/decoder3D.py:42-45
    seeded_params(dec)
    dec = dec.to(DEV)
    maps, probe = inputs()
    want = torch.from_numpy(F_["train.rec"])
    for mode in ("train", "eval"):
        dec.train(mode == "train")
        dec.zero_grad()
        xs = [x.to(DEV).requires_grad_(True) for x in maps]
        rec = dec(xs)
        assert float((rec.detach().cpu().double() - want.double()).norm() / want.double().norm()) <= tol, mode
        (rec * probe.to(DEV)).sum().backward()
        for i, x in enumerate(xs):
            w = torch.from_numpy(F_[f"train.gx{i}"]).double()
            assert float((x.grad.cpu().double() - w).norm() / w.norm()) <= 4 * tol, (mode, i)
        gtot = float(F_["train.gtot"])
        for n, p_ in dec.named_parameters():
            g = p_.grad.reshape(-1).float().cpu()
            idx = np.linspace(0, g.numel() - 1, min(64, g.numel())).astype(np.int64)
            want_n, want_s, floor = float(F_[f"train.gn.{n}"]), F_[f"train.gs.{n}"], float(F_[f"train.floor.{n}"])
            bound = 3 * floor + 4 * tol
            assert abs(float(g.norm()) - want_n) <= bound * max(want_n, 1e-3 * gtot), (mode, n, float(g.norm()), want_n, floor)
            scale = max(float(np.abs(want_s).max()), 1e-3 * gtot / np.sqrt(g.numel()))
            assert float(np.abs(g[torch.from_numpy(idx)].numpy() - want_s).max()) <= 2 * bound * scale + 1e-7, (mode, n, floor)
    # bf16 storage: same function within storage noise
    dec.compute_dtype = torch.bfloat16
    with torch.no_grad():
        rec = dec([x.to(DEV) for x in maps])
    assert float((rec.cpu().double() - want.double()).norm() / want.double().norm()) <= 3e-2


def test_spark_with_instance_norm_decoder_runs_end_to_end(fwd):
    """SparK(dense_decoder=LightDecoder(sbn=False, use_IN=True)) through the module API: forward, loss, backward; the decoder inside the
    model is the same function as the decoder alone (the stand-alone class is pinned to the reference by the test above)."""
    from anatomask_amd import modules as M
    from tests.helpers import seeded_params
    cfg = tiny_cfg(fwd)
    head = M.STUNet(1, 1, depth=cfg.depth, dims=cfg.dims)
    enc = M.SparseEncoder(head, input_size=cfg.input_size, sbn=False)
    dec = M.LightDecoder(enc.downsample_ratio, sbn=False, width=cfg.width, use_IN=True, out_channel=1)
    model = M.SparK(sparse_encoder=enc, dense_decoder=dec, mask_ratio=cfg.mask_ratio, densify_norm="in").train()
    seeded_params(model)
    model = model.to(DEV)
    assert not any("dense_decoder.dec" in k and (".conv.1." in k or ".conv.4." in k) for k in model.state_dict())
    x = np_volume(2, cfg.input_size, 7).to(DEV)
    mask = O.random_mask(cfg, 2, torch.Generator().manual_seed(2)).to(DEV)
    inp, rec = model(x, active_b1ff=mask)
    loss, _ = model.forward_loss(inp, rec, mask)
    loss.backward()
    assert np.isfinite(loss.item())
    for n, p_ in model.named_parameters():
        if n in model._dead:                       # level 4's densify norm / projection / token feed nothing (P/AnatoMask.py:163-171), as in the reference
            continue
        assert p_.grad is not None and torch.isfinite(p_.grad).all(), n
    assert float(model.dense_decoder.dec[0].conv[0].weight.grad.norm()) > 0
    # per-sample statistics: the reconstruction of sample 0 does not depend on what sample 1's decoder input was -- scale invariance of
    # InstanceNorm per sample: scaling ONE sample's maps into the decoder by a constant leaves every sample's output where it was
    maps = [torch.randn(2, c, *(v // 16 * 2 ** i for v in cfg.input_size), device=DEV) for i, c in enumerate(cfg.dec_chs[:4])]
    with torch.no_grad():
        a = model.dense_decoder(maps)
        scaled = [torch.cat([m_[:1], 3.0 * m_[1:]]) for m_ in maps]
        b = model.dense_decoder(scaled)
    assert float((a[0] - b[0]).abs().max()) <= 1e-5 * float(a[0].abs().max())
    with pytest.raises(NotImplementedError):
        from anatomask_amd.trainer import AnatoMaskTrainer
        AnatoMaskTrainer(model)


def test_standalone_encoder_outside_spark_matches(fwd):
    """STUNet + SparseEncoder constructed on their own (no SparK around them): same maps as inside the model."""
    from anatomask_amd import modules as M
    cfg = tiny_cfg(fwd)
    W = fixture_weights(cfg, fwd)
    x = np_volume(2, cfg.input_size, 7)
    mask = O.random_mask(cfg, 2, torch.Generator().manual_seed(2))
    masked = (x * O.upsample_mask(mask, x.shape[2:]).float()).to(DEV)
    m = make_model(cfg, W).eval()
    head = M.STUNet(1, 1, depth=cfg.depth, dims=cfg.dims)
    enc = M.SparseEncoder(head, input_size=cfg.input_size, sbn=False)
    enc.load_state_dict({k[len("sparse_encoder."):]: v for k, v in W.items() if k.startswith("sparse_encoder.")})
    enc = enc.to(DEV)
    M._cur_active = mask.to(DEV)
    with torch.no_grad():
        a, b = m.sparse_encoder(masked), enc(masked)
    M._cur_active = None
    for fa, fb in zip(a, b):
        assert torch.equal(fa, fb)


def test_plain_spark_validation_loss_vs_oracle(fwd):
    """AnatoMaskTrainer.eval_loss: the per-epoch validation pass of the plain-SparK driver (P/pretrain.py:426-441, eval mode: decoder
    BatchNorm on running statistics, no grad) against the CPU oracle."""
    from anatomask_amd.trainer import AnatoMaskTrainer
    cfg = tiny_cfg(fwd)
    W0 = fixture_weights(cfg, fwd)
    x = np_volume(2, cfg.input_size, 61)
    mask = O.random_mask(cfg, 2, torch.Generator().manual_seed(12))
    tr = AnatoMaskTrainer(make_model(cfg, W0), lr=1e-3, total_epochs=1000, distributed=False, self_distill=False)
    got = tr.eval_loss(x.to(DEV), mask).item()
    ip, rp = O.spark_forward(cfg, {k: v.clone() for k, v in W0.items()}, x, mask, train=False)
    want = float(O.forward_loss(ip, rp, mask)[0])
    assert abs(got - want) < 2e-4 * abs(want), (got, want)
    sd = tr.model.state_dict()
    assert all(int(v) == 0 for k, v in sd.items() if k.endswith("num_batches_tracked"))      # eval: buffers untouched
    assert torch.isfinite(tr.eval_loss(x.to(DEV))).all()                                     # random-mask form


def test_syncbn_two_ranks_equal_one_big_batch():
    """LightDecoder(sbn=True) (nn.SyncBatchNorm in the reference, P/decoder3D.py:42-43, P/pretrain_DDP.py:225): two ranks x 2 volumes ==
    one rank x 4 volumes with plain BatchNorm (tools/syncbn_two_ranks.py; gloo over device tensors on the one GPU a box has)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(_free_port()), os.path.join(root, "tools", "syncbn_two_ranks.py")],
                       capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert r.stdout.count("syncbn ok: True") == 2, r.stdout[-2000:]


def test_graphed_step_replays_the_eager_step():
    """AnatoMaskTrainer.graphed_step (the step captured in a hipGraph, AdamW's step count / lr through device memory, the mask draws
    through the graph-registered generator) against the eager step() from the same state, in the bit-reproducible mode (bf16, 32-channel
    stem, deterministic_wgrad): masks, losses, weights, teacher and AdamW moments must be BIT-identical after 2 eager + 3 replayed steps."""
    from anatomask_amd import modules as M, ops
    from anatomask_amd.trainer import AnatoMaskTrainer

    def make():
        torch.manual_seed(3)
        model = M.build_spark([32, 32, 48, 64, 64, 64], [1] * 6, 128, (48, 48, 48), 0.6, compute_dtype=torch.bfloat16).to(DEV)
        return AnatoMaskTrainer(model, lr=1e-3, total_epochs=100, seed=11, deterministic_wgrad=True)
    xs = [np_volume(2, (48, 48, 48), 40 + i).to(DEV) for i in range(5)]
    try:
        a, b = make(), make()
        for i, x in enumerate(xs):
            oa, ob = a.step(x, epoch=50), b.graphed_step(x, epoch=50)
            assert torch.equal(oa["mask"], ob["mask"]) and oa["loss"].item() == ob["loss"].item(), i
            if i >= 2:
                want = ops.adam_dyn_scalars(1e-3, (0.9, 0.999), i + 1, b.teacher.decay)
                assert torch.allclose(b._dyn_dev.cpu(), torch.tensor(want, dtype=torch.float32))
        assert b._graph is not None and a.step_count == b.step_count == 5
        n = a.model._live_end
        assert torch.equal(a.model._flat, b.model._flat) and torch.equal(a.teacher.ema._flat, b.teacher.ema._flat)
        assert torch.equal(a.m[:n], b.m[:n]) and torch.equal(a.v[:n], b.v[:n])
    finally:
        ops.DETERMINISTIC_WGRAD = False


def test_graphed_step_without_host_sync_equals_eager():
    """A training loop synchronises once per epoch, not per step: 6 replays enqueued back to back with NO host synchronisation in
    between (the host runs steps ahead of the GPU) must give the eager loop's weights, moments and teacher bit for bit -- AdamW's
    bias corrections of step N must be the ones step N's kernel reads, whatever the host has written since."""
    from anatomask_amd import modules as M, ops
    from anatomask_amd.trainer import AnatoMaskTrainer

    def make():
        torch.manual_seed(3)
        model = M.build_spark([32, 32, 48, 64, 64, 64], [1] * 6, 128, (48, 48, 48), 0.6, compute_dtype=torch.bfloat16).to(DEV)
        return AnatoMaskTrainer(model, lr=1e-3, total_epochs=100, seed=11, deterministic_wgrad=True)
    xs = [np_volume(2, (48, 48, 48), 60 + i).to(DEV) for i in range(8)]
    try:
        a, b = make(), make()
        for x in xs:
            a.step(x, epoch=50)
        for x in xs[:3]:                                   # 2 eager calls + the capture
            b.graphed_step(x, epoch=50)
        torch.cuda.synchronize()
        busy = torch.empty(1 << 28, device=DEV)
        for _ in range(20):
            busy.add_(1.0)                                 # the GPU is behind: every replay below is enqueued long before it runs
        for x in xs[3:]:
            b.graphed_step(x, epoch=50)                    # no .item(), no synchronize
        torch.cuda.synchronize()
        n = a.model._live_end
        assert a.step_count == b.step_count == 8
        assert torch.equal(a.model._flat, b.model._flat) and torch.equal(a.teacher.ema._flat, b.teacher.ema._flat)
        assert torch.equal(a.m[:n], b.m[:n]) and torch.equal(a.v[:n], b.v[:n])
    finally:
        ops.DETERMINISTIC_WGRAD = False


def test_deterministic_mode_is_bit_reproducible():
    """AnatoMaskTrainer(deterministic_wgrad=True), bf16 storage, a model whose stem has 32 channels (the matrix-core stem kernels; STUNet-B/L/H
    do): two runs from the same seed are BIT-identical after 3 full steps (teacher, sampler, student, backward, clip, AdamW, EMA).  The
    weight-gradient reductions (conv, stem, projection) fold per-workgroup partial sums in a fixed order, the bias-gradient and
    statistics sums are fp64 (arrival order shows at 1e-16, below every fp32 / bf16 rounding that follows).  Without the flag the fp32
    atomics of the split-K reductions make half of all gradient elements differ from run to run."""
    from anatomask_amd import modules as M, ops
    from anatomask_amd.trainer import AnatoMaskTrainer

    def run(det):
        ops.DETERMINISTIC_WGRAD = False
        torch.manual_seed(0)
        model = M.build_spark([32, 32, 48, 64, 64, 64], [1] * 6, 128, (48, 48, 48), 0.6, compute_dtype=torch.bfloat16).to(DEV)
        tr = AnatoMaskTrainer(model, lr=1e-3, total_epochs=100, seed=3, deterministic_wgrad=det)
        x = np_volume(2, (48, 48, 48), 77).to(DEV)
        for _ in range(3):
            o = tr.step(x, epoch=50)
        return model._flat.clone(), tr.teacher.ema._flat.clone(), o["loss"].item()
    try:
        a, b = run(True), run(True)
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and a[2] == b[2]
        c, d = run(False), run(False)
        assert not torch.equal(c[0], d[0])                       # (the default mode really is order-dependent: the flag is what fixes it)
        upd = (a[0] - c[0]).abs().max().item()
        assert upd <= 3 * 3e-3                                    # both modes take the same steps up to Adam's +-lr per step
    finally:
        ops.DETERMINISTIC_WGRAD = False


def test_resume_is_bit_exact_in_deterministic_mode(tmp_path):
    """4 steps straight vs 2 steps -> save_checkpoint -> a FRESH trainer -> load_checkpoint -> 2 steps, in the bit-reproducible mode
    (bf16, 32-channel stem, deterministic_wgrad): weights, EMA teacher, AdamW moments and the sampler's next draws must be identical
    bit for bit -- the checkpoint carries everything the step depends on (faithful resume)."""
    from anatomask_amd import checkpoint, modules as M, ops
    from anatomask_amd.trainer import AnatoMaskTrainer

    def fresh():
        torch.manual_seed(0)
        model = M.build_spark([32, 32, 48, 64, 64, 64], [1] * 6, 128, (48, 48, 48), 0.6, compute_dtype=torch.bfloat16).to(DEV)
        return AnatoMaskTrainer(model, lr=1e-3, total_epochs=100, seed=3, deterministic_wgrad=True)
    xs = [np_volume(2, (48, 48, 48), 60 + i).to(DEV) for i in range(4)]
    try:
        a = fresh()
        for x in xs:
            oa = a.step(x, epoch=50)
        b = fresh()
        for x in xs[:2]:
            b.step(x, epoch=50)
        p = str(tmp_path / "latest.pt")
        checkpoint.save_checkpoint(p, b, [1.0, 1.0], 49)
        c = fresh()
        with torch.no_grad():
            for p_ in c.model.parameters():                      # (whatever the fresh trainer held must not survive the load)
                p_.add_(0.123)
        checkpoint.load_checkpoint(p, c)
        for x in xs[2:]:
            oc = c.step(x, epoch=50)
        assert c.step_count == a.step_count == 4
        assert torch.equal(oa["mask"], oc["mask"]) and oa["loss"].item() == oc["loss"].item()
        n = a.model._live_end
        assert torch.equal(a.model._flat, c.model._flat) and torch.equal(a.teacher.ema._flat, c.teacher.ema._flat)
        assert torch.equal(a.m[:n], c.m[:n]) and torch.equal(a.v[:n], c.v[:n])
        assert torch.equal(a.model._bflat, c.model._bflat)       # BatchNorm running statistics
    finally:
        ops.DETERMINISTIC_WGRAD = False


def test_side_stream_schedule_does_not_change_a_bit():
    """Race check for the two-stream backward: in the bit-reproducible mode the step with the weight gradients on the side HIP stream
    (events: dy ready -> side stream; side stream joined before each parameter group is final and at the end of backward) must produce
    exactly the bits of the fully serialised schedule.  A missing wait shows up as a difference (a wgrad reading a half-written dy, the
    optimizer reading a gradient that is still being accumulated)."""
    from anatomask_amd import engine, modules as M, ops
    from anatomask_amd.trainer import AnatoMaskTrainer

    def run(side):
        engine._USE_SIDE = side
        torch.manual_seed(0)
        model = M.build_spark([32, 32, 48, 64, 64, 64], [1] * 6, 128, (48, 48, 48), 0.6, compute_dtype=torch.bfloat16).to(DEV)
        tr = AnatoMaskTrainer(model, lr=1e-3, total_epochs=100, seed=5, deterministic_wgrad=True)
        for i in range(4):
            o = tr.step(np_volume(2, (48, 48, 48), 90 + i).to(DEV), epoch=50)
        return model._flat.clone(), tr.teacher.ema._flat.clone(), o["loss"].item()
    try:
        a, b, c = run(True), run(False), run(True)
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and a[2] == b[2]
        assert torch.equal(a[0], c[0])
    finally:
        engine._USE_SIDE = True
        ops.DETERMINISTIC_WGRAD = False


def test_recompute_mode_is_bit_identical_in_deterministic_mode():
    """Activation recomputation (P/GC.py policy: encoder stages and decoder blocks re-run in backward) must change memory, not numbers: in
    the bit-reproducible mode the recompute step produces exactly the weights of the plain step (BatchNorm running statistics updated once)."""
    from anatomask_amd import modules as M, ops
    from anatomask_amd.trainer import AnatoMaskTrainer

    def run(recompute):
        torch.manual_seed(0)
        model = M.build_spark([32, 32, 48, 64, 64, 64], [1] * 6, 128, (48, 48, 48), 0.6, compute_dtype=torch.bfloat16, recompute=recompute).to(DEV)
        tr = AnatoMaskTrainer(model, lr=1e-3, total_epochs=100, seed=5, deterministic_wgrad=True)
        for i in range(3):
            tr.step(np_volume(2, (48, 48, 48), 30 + i).to(DEV), epoch=50)
        return model._flat.clone(), model._bflat.clone(), tr.teacher.ema._flat.clone()
    try:
        a, b = run(False), run(True)
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])
    finally:
        ops.DETERMINISTIC_WGRAD = False


def test_two_ranks_same_data_equal_the_single_process_run_bit_for_bit():
    """The gradient exchange (per-group async all-reduce from the backward hook, SUM + the 1/world factor folded into the optimizer
    kernel) adds nothing of its own: two ranks fed the SAME data in the bit-reproducible mode end up with exactly the weights, teacher,
    AdamW moments and BatchNorm buffers of a single-process run (SHA-256 over all of them)."""
    import os
    import re
    import subprocess
    import sys
    from anatomask_amd import ops
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tools"))
    try:
        import ddp_same_data_det
        want = ddp_same_data_det.run(False)
    finally:
        sys.path.pop(0)
        ops.DETERMINISTIC_WGRAD = False
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(_free_port()), os.path.join(root, "tools", "ddp_same_data_det.py")],
                       capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    got = re.findall(r"rank (\d) sha256 ([0-9a-f]{64})", r.stdout)
    assert len(got) == 2 and {h for _, h in got} == {want}, (want, got)
