"""GPU: the HIP path on the FULL BASELINE.json configurations (not the tiny fixture config).

  configs[1]  STUNet-B AnatoMask, 128^3, mask 0.6, bf16, B=1: one fused step against the CPU oracle's step (fp32) on the same
              weights / volume / random draws, plus the sampler / EMA / dead-parameter properties at full size.
  configs[3]  STUNet-L (depth 2, dims 64.., width 1024), 160^3, mask 0.7 (L=1000, keep=300), bf16: property checks.
  configs[4]  STUNet-H (depth 3, dims 96.., width 1536), 192^3 (L=1728, keep=691), bf16, activation recomputation (P/GC.py),
              B=2 so that single tensors exceed 2 GB (plane-anchored buffer descriptors, 64-bit sample offsets): property checks.
The 8-GPU variants of these configs differ only by the gradient all-reduce (tests/test_ddp_gloo.py, tools/ddp_two_ranks_one_gpu.py).
"""
import functools
import re

import numpy as np
import pytest
import torch

from oracle import anatomask_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

# Asserted bounds, each <= 3 x what was measured on MI355X (the tests print the measured values; round 3: B / L / H):
F32_REL = 1.8e-2                        # per gradient tensor, fp32 storage: measured 6.0e-3 (= the reference's own fp32-vs-fp64 distance)
F32_LOSS, F32_GNORM, F32_L2 = 2e-6, 2e-4, 1e-6     # measured < 1e-7, 6.7e-5, 7.3e-8
BF16_LOSS, BF16_GNORM, BF16_L2 = 1e-4, 2e-2, 1.5e-4   # measured 3e-6 / 3e-5 / 1e-5, 2.4e-3 / 7.6e-3 / 5.8e-3, 4.9e-5
# bf16 storage is a different function from fp32 (gate flips accumulate with depth): per gradient tensor the HIP evaluation must be no
# further from fp32 than BF16_VS_EMU_FACTOR x the ideal bf16 emulation is (+ 0.05), and no further from the emulation than
# BF16_PAIR_FACTOR x the emulation is from fp32 (+ 0.1) -- any two bf16 evaluations of this network differ by as much as either
# differs from fp32 (measured medians, HIP-fp32 / emulation-fp32 / HIP-emulation: B 0.29 / 0.28 / 0.32, L 0.47 / 0.44 / 0.49, H 0.60 / 0.57 / 0.65)
# Round 5 (STUNet-B 128^3, per tensor): HIP / emulation = 0.85 - 1.18 for all 92 tensors once the stem's volume stopped being rounded to bf16
# (the stem weight had sat at 1.7 x = ON this bound: profiles/r05_experiments.md section 7); other summation orders move a tensor by 1 - 5 %.
BF16_VS_EMU_FACTOR = 1.45             # round 5, measured per-tensor ratios HIP / emulation (tests print them): rel-L2 0.79 .. 1.19 (B / L / H), .. 1.35 (recipe shape, a 64-element bias); (1 - cos) up to 1.38 / 1.81 (bound: 1.45^2 = 2.1)
BF16_FULL_SIZE_FACTOR = 1.6            # full-size L / H steps against the reduced-patch emulation's worst tensor (another patch size: not the same tensors)
F32_L2_LARGE, BF16_L2_LARGE = 8e-6, 2.4e-2     # eval-forward per-patch loss of the STUNet-L / H shapes: measured 4.9e-7 / 2.7e-6 (fp32), 1.3e-3 / 7.8e-3 (bf16)
F32_REL_LARGE = 5e-2                    # per gradient tensor, fp32 storage, depth 2 / 3: measured 1.2e-2 (L), 1.8e-2 (H) -- the fp32 floor grows with depth
BF16_PAIR_FACTOR = 1.6


def _build(cfg, W, dtype=torch.bfloat16, recompute=False):
    from anatomask_amd import modules as M
    m = M.build_spark(cfg.dims, cfg.depth, cfg.width, cfg.input_size, cfg.mask_ratio, compute_dtype=dtype, recompute=recompute)
    if W is not None:
        m.load_state_dict({k: v.clone() for k, v in W.items()})
    return m.to(DEV)


_ANALYTIC_ZERO = re.compile(r"conv_blocks_context\.\d+\.\d+\.conv[12]\.bias$")     # a conv bias under an InstanceNorm: d loss / d bias == 0


def _per_tensor_errors(model, ref_grads, what):
    """(relative L2 error, cosine) of EVERY live gradient tensor of the HIP model (model._G: raw sums of this step, before clipping)
    against the oracle's autograd gradients; prints the five worst.  Tensors whose gradient is analytically zero are checked to BE
    small (relative to the global norm) instead."""
    gn = float(torch.sqrt(sum((g.double() ** 2).sum() for g in ref_grads.values())))
    rows = []
    for k, w in ref_grads.items():
        g = model._G[k].detach().double().cpu().reshape(-1)
        w = w.double().reshape(-1)
        if _ANALYTIC_ZERO.search(k):
            assert float(g.norm()) <= 2e-3 * gn and float(w.norm()) <= 2e-3 * gn, (k, float(g.norm()), float(w.norm()), gn)
            continue
        nw = float(w.norm())
        rows.append((k, w.numel(), float((g - w).norm()) / nw, float((g * w).sum() / (g.norm() * nw + 1e-300))))
    assert len(rows) >= len(ref_grads) - 2 * 5 * 3
    worst = sorted(rows, key=lambda r: -r[2])[:5]
    print(f"{what}: {len(rows)} gradient tensors, rel-L2 median {np.median([r[2] for r in rows]):.3e} max {worst[0][2]:.3e}; "
          f"min cos {min(r[3] for r in rows):.6f}; worst: " + ", ".join(f"{r[0]}[{r[1]}] {r[2]:.2e}/{r[3]:.5f}" for r in worst))
    return rows


def _emulated_bf16_grads(cfg, W0, x, mask):
    """The oracle's IDEAL bf16-storage evaluation of the same student pass (fp32 arithmetic; every stored C > 1 activation, its
    gradient and the MFMA weight copies rounded to bf16 at the points where the HIP path stores them: oracle.storage)."""
    with O.storage("bf16"):
        _, _, g, _ = O.student_loss_and_grads(cfg, W0, x, mask, train=True)
    return {k: v for k, v in g.items() if v is not None}


def _check_bf16_rows(rows_vs_fp32, emu_grads, model, fp32_grads, what):
    """bf16 storage is a different FUNCTION from the fp32 reference (activation gates of elements near zero flip, and the flips
    accumulate with depth: the ideal emulation itself sits at rel-L2 0.28 / 0.47 / 0.60 from fp32 for STUNet-B / L / H).  What is
    asserted: (1) the HIP gradients are as close to fp32 as the ideal emulation's are, per tensor; (2) the HIP gradients agree with
    the emulation's -- the same function evaluated in another summation order -- far more tightly than either does with fp32."""
    emu_vs_fp32 = {}
    for k, w in fp32_grads.items():
        if _ANALYTIC_ZERO.search(k):
            continue
        a, w = emu_grads[k].double().reshape(-1), w.double().reshape(-1)
        emu_vs_fp32[k] = (float((a - w).norm() / w.norm()), float((a * w).sum() / (a.norm() * w.norm() + 1e-300)))
    print(f"{what}: ideal bf16 emulation vs fp32: rel-L2 median {np.median([v[0] for v in emu_vs_fp32.values()]):.3e} "
          f"max {max(v[0] for v in emu_vs_fp32.values()):.3e}; min cos {min(v[1] for v in emu_vs_fp32.values()):.5f}")
    ratios = [(rel / emu_vs_fp32[k][0], (1 - cos) / (1 - emu_vs_fp32[k][1] + 1e-12), k) for k, n_el, rel, cos in rows_vs_fp32 if n_el >= 64 and emu_vs_fp32[k][0] > 0.02]
    print(f"{what}: HIP / emulation per tensor (distance from fp32, tensors the emulation moves by > 2 %): rel-L2 ratio {min(r[0] for r in ratios):.2f} .. "
          f"{max(r[0] for r in ratios):.2f} (worst: {max(ratios)[2]}), (1 - cos) ratio up to {max(r[1] for r in ratios):.2f}")
    for k, n_el, rel, cos in rows_vs_fp32:
        if n_el >= 64:
            e_rel, e_cos = emu_vs_fp32[k]
            assert rel <= BF16_VS_EMU_FACTOR * e_rel + 0.05 and 1 - cos <= BF16_VS_EMU_FACTOR ** 2 * (1 - e_cos) + 0.01, (k, n_el, rel, cos, e_rel, e_cos)
    rows = _per_tensor_errors(model, emu_grads, f"{what} bf16 storage vs the ideal bf16 emulation")
    for k, n_el, rel, cos in rows:
        if n_el >= 64:
            assert rel <= BF16_PAIR_FACTOR * emu_vs_fp32[k][0] + 0.1, (k, n_el, rel, cos, emu_vs_fp32[k])
    med = lambda r: float(np.median([v[2] for v in r]))
    assert med(rows_vs_fp32) <= 1.15 * float(np.median([v[0] for v in emu_vs_fp32.values()])) + 0.01     # the medians: measured 1.05 / 1.06 / 1.04 x


@functools.lru_cache(maxsize=None)
def _oracle_step_b128():
    """ONE fp32 CPU oracle step of STUNet-B at 128^3 (shared by the bf16- and the fp32-storage test)."""
    cfg = O.Config.stunet_b((128, 128, 128), 0.6)
    W0 = O.seeded_state(cfg, 5)
    x = O.smooth_volume(1, cfg.input_size, 9)
    g = torch.Generator().manual_seed(17)
    mask1 = O.random_mask(cfg, 1, g)
    keys = torch.rand(1, cfg.L, generator=g)
    torch.set_num_threads(min(32, torch.get_num_threads()))
    st = O.StepState(cfg, W0)
    o = O.train_step(st, x, mask1, keys, 0, 999, 1e-4, 0.999, return_grads=True)
    return cfg, W0, x, mask1, keys, o


def _step_properties(tr, cfg, out, B, epoch, total):
    """size-independent invariants of one AnatoMask step."""
    m = tr.model
    L, keep = cfg.L, cfg.len_keep
    mask = out["mask"].view(B, L).bool().cpu()
    assert (mask.sum(1) == keep).all(), mask.sum(1)                                   # exactly len_keep visible per sample
    ll = O.len_loss_for(cfg, epoch, total)
    recon = out["recon_loss"].cpu()
    assert ll > 0 and (recon >= 0).all() and torch.isfinite(recon).all()
    hard = torch.argsort(recon, dim=1)[:, L - ll:]                                    # the len_loss highest teacher-loss patches
    assert not mask.gather(1, hard).any(), "a hard patch is visible"                  # ... are never visible (P/AnatoMask.py:110)
    loss, gn = out["loss"].item(), out["grad_norm"].item()
    assert np.isfinite(loss) and 0.5 < loss < 1.5 and np.isfinite(gn) and gn > 0, (loss, gn)
    n = m._live_end
    assert torch.isfinite(m._flat).all() and torch.isfinite(tr.teacher.ema._flat).all()
    assert float(m._gflat[n:].abs().max()) == 0.0, "dead densify[4] tensors received a gradient"   # grad=None in the reference
    assert float(m._gflat[:n].abs().max()) > 0
    rl = out["rec_loss"].view(B, L).cpu()
    assert (rl[mask] == 0).all() and (rl[~mask] > 0).all()                            # per-patch loss only on masked patches


def test_config2_stunet_b_128_bf16_step_vs_oracle():
    from anatomask_amd.trainer import AnatoMaskTrainer
    # ---- oracle step (fp32, CPU).  epoch 0 of 1000: len_loss = 0, the student mask is the keys' choice alone, so both sides
    # train on the SAME mask by construction (at epoch 500 a bf16-level difference of the teacher loss may swap a hard patch)
    cfg, W0, x, mask1, keys, o = _oracle_step_b128()
    assert (cfg.L, cfg.len_keep) == (512, 205)
    assert sum(v.numel() for k, v in W0.items() if not O.is_buffer(k)) == 53_050_177   # 53.05 M parameters
    lr, decay = 1e-4, 0.999
    # ---- HIP step, bf16 storage
    m = _build(cfg, W0)
    tr = AnatoMaskTrainer(m, lr=lr, ema_decay=decay, total_epochs=1000, distributed=False)
    out = tr.step(x.to(DEV), epoch=0, mask1=mask1, keys=keys)
    assert torch.equal(out["mask"].view(1, -1).bool().cpu(), o["mask"].view(1, -1))
    rec_h, rec_o = out["recon_loss"].cpu().numpy(), o["recon_loss"].numpy()
    print("STUNet-B 128^3 bf16 vs oracle fp32: loss %.6f / %.6f  grad-norm %.5f / %.5f  teacher-l2 rel err %.2e"
          % (out["loss"].item(), o["loss"], out["grad_norm"].item(), o["grad_norm"], np.abs(rec_h - rec_o).max() / rec_o.max()))
    assert np.abs(rec_h - rec_o).max() < BF16_L2 * rec_o.max()                        # teacher pass, bf16 storage
    assert abs(out["loss"].item() - o["loss"]) < BF16_LOSS * o["loss"]
    rl_h, rl_o = out["rec_loss"].cpu().numpy(), o["rec_loss"].numpy()
    assert np.abs(rl_h - rl_o).max() < 2e-2 * rl_o.max()                              # per-patch student loss
    assert abs(out["grad_norm"].item() / o["grad_norm"] - 1) < BF16_GNORM
    # EVERY gradient tensor of the step against the oracle's autograd gradient of the same step (102 live tensors)
    rows = _per_tensor_errors(m, o["grads"], "STUNet-B 128^3 bf16 storage vs fp32 oracle")
    _check_bf16_rows(rows, _emulated_bf16_grads(cfg, W0, x, o["mask"]), m, o["grads"], "STUNet-B 128^3")
    # EMA identity at full size: teacher = decay * W0 + (1 - decay) * student, elementwise (fp32 flat buffers)
    n = m._live_end
    w0 = torch.cat([W0[k].flatten() for k in m._pnames if k not in m._dead]).to(DEV)
    stu = torch.cat([m._W[k].flatten() for k in m._pnames if k not in m._dead])
    tea = torch.cat([tr.teacher.ema._W[k].flatten() for k in m._pnames if k not in m._dead])
    assert (tea - (decay * w0 + (1 - decay) * stu)).abs().max().item() < 1e-6
    # AdamW's first step moves every live element by ~lr (|m/sqrt(v)| = 1): sign agreement with the oracle where its gradient is
    # not noise; here only the magnitude is asserted (bf16 gate flips decorrelate individual signs)
    step = (stu - w0 * (1 - lr * 1e-5)).abs()
    assert 0.5 * lr < step.median().item() < 1.01 * lr and step.max().item() < 1.01 * lr
    # ---- a second step in the hard-mask regime (epoch 500): sampler / dead-gradient / finiteness properties at full size
    out2 = tr.step(x.to(DEV), epoch=500)
    _step_properties(tr, cfg, out2, 1, 500, 999)
    assert n == m._flat.numel() - sum(((m._W[k].numel() + 3) // 4) * 4 for k in m._dead)


@pytest.mark.parametrize("f32_split", [False, True])
def test_config2_stunet_b_128_fp32_storage_step_vs_oracle(f32_split):
    """The same step with fp32 storage: every gradient tensor, loss, grad-norm and the teacher's per-patch loss at reduction-order
    tolerances.  f32_split=False: exact-f32 MFMA (the parity mode).  f32_split=True: products from bf16 hi / lo splits of both operands
    (AM_DT_F32S: 16 significant bits per operand, 4x the matrix rate) -- the SAME per-tensor / gradient-norm bounds, loss within 1e-5."""
    from anatomask_amd import ops
    from anatomask_amd.trainer import AnatoMaskTrainer
    cfg, W0, x, mask1, keys, o = _oracle_step_b128()
    m = _build(cfg, W0, dtype=torch.float32)
    tr = AnatoMaskTrainer(m, lr=1e-4, ema_decay=0.999, total_epochs=1000, distributed=False, f32_split=f32_split)
    out = tr.step(x.to(DEV), epoch=0, mask1=mask1, keys=keys)
    torch.cuda.synchronize()
    loss_tol = 1e-5 if f32_split else F32_LOSS
    l2_tol = 2e-5 if f32_split else F32_L2
    # per gradient tensor: the exact mode sits at the reference's own fp32-vs-fp64 distance (median 4.2e-3, max 6.0e-3, bound 1.8e-2); the
    # split products (2^-17 per operand instead of 2^-24) measure median 9.3e-3, max 1.87e-2 -- on the Cin = 1 stem weight, the END of the
    # backward chain; 91 of the 92 tensors are inside the exact mode's 1.8e-2 -- stated bound for this mode: 2.5e-2, cosine >= 0.9995
    rel_tol = 2.5e-2 if f32_split else F32_REL
    assert torch.equal(out["mask"].view(1, -1).bool().cpu(), o["mask"].view(1, -1))
    rec_h, rec_o = out["recon_loss"].cpu().numpy(), o["recon_loss"].numpy()
    print("STUNet-B 128^3 fp32%s vs oracle fp32: loss %.7f / %.7f  grad-norm %.6f / %.6f  teacher-l2 rel err %.2e"
          % (" (split products)" if f32_split else "", out["loss"].item(), o["loss"], out["grad_norm"].item(), o["grad_norm"], np.abs(rec_h - rec_o).max() / rec_o.max()))
    assert np.abs(rec_h - rec_o).max() < l2_tol * rec_o.max()
    assert abs(out["loss"].item() - o["loss"]) < loss_tol * o["loss"]
    assert abs(out["grad_norm"].item() / o["grad_norm"] - 1) < F32_GNORM
    rows = _per_tensor_errors(m, o["grads"], "STUNet-B 128^3 fp32 storage" + (" (split products)" if f32_split else ""))
    for k, n_el, rel, cos in rows:
        assert rel <= rel_tol and cos >= (0.9995 if f32_split else 1 - F32_REL), (k, n_el, rel, cos)
    if f32_split:
        assert float(np.median([r_[2] for r_ in rows])) <= 1.2e-2 and sum(r_[2] > F32_REL for r_ in rows) <= 3


# ------------------------------------------------------------------ N steps at the headline configuration
# north_star: "reconstruction loss and encoder weights after N steps must match the reference PyTorch CPU path" (P/pretrain_AntoMask.py:418-441).
# N = 3 full AnatoMask steps of STUNet-B at 128^3, B = 1, epoch 500 of 999 (len_loss = 76 of the 307 masked patches are the teacher's hardest:
# the hard-mask branch of generate_mask is INSIDE every compared step), a fresh volume per step, the two random draws teacher-forced,
# against oracle.train_step.  Every compared step runs the persistent LDS-DMA kernels (conv_k3, its ConvT instantiation, wgrad_k3) that
# carry 70 % of a step; steps 2 and 3 run them on UPDATED weights with a teacher that is no longer the student.
NS_N, NS_EPOCH, NS_TOTAL, NS_LR, NS_DECAY = 3, 500, 999, 1e-4, 0.999


def _ns_draws(cfg):
    g = torch.Generator().manual_seed(23)
    return [(O.smooth_volume(1, cfg.input_size, 100 + s), O.random_mask(cfg, 1, g), torch.rand(1, cfg.L, generator=g)) for s in range(NS_N)]


def _ns_run_oracle(cfg, W0, draws, forced_masks=None):
    """N oracle steps; forced_masks: the student masks of another run, handed to the sampler as keys (0 = visible, ties by id)."""
    st = O.StepState(cfg, W0)
    outs, snap1 = [], None
    for s, (x, mask1, keys) in enumerate(draws):
        if forced_masks is not None:
            keys = 1.0 - forced_masks[s].reshape(1, cfg.L).float()
        outs.append(O.train_step(st, x, mask1, keys, NS_EPOCH, NS_TOTAL, NS_LR, NS_DECAY))
        if s == 0:
            snap1 = ({k: v.clone() for k, v in st.student.items()}, {k: v.clone() for k, v in st.teacher.items()})
    return st, outs, snap1


@functools.lru_cache(maxsize=None)
def _oracle_nsteps_b128(shape=(128, 128, 128)):
    cfg = O.Config.stunet_b(shape, 0.6)
    assert O.len_loss_for(cfg, NS_EPOCH, NS_TOTAL) == {512: 76, 392: 58}[cfg.L]
    W0 = O.seeded_state(cfg, 5)
    draws = _ns_draws(cfg)
    torch.set_num_threads(min(32, torch.get_num_threads()))
    st, outs, snap1 = _ns_run_oracle(cfg, W0, draws)
    return cfg, W0, draws, st, outs, snap1


@functools.lru_cache(maxsize=None)
def _oracle_nsteps_b128_bf16_emulation():
    """the same N steps under oracle.storage("bf16") (the IDEAL bf16-storage evaluation), on the fp32 run's student masks."""
    cfg, W0, draws, st, outs, _ = _oracle_nsteps_b128()
    with O.storage("bf16"):
        ste, oute, _ = _ns_run_oracle(cfg, W0, draws, forced_masks=[o["mask"] for o in outs])
    return ste, oute


def _delta_rows(state, W0, ref_state, keys):
    """per tensor: (name, numel, relative L2 error of the update, cosine, the same error relative to the tensor's own norm)."""
    rows = []
    for k in keys:
        dw = (ref_state[k].double() - W0[k].double()).reshape(-1)
        if float(dw.norm()) < 1e-12:
            continue
        dg = (state[k].double().cpu() - W0[k].double()).reshape(-1)
        e = float((dg - dw).norm() / dw.norm())
        rows.append((k, dw.numel(), e, float((dg * dw).sum() / (dg.norm() * dw.norm() + 1e-300)), float(dg.norm()) * e / (float(W0[k].norm()) + 1e-30)))
    return rows


@pytest.mark.parametrize("mode", ["f32", "f32s", "bf16", "f32-recipe"])
def test_config2_n_steps_vs_oracle(mode):
    """mode "f32-recipe": the same three steps on the reference's SHIPPED recipe shape 112 x 112 x 128 (P/pretrain_AntoMask.py:188,209: 392
    patches, 58 of the 235 masked ones hard; the ragged 28^3 / 14^3 / 7^3 grids take the tail branches of the brick kernels), exact fp32."""
    from anatomask_amd.trainer import AnatoMaskTrainer
    from tests.test_oracle_golden import NSTEP_UPDATE_MEDIAN, NSTEP_WEIGHT_MAX, NSTEP_WEIGHT_MEDIAN, STEP1_FLIPPED, STEP1_UPDATE_MEDIAN
    recipe = mode == "f32-recipe"
    cfg, W0, draws, st, outs, snap1 = _oracle_nsteps_b128((112, 112, 128)) if recipe else _oracle_nsteps_b128()
    mode = "f32" if recipe else mode
    n_hard = O.len_loss_for(cfg, NS_EPOCH, NS_TOTAL)
    bf = mode == "bf16"
    m = _build(cfg, W0, dtype=torch.bfloat16 if bf else torch.float32)
    tr = AnatoMaskTrainer(m, lr=NS_LR, ema_decay=NS_DECAY, total_epochs=NS_TOTAL + 1, distributed=False, f32_split=mode == "f32s")
    # (tensors the oracle's first step moved: the dead densify[4] branch has grad = None in the reference and is never stepped)
    trainable = [k for k in O.trainable_keys(cfg) if not _ANALYTIC_ZERO.search(k) and not torch.equal(snap1[0][k], W0[k])]
    assert len(trainable) == 92
    bn_stats = [k for k in W0 if k.endswith(("running_mean", "running_var"))]
    hsnap1 = None
    for s, (x, mask1, keys) in enumerate(draws):
        o = outs[s]
        if bf:      # bf16 storage: the student trains on the fp32 run's mask (a 5e-5 difference of the teacher's loss may swap a hard patch)
            keys = 1.0 - o["mask"].reshape(1, cfg.L).float()
        out = tr.step(x.to(DEV), epoch=NS_EPOCH, mask1=mask1, keys=keys)
        torch.cuda.synchronize()
        same = torch.equal(out["mask"].view(1, -1).bool().cpu(), o["mask"].view(1, -1))
        rec_h, rec_o = out["recon_loss"].cpu().numpy(), o["recon_loss"].numpy()
        l2 = np.abs(rec_h - rec_o).max() / rec_o.max()
        rl_h, rl_o = out["rec_loss"].cpu().numpy(), o["rec_loss"].numpy()
        print(f"STUNet-B {'x'.join(map(str, cfg.input_size))} {mode} step {s + 1}/{NS_N} (epoch {NS_EPOCH}, {n_hard} hard patches): loss {out['loss'].item():.7f} / {o['loss']:.7f}  "
              f"grad-norm {out['grad_norm'].item():.6f} / {o['grad_norm']:.6f}  teacher-l2 rel err {l2:.2e}  student per-patch rel err "
              f"{np.abs(rl_h - rl_o).max() / rl_o.max():.2e}  mask equal {same}")
        assert same, f"sampler mask diverged at step {s + 1}"
        hard_o = np.argsort(rec_o, axis=1)[:, cfg.L - n_hard:]
        assert not out["mask"].view(1, -1).bool().cpu().numpy()[0, hard_o[0]].any()
        rel = abs(out["loss"].item() - o["loss"]) / o["loss"]
        if bf:
            assert l2 < (BF16_L2 if s == 0 else 4 * BF16_L2) and rel < 2e-3 and abs(out["grad_norm"].item() / o["grad_norm"] - 1) < 0.15
            assert np.abs(rl_h - rl_o).max() < 3e-2 * rl_o.max()
        else:
            # (the tiny fixture's bounds, tests/test_e2e_gpu.py::test_trainer_n_steps_fp32_matches_reference: the trajectories decorrelate step by step)
            assert rel < (3e-4 if s < 2 else 1e-3), (s, rel)
            assert abs(out["grad_norm"].item() / o["grad_norm"] - 1) < (1e-3 if s == 0 else 4e-2), s
            assert l2 < (2e-5 if mode == "f32s" else F32_L2) * (1 if s == 0 else 50)
        if s == 0:
            hsnap1 = ({k: v.detach().float().cpu().clone() for k, v in m.state_dict().items()},
                      {k: v.detach().float().cpu().clone() for k, v in tr.teacher.ema.state_dict().items()})
    fsd = {k: v.detach().float().cpu() for k, v in m.state_dict().items()}
    esd = {k: v.detach().float().cpu() for k, v in tr.teacher.ema.state_dict().items()}
    assert all(torch.isfinite(v).all() for v in fsd.values())
    for k in W0:                                           # integer buffers: exact, student and EMA
        if k.endswith("num_batches_tracked"):
            assert int(fsd[k]) == int(st.student[k]) == NS_N and int(esd[k]) == int(st.teacher[k]), k
    # ---- strict after ONE step (pins clip / AdamW / EMA at full size): Adam's first update is lr * sign(g)
    r1, r1e = _delta_rows(hsnap1[0], W0, snap1[0], trainable), _delta_rows(hsnap1[1], W0, snap1[1], trainable)
    med1 = float(np.median([r[2] for r in r1]))
    flipped = lambda hs, os_: sum(int(((hs[k].double() - os_[k].double()).abs() > 0.1 * (os_[k].double() - W0[k].double()).abs().max()).sum()) for k in trainable) / sum(W0[k].numel() for k in trainable)
    f1, f1e = flipped(hsnap1[0], snap1[0]), flipped(hsnap1[1], snap1[1])
    print(f"   first update vs oracle: median rel {med1:.2e}, elements off by > 10 % of the largest update: student {f1:.2e}, EMA {f1e:.2e}")
    if not bf:
        # (full size: far more elements whose gradient is float noise than in the 8..128-channel fixture, and Adam's first update is
        # lr * sign(g) for them too: measured median 1.1e-2, 3.7e-3 of the elements off by > 10 % of the largest update)
        # the split products perturb the gradients at 2^-17 instead of 2^-24: measured median 7.4e-2, 6.0e-3 of the elements.  Bounds <= 3 x measured.
        assert med1 < (0.2 if mode == "f32s" else 30 * STEP1_UPDATE_MEDIAN) and f1 <= 4 * STEP1_FLIPPED and f1e <= 4 * STEP1_FLIPPED
    # ---- after N steps: encoder / decoder weight updates, EMA, BatchNorm running statistics
    rows, rows_e = _delta_rows(fsd, W0, st.student, trainable), _delta_rows(esd, W0, st.teacher, trainable)
    enc = [r for r in rows if r[0].startswith(O.ENC)]
    med, mx = float(np.median([r[2] for r in rows])), max(rows, key=lambda r: r[2])
    wmed, wmx = float(np.median([r[4] for r in rows])), max(r[4] for r in rows)
    print(f"   N-step update error vs oracle ({len(rows)} tensors, {len(enc)} of the encoder): median {med:.3f} (encoder {np.median([r[2] for r in enc]):.3f}) max {mx[2]:.3f} ({mx[0]}); "
          f"weight-level median {wmed:.2e} max {wmx:.2e}; EMA median {np.median([r[2] for r in rows_e]):.3f}; min cos {min(r[3] for r in rows):.3f}")
    bn = _delta_rows(fsd, W0, st.student, bn_stats)
    bn_e = _delta_rows(esd, W0, st.teacher, bn_stats)
    print(f"   BatchNorm running statistics after N steps ({len(bn)} buffers): update error max {max(r[2] for r in bn):.2e}, EMA copy max {max(r[2] for r in bn_e):.2e}")
    if bf:
        ste, oute = _oracle_nsteps_b128_bf16_emulation()
        for s in range(NS_N):
            assert torch.equal(oute[s]["mask"], outs[s]["mask"]), "the emulation's forced mask differs (a hard patch inside the forced visible set)"
        ide = _delta_rows(ste.student, W0, st.student, trainable)
        imed, imx = float(np.median([r[2] for r in ide])), max(r[2] for r in ide)
        iw, iwx = float(np.median([r[4] for r in ide])), max(r[4] for r in ide)
        print(f"   ideal bf16-storage emulation of the same N steps vs fp32 oracle: median {imed:.3f} max {imx:.3f}; weight-level median {iw:.2e} max {iwx:.2e}")
        assert med < 1.3 * imed and mx[2] < 1.5 * imx and wmed < 1.3 * iw and wmx < 1.5 * iwx, (med, imed, mx, imx)
        assert max(r[2] for r in bn) < 5e-2
    else:
        assert med < NSTEP_UPDATE_MEDIAN and wmed < NSTEP_WEIGHT_MEDIAN and wmx < NSTEP_WEIGHT_MAX, (med, wmed, wmx)
        assert float(np.median([r[2] for r in rows_e])) < NSTEP_UPDATE_MEDIAN
        for k, n_el, e, c, _ in rows + rows_e:
            if n_el >= 64:
                assert e < 0.9 and c > 0.5, (k, n_el, e, c)
        assert max(r[2] for r in bn) < 5e-3 and max(r[2] for r in bn_e) < 5e-3


@functools.lru_cache(maxsize=None)
def _oracle_step_recipe():
    """ONE fp32 CPU oracle step of the reference's SHIPPED recipe shape: STUNet-B, input (112, 112, 128) (P/pretrain_AntoMask.py:188,209),
    mask 0.6 (:215): feature map 7 x 7 x 8 = 392 patches, 157 visible; encoder grids 112/56/28/14/7 x ... x 128/64/32/16/8, decoder
    grids 14 / 28 / 56 / 112 wide -- the ragged extents where the brick kernels run their tail / padding branches."""
    cfg = O.Config.stunet_b((112, 112, 128), 0.6)
    W0 = O.seeded_state(cfg, 6)
    x = O.smooth_volume(1, cfg.input_size, 10)
    g = torch.Generator().manual_seed(19)
    mask1 = O.random_mask(cfg, 1, g)
    keys = torch.rand(1, cfg.L, generator=g)
    torch.set_num_threads(min(32, torch.get_num_threads()))
    st = O.StepState(cfg, W0)
    o = O.train_step(st, x, mask1, keys, 0, 999, 1e-4, 0.999, return_grads=True)
    return cfg, W0, x, mask1, keys, o


@pytest.mark.parametrize("mode", ["f32", "f32s", "bf16"])
def test_reference_recipe_112x112x128_step_vs_oracle(mode):
    """A full AnatoMask step on the reference's shipped recipe shape (P/pretrain_AntoMask.py:209,229,239: STUNet-B, 112 x 112 x 128, fp32)
    against the CPU oracle's step on the same weights / volume / random draws -- the SAME assertions and bounds as the 128^3 tests
    (test_config2_*): fp32 storage with exact products, fp32 storage with split-bf16 products, bf16 storage."""
    from anatomask_amd.trainer import AnatoMaskTrainer
    cfg, W0, x, mask1, keys, o = _oracle_step_recipe()
    assert (cfg.L, cfg.len_keep) == (392, 157)
    dtype = torch.bfloat16 if mode == "bf16" else torch.float32
    m = _build(cfg, W0, dtype=dtype)
    tr = AnatoMaskTrainer(m, lr=1e-4, ema_decay=0.999, total_epochs=1000, distributed=False, f32_split=mode == "f32s")
    out = tr.step(x.to(DEV), epoch=0, mask1=mask1, keys=keys)
    torch.cuda.synchronize()
    assert torch.equal(out["mask"].view(1, -1).bool().cpu(), o["mask"].view(1, -1))
    rec_h, rec_o = out["recon_loss"].cpu().numpy(), o["recon_loss"].numpy()
    print("recipe 112x112x128 %s vs oracle fp32: loss %.7f / %.7f  grad-norm %.6f / %.6f  teacher-l2 rel err %.2e"
          % (mode, out["loss"].item(), o["loss"], out["grad_norm"].item(), o["grad_norm"], np.abs(rec_h - rec_o).max() / rec_o.max()))
    what = f"recipe 112x112x128 {mode}"
    if mode == "bf16":
        assert np.abs(rec_h - rec_o).max() < BF16_L2 * rec_o.max()
        assert abs(out["loss"].item() - o["loss"]) < BF16_LOSS * o["loss"]
        assert abs(out["grad_norm"].item() / o["grad_norm"] - 1) < BF16_GNORM
        rows = _per_tensor_errors(m, o["grads"], what + " storage vs fp32 oracle")
        _check_bf16_rows(rows, _emulated_bf16_grads(cfg, W0, x, o["mask"]), m, o["grads"], what)
    else:
        split = mode == "f32s"
        assert np.abs(rec_h - rec_o).max() < (2e-5 if split else F32_L2) * rec_o.max()
        assert abs(out["loss"].item() - o["loss"]) < (1e-5 if split else F32_LOSS) * o["loss"]
        assert abs(out["grad_norm"].item() / o["grad_norm"] - 1) < F32_GNORM
        rows = _per_tensor_errors(m, o["grads"], what + " storage")
        for k, n_el, rel, cos in rows:
            assert rel <= (2.5e-2 if split else F32_REL) and cos >= (0.9995 if split else 1 - F32_REL), (k, n_el, rel, cos)
    # a second step in the hard-mask regime: sampler / dead-gradient / finiteness properties on the 7 x 7 x 8 patch grid
    out2 = tr.step(x.to(DEV), epoch=500)
    _step_properties(tr, cfg, out2, 1, 500, 999)


def _student_step_vs_oracle(size, patch, mask_ratio, recompute, seed):
    """Plain-SparK step (teacher-forced mask, no teacher) of a large-model SHAPE at a reduced patch against the oracle's autograd --
    the kernel paths only STUNet-L/H reach (depth 2/3 identity-shortcut blocks, 64..1536 channels, one-voxel patches at level 4,
    20- / 24- / 12-wide grids, activation recomputation):
      fp32 storage: loss, gradient norm and EVERY gradient tensor at reduction-order tolerance (a wrong tap fails here);
      bf16 storage: loss / gradient norm, every gradient tensor against the ideal bf16 emulation's distances (_check_bf16_rows);
      eval-mode (teacher) forward: per-patch loss (folded BatchNorm epilogues, skipped visible patches)."""
    from anatomask_amd import engine, modules as M, ops
    from anatomask_amd.trainer import AnatoMaskTrainer
    kw = M.STUNET_CONFIGS[size]
    cfg = O.Config(kw["dims"], kw["depth"], kw["width"], (patch,) * 3, mask_ratio)
    W0 = O.seeded_state(cfg, seed)
    x = O.smooth_volume(1, cfg.input_size, seed + 1)
    mask = O.random_mask(cfg, 1, torch.Generator().manual_seed(seed + 2))
    torch.set_num_threads(min(32, torch.get_num_threads()))
    loss_o, rl_o, grads_o, _ = O.student_loss_and_grads(cfg, W0, x, mask, train=True)
    live = {k: g for k, g in grads_o.items() if g is not None}
    gn_o = float(torch.sqrt(sum((g.double() ** 2).sum() for g in live.values())))
    with torch.no_grad():
        inp1, rec1 = O.spark_forward(cfg, W0, x, mask, train=False)
        recon_o = O.teacher_patch_loss(inp1, rec1, mask)
    emu = _emulated_bf16_grads(cfg, W0, x, mask)
    for dtype, name in ((torch.float32, "fp32"), (torch.bfloat16, "bf16")):
        m = _build(cfg, W0, dtype=dtype, recompute=recompute)
        tr = AnatoMaskTrainer(m, lr=1e-4, total_epochs=1000, distributed=False, self_distill=False)
        out = tr.step(x.to(DEV), epoch=0, mask1=mask)
        loss_h, gn_h = out["loss"].item(), out["grad_norm"].item()
        print(f"STUNet-{size} {patch}^3 {name} vs oracle fp32: loss {loss_h:.7f} / {float(loss_o):.7f}  grad-norm {gn_h:.6f} / {gn_o:.6f}")
        rl_h = out["rec_loss"].cpu().numpy()
        rows = _per_tensor_errors(m, live, f"STUNet-{size} {patch}^3 {name} storage vs fp32 oracle")
        if dtype == torch.float32:
            assert abs(loss_h - float(loss_o)) < F32_LOSS * float(loss_o) and abs(gn_h / gn_o - 1) < F32_GNORM
            assert np.abs(rl_h - rl_o.numpy()).max() < F32_L2 * rl_o.numpy().max()
            for k, n_el, rel, cos in rows:
                assert rel <= F32_REL_LARGE and cos >= 1 - F32_REL_LARGE / 10, (k, n_el, rel, cos)
        else:
            assert abs(loss_h - float(loss_o)) < BF16_LOSS * float(loss_o) and abs(gn_h / gn_o - 1) < BF16_GNORM
            assert np.abs(rl_h - rl_o.numpy()).max() < 2e-2 * rl_o.numpy().max()
            _check_bf16_rows(rows, emu, m, live, f"STUNet-{size} {patch}^3")
        # eval-mode forward of the same weights (the teacher's path)
        W1 = _build(cfg, W0, dtype=dtype)                  # (fresh copy: the step above updated `m`)
        W1._ensure_flat()
        mi = ops.MaskInfo(mask.reshape(1, *W1.spec.fmap).to(device=DEV, dtype=torch.uint8).contiguous())
        xs = x[:, 0].to(DEV).contiguous()
        need = ops.MaskInfo((1 - mi.t).contiguous())
        rec = engine.forward(W1.spec, W1._W, W1._pack, xs, mi, train=False, needed_patches=need)
        recon_h, _, _, _ = ops.patch_loss_fwd(xs, rec, mi, normalized=False, want_loss=False)
        err = np.abs(recon_h.cpu().numpy() - recon_o.numpy()).max() / recon_o.numpy().max()
        print(f"STUNet-{size} {patch}^3 {name} eval forward: teacher-l2 rel err {err:.2e}")
        assert err < (F32_L2_LARGE if dtype == torch.float32 else BF16_L2_LARGE)
        del m, tr, W1, rec
        torch.cuda.empty_cache()


def test_config4_shape_stunet_l_80_mask07_step_vs_oracle():
    _student_step_vs_oracle("L", 80, 0.7, False, 31)


def test_config5_shape_stunet_h_96_recompute_step_vs_oracle():
    _student_step_vs_oracle("H", 96, 0.6, True, 41)


def test_config4_stunet_l_160_mask07_properties():
    from anatomask_amd import modules as M
    from anatomask_amd.trainer import AnatoMaskTrainer
    kw = M.STUNET_CONFIGS["L"]
    cfg = O.Config(kw["dims"], kw["depth"], kw["width"], (160, 160, 160), 0.7)
    assert (cfg.L, cfg.len_keep) == (1000, 300)
    torch.manual_seed(0)
    m = _build(cfg, None)
    assert len(m.state_dict()) == 131 + 5 * 8                                          # depth 2: one more block per stage
    tr = AnatoMaskTrainer(m, lr=1e-4, total_epochs=1000, distributed=False, seed=7)
    x = O.smooth_volume(1, cfg.input_size, 21).to(DEV)
    for _ in range(2):
        out = tr.step(x, epoch=500)
    _step_properties(tr, cfg, out, 1, 500, 999)


def test_config5_stunet_h_192_recompute_properties():
    from anatomask_amd import modules as M
    from anatomask_amd.trainer import AnatoMaskTrainer
    kw = M.STUNET_CONFIGS["H"]
    cfg = O.Config(kw["dims"], kw["depth"], kw["width"], (192, 192, 192), 0.6)
    assert (cfg.L, cfg.len_keep) == (1728, 691)
    torch.manual_seed(0)
    m = _build(cfg, None, recompute=True)
    assert len(m.state_dict()) == 131 + 5 * 8 * 2                                      # depth 3
    B = 2
    assert B * 192 ** 3 * 96 * 2 > 2 ** 31                                             # a level-0 bf16 map exceeds 2 GB: 64-bit sample offsets
    tr = AnatoMaskTrainer(m, lr=1e-4, total_epochs=1000, distributed=False, seed=7)
    x = O.smooth_volume(B, cfg.input_size, 23).to(DEV)
    out = tr.step(x, epoch=500)
    _step_properties(tr, cfg, out, B, 500, 999)
    bt = [v for k, v in m.state_dict().items() if k.endswith("num_batches_tracked")]
    assert all(int(v) == 1 for v in bt)                                                # recomputation does not update BN buffers twice


# ------------------------------------------------------------------ full size: bf16 storage against the HIP path's own fp32 storage
# configs[3] / configs[4] were property-checked only at full size (the oracle comparisons stop at 80^3 / 96^3,
# and the 10^3 / 20^3 / 40^3 and 12^3 / 24^3 / 48^3 grids of the full patches reach kernel-selection branches -- gather vs brick, plane
# bricks, slot counts, the persistent conv_k3 kernel -- by different shapes).  The CPU oracle does not fit the test budget at 160^3 /
# 192^3 (STUNet-H: ~30 TFLOP per pass); the fp32-storage mode of the HIP path does, and it IS pinned to the oracle at 6e-3 .. 1.8e-2 per
# gradient tensor on the same model shapes (test_config4/5_shape_*).  Asserted per configuration, ONE step, same weights / volume / mask
# (epoch 0: the mask is the keys' choice, both storages train on the same one):
#   loss (BF16_LOSS), gradient norm (BF16_GNORM), the teacher's per-patch loss (BF16_L2_LARGE),
#   per gradient tensor: the bf16 step is as far from the fp32-storage step as an ideal bf16 evaluation is from fp32 on that model
#   (medians measured against the oracle at the reduced patches: L 0.47, H 0.60) -- median <= 1.25 x that, no tensor beyond 1.6 x its
#   worst tensor + 0.05, and every gradient tensor is finite and non-zero where the fp32 one is.
FULL_MEDIAN = {"L": 0.47, "H": 0.60}        # ideal bf16 emulation vs fp32, median rel-L2 per tensor (profiles/r03_experiments.md)
FULL_WORST = {"L": 0.75, "H": 0.92}         # ... worst tensor


def _bf16_vs_fp32_storage_full_size(size, patch, mask_ratio, recompute, seed):
    from anatomask_amd import modules as M
    from anatomask_amd.trainer import AnatoMaskTrainer
    kw = M.STUNET_CONFIGS[size]
    cfg = O.Config(kw["dims"], kw["depth"], kw["width"], (patch,) * 3, mask_ratio)
    W0 = O.seeded_state(cfg, seed)
    x = O.smooth_volume(1, cfg.input_size, seed + 1).to(DEV)
    g = torch.Generator().manual_seed(seed + 2)
    mask1 = O.random_mask(cfg, 1, g)
    keys = torch.rand(1, cfg.L, generator=g)
    res = {}
    for dtype, name in ((torch.float32, "fp32"), (torch.bfloat16, "bf16")):
        m = _build(cfg, W0, dtype=dtype, recompute=recompute)
        tr = AnatoMaskTrainer(m, lr=1e-4, ema_decay=0.999, total_epochs=1000, distributed=False)
        out = tr.step(x, epoch=0, mask1=mask1, keys=keys)
        torch.cuda.synchronize()
        live = [k for k in m._pnames if k not in m._dead]
        res[name] = {"loss": out["loss"].item(), "gn": out["grad_norm"].item(), "mask": out["mask"].view(1, -1).bool().cpu(),
                     "recon": out["recon_loss"].double().cpu(), "rl": out["rec_loss"].double().cpu(),
                     "grads": {k: m._G[k].detach().double().cpu().reshape(-1) for k in live}}
        del m, tr, out
        torch.cuda.empty_cache()
    a, b = res["fp32"], res["bf16"]
    assert torch.equal(a["mask"], b["mask"])
    l2 = float((a["recon"] - b["recon"]).abs().max() / a["recon"].max())
    print(f"STUNet-{size} {patch}^3 full size, bf16 vs fp32 storage: loss {b['loss']:.6f} / {a['loss']:.6f}  grad-norm {b['gn']:.5f} / {a['gn']:.5f}  teacher-l2 rel err {l2:.2e}")
    assert abs(b["loss"] - a["loss"]) < BF16_LOSS * a["loss"] and abs(b["gn"] / a["gn"] - 1) < BF16_GNORM
    assert l2 < BF16_L2_LARGE
    assert float((a["rl"] - b["rl"]).abs().max()) < 2e-2 * float(a["rl"].max())
    rows = []
    for k, w in a["grads"].items():
        gb = b["grads"][k]
        assert torch.isfinite(gb).all() and torch.isfinite(w).all(), k
        if _ANALYTIC_ZERO.search(k) or w.numel() < 64:
            continue
        nw = float(w.norm())
        assert nw > 0 and float(gb.norm()) > 0, k
        rows.append((k, float((gb - w).norm()) / nw, float((gb * w).sum() / (gb.norm() * nw + 1e-300))))
    med, worst = float(np.median([r[1] for r in rows])), max(rows, key=lambda r: r[1])
    print(f"   {len(rows)} gradient tensors: rel-L2 median {med:.3f} (ideal emulation at the reduced patch {FULL_MEDIAN[size]}), worst {worst[0]} {worst[1]:.3f}, min cos {min(r[2] for r in rows):.4f}")
    assert med <= 1.25 * FULL_MEDIAN[size], med
    assert worst[1] <= BF16_FULL_SIZE_FACTOR * FULL_WORST[size] + 0.05, worst


def test_config4_stunet_l_160_mask07_bf16_vs_fp32_storage_full_size():
    _bf16_vs_fp32_storage_full_size("L", 160, 0.7, False, 131)


def test_config5_stunet_h_192_recompute_bf16_vs_fp32_storage_full_size():
    _bf16_vs_fp32_storage_full_size("H", 192, 0.6, True, 141)


def test_config2_bf16_training_tracks_fp32_storage_training():
    """configs[1] at TRAINING level: 10 optimizer steps of STUNet-B 128^3 (B=2, one repeated batch, plain-SparK mode so that both runs see
    the same random masks) in bf16 storage -- the dense convolutions on conv_k3_kernel -- against the same steps in fp32 storage
    (split-bf16 products): the loss curve must fall, and the bf16 curve must follow the fp32 one step by step.  A per-step bias of the
    bf16 kernels (a wrong tap, a dropped halo row, a statistics epilogue off by a brick) compounds over the steps and shows here even if
    a single step's tensors pass the per-tensor bounds.  Measured (printed): loss 1.013 -> 0.775 (bf16) / 0.773 (fp32), max per-step relative distance 7e-3."""
    from anatomask_amd import modules as M
    from anatomask_amd.trainer import AnatoMaskTrainer
    kw = M.STUNET_CONFIGS["B"]
    steps, lr = 10, 1e-3
    torch.manual_seed(3)
    base = M.build_spark(kw["dims"], kw["depth"], kw["width"], (128,) * 3, 0.6)
    W0 = {k: v.clone() for k, v in base.state_dict().items()}
    g = torch.Generator().manual_seed(17)
    # a learnable volume: smooth structure + noise (pure noise has nothing to reconstruct and the curve would not move)
    z = torch.randn(2, 1, 16, 16, 16, generator=g)
    x = (torch.nn.functional.interpolate(z, size=(128,) * 3, mode="trilinear", align_corners=False) + 0.1 * torch.randn(2, 1, 128, 128, 128, generator=g)).to(DEV)
    curves = {}
    for name, dtype, split in (("bf16", torch.bfloat16, False), ("fp32", torch.float32, True)):
        m = M.build_spark(kw["dims"], kw["depth"], kw["width"], (128,) * 3, 0.6, compute_dtype=dtype)
        m.load_state_dict({k: v.clone() for k, v in W0.items()})
        tr = AnatoMaskTrainer(m.to(DEV), lr=lr, total_epochs=1000, seed=99, distributed=False, self_distill=False, f32_split=split)
        curves[name] = [tr.step(x, epoch=0)["loss"].item() for _ in range(steps)]
        del tr, m
        torch.cuda.empty_cache()
    from anatomask_amd import ops
    b, f = np.array(curves["bf16"]), np.array(curves["fp32"])
    print("bf16 curve", np.round(b, 5).tolist()); print("fp32 curve", np.round(f, 5).tolist())
    print("per-step relative distance", np.round(np.abs(b - f) / f, 5).tolist())
    assert f[-1] < 0.9 * f[0], "the fp32-storage run must learn something on this batch"
    assert np.all(np.abs(b - f) <= 1e-2 * f), (b, f)
    assert abs((b[0] - b[-1]) - (f[0] - f[-1])) <= 0.1 * (f[0] - f[-1])             # the same amount of progress
