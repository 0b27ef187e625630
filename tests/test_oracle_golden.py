"""CPU: the oracle (oracle/anatomask_oracle.py) against fixtures produced by running the
reference itself (tests/golden/make_fixtures.py).  Tolerances: fp32 CPU restatement vs
reference <= 1e-5 relative on activations/loss, <= 1e-4 on weights after N steps."""
import numpy as np
import pytest
import torch

from oracle import anatomask_oracle as O
from tests.helpers import assert_checks, fixture_weights, grad_errors, load, np_volume, rel_err, sample, tiny_cfg

torch.set_num_threads(8)

# End-to-end gradient tolerances (per parameter tensor, relative L2 over <= 2048 sampled elements, against the REFERENCE's
# gradient in tests/golden/forward_tiny.npz).  Floors measured with tests/calibrate_tolerances.py on these fixtures
# (reference-initialiser weights, CT-like input):
#   * the oracle runs the same torch-CPU fp32 ops as the reference:            median 8e-5, max 2e-4  -> ORACLE bound 1e-3;
#   * the reference's own fp32 gradient vs an fp64 evaluation of the same graph: median 2.6e-3, max 6.1e-3, min cos 0.99998.
#     That second number is what ANY other fp32 implementation (different summation order) can achieve: a 1e-6 activation
#     difference flips a few LeakyReLU(0.01) / ReLU6 gates and every flip is an O(1) change of that element's gradient (with
#     smooth activations the same comparison gives 2.5e-6) -> HIP-fp32 bound: max 1.5e-2, median 6e-3, cos >= 0.9998.
#   * an ideal bf16-STORAGE evaluation of the graph (oracle.storage("bf16")) vs fp64: median 0.27, worst-tensor cos 0.75:
#     the HIP-bf16 bound is stated relative to that emulation (tests/test_e2e_gpu.py).
ORACLE_GRAD_RTOL = 1e-3
GRAD_RTOL = 1.5e-2          # HIP fp32, per tensor (max);  GRAD_RTOL_MEDIAN over tensors below
GRAD_RTOL_MEDIAN = 6e-3
GRAD_COS_MIN = 0.9998


@pytest.fixture(scope="module")
def fwd():
    return load("forward_tiny.npz")


@pytest.fixture(scope="module")
def setup(fwd):
    cfg = tiny_cfg(fwd)
    W = fixture_weights(cfg, fwd)
    x = np_volume(int(fwd["B"]), cfg.input_size, fwd["x_seed"])
    np.testing.assert_allclose(sample(x), fwd["x_sample"], rtol=0, atol=0)
    return cfg, W, x, torch.from_numpy(fwd["fwd_mask"])


def test_param_inventory_matches_reference(setup):
    cfg, W, _, _ = setup
    assert len(W) == 131                      # 131 state_dict entries for depth=[1]*6
    assert len(O.trainable_keys(cfg)) == 107


def test_encoder_stages(fwd, setup):
    cfg, W, x, mask = setup
    masked = x * O.upsample_mask(mask, x.shape[2:]).float()
    feats = O.encoder_forward(cfg, W, masked, mask)
    for i, f in enumerate(feats):
        assert_checks(f, fwd[f"enc{i}_checks"], 1e-5, f"enc{i}")
        assert rel_err(sample(f), fwd[f"enc{i}_sample"]) < 1e-5


def test_forward_loss_and_grads(fwd, setup):
    cfg, W, x, mask = setup
    loss, rec_loss, grads, newbuf = O.student_loss_and_grads(cfg, W, x, mask, train=True)
    assert abs(float(loss) - float(fwd["fwd_loss"])) < 1e-5 * abs(float(fwd["fwd_loss"]))
    assert rel_err(rec_loss.numpy(), fwd["fwd_l2"]) < 1e-5
    for k, gn in zip(fwd["grad_keys"], fwd["grad_norms"]):
        g = grads[str(k)]
        if gn < 0:
            assert g is None, k              # the 5 dead densify[4] tensors
            continue
        # conv biases that feed a norm have analytically zero gradient: float noise only
        if gn < 1e-6:
            assert float(g.norm()) < 1e-5, k
            continue
        assert abs(float(g.norm()) - gn) < ORACLE_GRAD_RTOL * gn, (k, float(g.norm()), gn)
    for k in fwd:
        if k.startswith("grad::"):
            name = k[6:]
            scale = np.abs(fwd[k]).max()
            if scale > 1e-6:
                assert np.abs(grads[name].numpy() - fwd[k]).max() < ORACLE_GRAD_RTOL * scale, name
    errs = grad_errors(grads, fwd)
    assert len(errs) == 102                                     # every live parameter tensor (107 - 5 dead)
    for name, (e, c) in errs.items():
        if c is not None:
            assert e < ORACLE_GRAD_RTOL and c > 1 - 1e-6, (name, e, c)
    for k in fwd:
        if k.startswith("bn1::"):
            np.testing.assert_allclose(newbuf[k[5:]].numpy(), fwd[k], rtol=1e-5, atol=1e-6)


def test_rec_and_teacher_loss(fwd, setup):
    cfg, W, x, mask = setup
    with torch.no_grad():
        ip, rp = O.spark_forward(cfg, W, x, mask, train=True, new_buffers={})
        assert_checks(rp, fwd["fwd_rec_checks"], 1e-5, "rec")
        assert rel_err(sample(rp, 256), fwd["fwd_rec_sample"]) < 1e-4   # fp32 floor after ~25 layers is ~2e-5
        assert rel_err(O.teacher_patch_loss(ip, rp, mask).numpy(), fwd["fwd_teacher_l2"]) < 1e-5
        ie, re_ = O.spark_forward(cfg, W, x, mask, train=False)
        assert_checks(re_, fwd["eval_rec_checks"], 1e-5, "eval rec")
        assert rel_err(O.teacher_patch_loss(ie, re_, mask).numpy(), fwd["eval_teacher_l2"]) < 1e-5


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_generate_mask_exact(fwd, setup, tag):
    cfg = setup[0]
    ep, tot, ll = (int(v) for v in fwd[f"gm_{tag}_ep"])
    assert O.len_loss_for(cfg, ep, tot) == ll
    m = O.generate_mask_from_keys(cfg, torch.from_numpy(fwd[f"gm_{tag}_loss"]), torch.from_numpy(fwd[f"gm_{tag}_keys"]), ll)
    assert np.array_equal(m.numpy(), fwd[f"gm_{tag}_mask"])           # bit-exact
    assert (m.view(m.shape[0], -1).sum(1) == cfg.len_keep).all()


def test_patchify_and_roundtrip(fwd, setup):
    cfg, _, x, _ = setup
    B = x.shape[0]
    ramp = torch.arange(B * int(np.prod(cfg.input_size)), dtype=torch.float32).view(B, 1, *cfg.input_size)
    np.testing.assert_array_equal(sample(O.patchify(cfg, ramp), 512), fwd["patchify_sample"])
    assert torch.equal(O.unpatchify(cfg, O.patchify(cfg, x)), x)


def test_lr_and_ema_schedules(fwd):
    np.testing.assert_allclose(O.lr_schedule(1000), fwd["lr_sched"], rtol=1e-12, atol=1e-18)
    np.testing.assert_allclose([O.ema_decay_for_epoch(i, 1000) for i in range(1000)], fwd["ema_decay_sched"], rtol=1e-15)


def test_pooled_instance_norm_microcase(fwd):
    xin = torch.from_numpy(np.random.RandomState(5).standard_normal((2, 8, 16, 16, 32)).astype(np.float32))
    am = torch.from_numpy(fwd["in_mask"])
    y = O.sparse_instance_norm(xin * O.upsample_mask(am, xin.shape[2:]).float(), torch.linspace(0.5, 1.5, 8),
                               torch.linspace(-0.2, 0.2, 8), 1e-5, am)
    assert_checks(y, fwd["in_checks"], 1e-5, "pooled IN")
    assert rel_err(sample(y, 256), fwd["in_sample"]) < 1e-5


def _zero_grad_bias(k):
    # conv biases feeding a norm have analytically zero gradient; Adam turns their float-noise
    # gradient into O(lr) moves, so they are not comparable across implementations
    return k.startswith(O.ENC) and k.endswith(("conv1.bias", "conv2.bias"))


def delta_metrics(got_delta, want_sample, n=1024):
    g = sample(got_delta, n).astype(np.float64); w = want_sample.astype(np.float64)
    nw = np.linalg.norm(w) + 1e-30
    return np.linalg.norm(g - w) / nw, float((g * w).sum() / (np.linalg.norm(g) * nw + 1e-30))


def mismatch_fraction(got_delta, want_sample, n=1024):
    """Adam's first step is ~lr*sign(g): an element whose gradient is float noise may flip sign.
    Robust metric: fraction of sampled elements off by more than 10% of the largest update."""
    g = sample(got_delta, n).astype(np.float64); w = want_sample.astype(np.float64)
    return float((np.abs(g - w) > 0.1 * np.abs(w).max()).mean())


# N-step bounds (shared with the HIP tests).  The step is CHAOTIC at this horizon in the reference itself: the same reference code
# evaluated in fp64 instead of fp32 gives, after 6 steps at lr 1e-3, a per-tensor update (weights - initial weights) that differs
# by 0.23 (median over tensors, relative L2; max 0.40), although loss and grad-norm agree to 1.4e-4 / 1.6e-2 at every step and
# the first step's update agrees to 3e-5 (median) with 2e-4 of the sampled elements sign-flipped.  Mechanism: Adam's first
# updates are lr*sign(g) per element, elements whose gradient is float noise flip, and tiny-sample norms (the coarsest maps hold
# a few dozen voxels) amplify.  So: strict after ONE step, statistical after N.
STEP1_UPDATE_MEDIAN = 1e-3       # median over tensors of the relative L2 error of the first update (measured 3e-5 .. 5e-5)
STEP1_FLIPPED = 3e-3             # fraction of sampled elements whose first update is off by > 10 % of the largest update (measured 2e-4 .. 7e-4)
NSTEP_UPDATE_MEDIAN = 0.45       # median over tensors, relative L2 error of the N-step update (reference fp32-vs-fp64: 0.23 .. 0.27)
NSTEP_WEIGHT_MEDIAN, NSTEP_WEIGHT_MAX = 4e-2, 0.6   # the same error relative to each tensor's own norm (measured 1.7e-2 / 0.34)


def test_n_step_run_matches_reference():
    """Teacher-forced N-step AnatoMask run vs the reference (P/pretrain_AntoMask.py:418-441): sampler masks bit-exact at every
    step, loss 2e-4, grad-norm 5e-2; strict after ONE step (pins clip/AdamW/EMA arithmetic); statistical after N=6."""
    r = load("train_tiny.npz")
    f = load("forward_tiny.npz")
    cfg = tiny_cfg(f)
    W0 = fixture_weights(cfg, f)
    st = O.StepState(cfg, W0)
    N, lr = int(r["N"]), float(r["lr"])
    ep, tot = (int(v) for v in r["epoch"])
    names = [str(n) for n in r["names"]]
    for s in range(N):
        x = np_volume(int(f["B"]), cfg.input_size, r["x_seeds"][s])
        o = O.train_step(st, x, torch.from_numpy(r["mask1"][s]), torch.from_numpy(r["keys"][s]), ep, tot, lr,
                         float(r["ema_decay"]))
        assert np.array_equal(o["mask"].numpy(), r["mask"][s]), f"sampler mask diverged at step {s}"
        assert abs(o["loss"] - r["losses"][s]) < 2e-4 * abs(r["losses"][s]), (s, o["loss"], r["losses"][s])
        assert abs(o["grad_norm"] - r["grad_norms"][s]) < 5e-2 * r["grad_norms"][s], (s, o["grad_norm"])
        if s == 0:
            errs, mfs, mfe = [], [], []
            for k in names:
                if "step1delta::" + k not in r or _zero_grad_bias(k):
                    continue
                if np.linalg.norm(r["step1delta::" + k]) < 1e-9:
                    continue
                e, c = delta_metrics(st.student[k] - W0[k], r["step1delta::" + k])
                e2, _ = delta_metrics(st.teacher[k] - W0[k], r["step1ema::" + k])
                errs.append(e)
                n_el = min(1024, W0[k].numel())
                mfs.append((mismatch_fraction(st.student[k] - W0[k], r["step1delta::" + k]) * n_el, n_el))
                mfe.append((mismatch_fraction(st.teacher[k] - W0[k], r["step1ema::" + k]) * n_el, n_el))
                assert c > 0.7, ("step1", k, e, c)
            assert np.median(errs) < STEP1_UPDATE_MEDIAN, np.median(errs)
            for m in (mfs, mfe):                      # sign-flipped elements over the whole model
                assert sum(a for a, _ in m) / sum(b for _, b in m) <= STEP1_FLIPPED
    errs, werrs = [], []
    for k in names:
        if "final::" + k in r:                       # integer buffers
            assert np.array_equal(st.student[k].numpy(), r["final::" + k]), k
            assert np.array_equal(st.teacher[k].numpy(), r["ema::" + k]), k
            continue
        if _zero_grad_bias(k) or np.linalg.norm(r["finaldelta::" + k]) < 1e-9:
            continue
        e, c = delta_metrics(st.student[k] - W0[k], r["finaldelta::" + k])
        e2, c2 = delta_metrics(st.teacher[k] - W0[k], r["emadelta::" + k])
        errs.append(e)
        werrs.append(float((st.student[k] - W0[k]).norm()) * e / (float(W0[k].norm()) + 1e-30))
        if W0[k].numel() >= 64:
            assert e < 0.9 and c > 0.5, ("final", k, e, c)
            assert e2 < 0.9 and c2 > 0.5, ("ema", k, e2, c2)
    print("update L2 error after N steps: median %.3f max %.3f; weight-level error: median %.2e max %.2e"
          % (np.median(errs), max(errs), np.median(werrs), max(werrs)))
    assert np.median(errs) < NSTEP_UPDATE_MEDIAN
    assert np.median(werrs) < NSTEP_WEIGHT_MEDIAN and max(werrs) < NSTEP_WEIGHT_MAX   # relative to each tensor's own norm


def test_overfit_curve_first_steps_match_reference():
    """tests/golden/overfit_tiny.npz (120 reference steps on one fixed CT-like batch, loss 1.005 -> 0.044): the oracle follows
    the reference's curve (first 25 steps here; the GPU tests run all 120): to 1e-3 over the first 5 steps, then -- the hard-patch
    set of the sampler flips with 1e-7 differences of the teacher's loss and the trajectories decorrelate -- to 6 %."""
    ov, f = load("overfit_tiny.npz"), load("forward_tiny.npz")
    cfg = tiny_cfg(f)
    st = O.StepState(cfg, fixture_weights(cfg, f))
    ep, tot = (int(v) for v in ov["epoch"])
    x = np_volume(int(f["B"]), cfg.input_size, ov["x_seed"])
    assert ov["losses"][-10:].mean() < 0.06 and ov["losses"][0] > 1.0          # the reference itself learns on this batch
    for s in range(25):
        o = O.train_step(st, x, torch.from_numpy(ov["mask1"][s]), torch.from_numpy(ov["keys"][s]), ep, tot, float(ov["lr"]),
                         float(ov["ema_decay"]))
        assert abs(o["loss"] / ov["losses"][s] - 1) < (1e-3 if s < 5 else 6e-2), (s, o["loss"], ov["losses"][s])


def test_bf16_storage_emulation_is_identity_in_fp32_mode_and_close_in_bf16(fwd, setup):
    """oracle.storage("bf16") (tolerance derivation for the bf16 HIP tests): off by default (the pinned fp32 results above are
    unaffected), and under it the loss moves by < 1e-3 while gradients degrade to the documented bf16-storage floor."""
    cfg, W, x, mask = setup
    l0, _, g0, _ = O.student_loss_and_grads(cfg, W, x, mask)
    with O.storage("bf16"):
        l1, _, g1, _ = O.student_loss_and_grads(cfg, W, x, mask)
    l2, _, g2, _ = O.student_loss_and_grads(cfg, W, x, mask)
    assert float(l0) == float(l2) and all(torch.equal(g0[k], g2[k]) for k in g0 if g0[k] is not None)
    assert abs(float(l1) - float(l0)) < 1e-3 * float(l0)
    e = grad_errors({k: v for k, v in g1.items() if v is not None}, fwd)
    med = np.median([v[0] for v in e.values() if v[1] is not None])
    assert 0.02 < med < 0.6, med


def test_finetune_handoff_matches_reference_loader():
    """tests/golden/finetune_tiny.npz: what the reference's load_stunet_ssl_weights (nnunetv2/run/load_pretrained_weights.py:66-106)
    put into the encoder-only STUNet from a checkpoint in our layout.  encoder_weights_for_finetuning must hand over exactly those
    50 tensors under exactly those keys."""
    from anatomask_amd.checkpoint import encoder_weights_for_finetuning
    ft, f = load("finetune_tiny.npz"), load("forward_tiny.npz")
    cfg = tiny_cfg(f)
    W0 = O.seeded_state(cfg, int(ft["weight_seed"]))
    enc = encoder_weights_for_finetuning({"module." + k: v for k, v in W0.items()})
    keys = [str(k) for k in ft["keys"]]
    assert sorted(enc) == sorted(keys) and len(keys) == 50 and bool(ft["changed"].all())
    for k, want in zip(keys, ft["checks"]):
        assert_checks(enc[k], want, 0.0, k)
