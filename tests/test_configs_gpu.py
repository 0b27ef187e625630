"""GPU: the HIP path on the FULL BASELINE.json configurations (not the tiny fixture config).

  configs[1]  STUNet-B AnatoMask, 128^3, mask 0.6, bf16, B=1: one fused step against the CPU oracle's step (fp32) on the same
              weights / volume / random draws, plus the sampler / EMA / dead-parameter properties at full size.
  configs[3]  STUNet-L (depth 2, dims 64.., width 1024), 160^3, mask 0.7 (L=1000, keep=300), bf16: property checks.
  configs[4]  STUNet-H (depth 3, dims 96.., width 1536), 192^3 (L=1728, keep=691), bf16, activation recomputation (P/GC.py),
              B=2 so that single tensors exceed 2 GB (plane-anchored buffer descriptors, 64-bit sample offsets): property checks.
The 8-GPU variants of these configs differ only by the gradient all-reduce (tests/test_ddp_gloo.py, tools/ddp_two_ranks_one_gpu.py).
"""
import numpy as np
import pytest
import torch

from oracle import anatomask_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _build(cfg, W, dtype=torch.bfloat16, recompute=False):
    from anatomask_amd import modules as M
    m = M.build_spark(cfg.dims, cfg.depth, cfg.width, cfg.input_size, cfg.mask_ratio, compute_dtype=dtype, recompute=recompute)
    if W is not None:
        m.load_state_dict({k: v.clone() for k, v in W.items()})
    return m.to(DEV)


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
    cfg = O.Config.stunet_b((128, 128, 128), 0.6)
    assert (cfg.L, cfg.len_keep) == (512, 205)
    W0 = O.seeded_state(cfg, 5)
    assert sum(v.numel() for k, v in W0.items() if not O.is_buffer(k)) == 53_050_177   # 53.05 M parameters
    x = O.smooth_volume(1, cfg.input_size, 9)
    g = torch.Generator().manual_seed(17)
    mask1 = O.random_mask(cfg, 1, g)
    keys = torch.rand(1, cfg.L, generator=g)
    lr, decay = 1e-4, 0.999
    # ---- oracle step (fp32, CPU).  epoch 0 of 1000: len_loss = 0, the student mask is the keys' choice alone, so both sides
    # train on the SAME mask by construction (at epoch 500 a bf16-level difference of the teacher loss may swap a hard patch)
    torch.set_num_threads(min(32, torch.get_num_threads()))
    st = O.StepState(cfg, W0)
    o = O.train_step(st, x, mask1, keys, 0, 999, lr, decay)
    # ---- HIP step, bf16 storage
    m = _build(cfg, W0)
    tr = AnatoMaskTrainer(m, lr=lr, ema_decay=decay, total_epochs=1000, distributed=False)
    out = tr.step(x.to(DEV), epoch=0, mask1=mask1, keys=keys)
    assert torch.equal(out["mask"].view(1, -1).bool().cpu(), o["mask"].view(1, -1))
    rec_h, rec_o = out["recon_loss"].cpu().numpy(), o["recon_loss"].numpy()
    print("STUNet-B 128^3 bf16 vs oracle fp32: loss %.6f / %.6f  grad-norm %.5f / %.5f  teacher-l2 rel err %.2e"
          % (out["loss"].item(), o["loss"], out["grad_norm"].item(), o["grad_norm"], np.abs(rec_h - rec_o).max() / rec_o.max()))
    assert np.abs(rec_h - rec_o).max() < 2e-2 * rec_o.max()                           # teacher pass, bf16 storage
    assert abs(out["loss"].item() - o["loss"]) < 2e-3 * o["loss"]
    rl_h, rl_o = out["rec_loss"].cpu().numpy(), o["rec_loss"].numpy()
    assert np.abs(rl_h - rl_o).max() < 2e-2 * rl_o.max()                              # per-patch student loss
    assert abs(out["grad_norm"].item() / o["grad_norm"] - 1) < 0.1
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
