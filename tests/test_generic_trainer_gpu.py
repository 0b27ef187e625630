"""GPU: the fused-optimizer trainer for SparK models around zoo-converted backbones (anatomask_amd/generic_trainer.py) against the unfused
drop-in route -- torch.optim.AdamW + clip_grad_norm_ + ModelEma.update on the same modules, driven exactly as
P/pretrain_AntoMask.py:418-441 drives them -- on the MedNeXt-shaped encoder whose single forward / backward is pinned to the
reference's own SparK by tests/test_layers_gpu.py::test_spark_around_a_mednext_encoder_matches_the_reference."""
import copy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def build(dn="in"):
    from anatomask_amd import modules as M
    from tests.helpers import seeded_params, tiny_mednext
    dense = tiny_mednext()
    dense.get_downsample_ratio = lambda: 16
    dense.get_feature_map_channels = lambda: [8, 16, 32, 64, 128]
    enc = M.SparseEncoder(dense, input_size=(64, 64, 64))
    dec = M.LightDecoder(enc.downsample_ratio, sbn=False, width=128, out_channel=1)
    model = M.SparK(sparse_encoder=enc, dense_decoder=dec, mask_ratio=0.5, densify_norm=dn, compute_dtype=torch.float32).train()
    seeded_params(model)
    return model.to(DEV)


def draws(step, B=2, L=64, keep=32):
    rs = np.random.RandomState(100 + step)
    m1 = torch.zeros(B, L, dtype=torch.bool)
    for b in range(B):
        m1[b, rs.permutation(L)[:keep]] = True
    return (torch.from_numpy(rs.standard_normal((B, 1, 64, 64, 64)).astype(np.float32)).to(DEV), m1.view(B, 1, 4, 4, 4).to(DEV),
            torch.from_numpy(rs.random_sample((B, L)).astype(np.float32)).to(DEV))


@pytest.mark.parametrize("self_distill", [True, False])
def test_generic_trainer_matches_the_unfused_driver_loop(self_distill):
    from anatomask_amd import modules as M
    from anatomask_amd.generic_trainer import GenericTrainer
    lr, decay, steps, ep, tot = 1e-3, 0.99, 3, 150, 200
    # ---- the reference's loop on our modules (P/pretrain_AntoMask.py:418-441; plain SparK: P/pretrain.py, the random mask is the student mask)
    model = build()
    ema = M.ModelEma(model, decay=decay, device=DEV, resume="")
    opt = torch.optim.AdamW(M.get_param_groups(model, nowd_keys={"cls_token", "pos_embed", "mask_token", "gamma"}), lr=lr, weight_decay=1e-5,
                            betas=(0.9, 0.999))
    ref = []
    for s in range(steps):
        x, mask1, keys = draws(s)
        if self_distill:
            with torch.no_grad():
                inp1, rec1 = ema.ema(x, active_b1ff=mask1)
                recon = ((rec1 - inp1) ** 2).mean(dim=2) * mask1.logical_not().int().view(mask1.shape[0], -1)
            mask, _ = ema.ema.generate_mask(recon, guide=True, epoch=ep, total_epoch=tot - 1, keys=keys)
        else:
            mask = mask1
        inpp, recc = model(x, active_b1ff=mask)
        loss, _ = model.forward_loss(inpp, recc, mask)
        opt.zero_grad()
        loss.backward()
        gn = torch.nn.utils.clip_grad_norm_(model.parameters(), 12.0).item()
        opt.step()
        model.weights_changed()
        if self_distill:
            ema.update(model)
        ref.append((loss.item(), gn, mask.clone()))
    # ---- the fused trainer from the same start
    m2 = build()
    tr = GenericTrainer(m2, lr=lr, weight_decay=1e-5, clip=12.0, ema_decay=decay, total_epochs=tot, distributed=False, self_distill=self_distill)
    for s in range(steps):
        x, mask1, keys = draws(s)
        out = tr.step(x, epoch=ep, mask1=mask1, keys=keys)
        assert torch.equal(out["mask"], ref[s][2]), s                               # same hard masks: same teacher, same sampler
        assert abs(out["loss"].item() - ref[s][0]) <= 2e-5 * abs(ref[s][0]), (s, out["loss"].item(), ref[s][0])
        assert abs(out["grad_norm"].item() - ref[s][1]) <= 1e-3 * ref[s][1], (s, out["grad_norm"].item(), ref[s][1])
    # weights after 3 Adam steps of |update| ~ lr: the two routes differ by the arrival order of the weight-gradient atomics only; an
    # element whose gradient is at the noise level can flip the sign of its first updates -- count those, as the STUNet trainer's test does
    flips = tot_el = 0
    for (k, a), (_, b) in zip(model.state_dict().items(), m2.state_dict().items()):
        if a.is_floating_point():
            d = (a - b).abs().flatten()
            flips += int((d > 0.3 * lr).sum()); tot_el += d.numel()
    assert flips / tot_el < 5e-3, flips / tot_el
    # parameters without a gradient: torch.optim.AdamW skipped them (no decay, no state) -- so did the fused pass: the same set, bit-identical
    # to the initial weights on both routes (before round 5 the fused pass decayed them by lr * wd per step)
    none_ref = sorted(n for n, p in model.named_parameters() if p.requires_grad and p.grad is None)
    assert sorted(tr.dead_parameters) == none_ref, (tr.dead_parameters, none_ref)
    sd_init = build().state_dict()
    for n in none_ref:
        assert torch.equal(m2.state_dict()[n], sd_init[n]) and torch.equal(model.state_dict()[n], sd_init[n]), n
    o = 0
    for (n, p) in tr._params:                                                        # ... and no Adam state
        num = (p.numel() + 3) // 4 * 4
        if n in none_ref:
            assert float(tr.m[o:o + num].abs().max()) == 0.0 and float(tr.v[o:o + num].abs().max()) == 0.0, n
        o += num
    if self_distill:
        worst = 0.0
        for (k, a), (_, b) in zip(ema.ema.state_dict().items(), tr.teacher.ema.state_dict().items()):
            if a.is_floating_point():
                worst = max(worst, float((a - b).abs().max()))
        assert worst <= 3 * steps * (1 - decay) * lr + 1e-7, worst                   # EMA of weights that differ by <= 2 lr per step
        # every state_dict entry moved (timm's update covers buffers too)
        sd0 = build().state_dict()
        moved = [k for k, v in tr.teacher.ema.state_dict().items() if v.is_floating_point() and not torch.equal(v.cpu(), sd0[k].cpu())]
        assert any("running_mean" in k for k in moved) and any("mask_tokens" in k for k in moved)


def test_generic_trainer_refuses_stunet_and_resumes_bit_exactly():
    from anatomask_amd import modules as M
    from anatomask_amd.generic_trainer import GenericTrainer
    kw = dict(lr=1e-3, ema_decay=0.99, total_epochs=200, distributed=False, seed=7)
    with pytest.raises(TypeError):
        GenericTrainer(M.build_spark([8, 16, 32, 64, 128, 128], [1] * 6, 128, (32, 32, 32)).to(DEV))
    a = GenericTrainer(build(), **kw)
    x = draws(0)[0]
    a.step(x, epoch=10)
    sd = copy.deepcopy(a.state_dict())
    import os
    import tempfile
    with tempfile.TemporaryDirectory() as td:                                        # ... and through a FILE (restricted unpickler on the way back)
        a.save(os.path.join(td, "generic.pt"), epoch=10, extra={"note": "x"})
        o1 = [a.step(draws(s)[0], epoch=10) for s in (1, 2)]
        b = GenericTrainer(build(), **kw)
        b.load_state_dict(sd)
        c = GenericTrainer(build(), **kw)
        info = c.load(os.path.join(td, "generic.pt"))
    assert info == {"current_epoch": 10, "extra": {"note": "x"}}
    assert c.dead_parameters == a.dead_parameters and c._live_ranges == a._live_ranges and c.step_count == 1
    for t in (b, c):
        o2 = [t.step(draws(s)[0], epoch=10) for s in (1, 2)]
        for p, q in zip(o1, o2):
            assert torch.equal(p["mask"], q["mask"])                                 # the generator state travelled
            assert abs(p["loss"].item() - q["loss"].item()) <= 1e-5 * abs(p["loss"].item())


def test_generic_trainer_two_ranks_on_one_gpu_stay_in_sync():
    """world_size 2 (gloo over device tensors on the one GPU a box has): different init and data per rank; the bucketed exchange
    (several collectives per step) delivers the mean of the local gradients and the ranks' students / teachers stay bit-identical."""
    import os
    import subprocess
    import sys
    from anatomask_amd.launch import free_port
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(free_port()), os.path.join(root, "tools", "generic_two_ranks_one_gpu.py")],
                       capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert r.stdout.count("weights identical across ranks: True; teacher identical: True") == 2, r.stdout[-2000:]
    assert r.stdout.count("mean-of-gradients ok: True") == 2, r.stdout[-2000:]
