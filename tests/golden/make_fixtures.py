"""Generate the golden fixtures in this directory by RUNNING THE REFERENCE.

Run in the build container only (needs /root/reference; it does not exist on the
GPU box):   python tests/golden/make_fixtures.py

The reference's model files are imported from where they lie (nothing is copied);
`timm` is absent from the image, so the three trivial helpers the model files
import from it are provided by an in-memory shim.  The driver
script itself cannot be imported (hard-coded CUDA device / paths / missing
packages), so its loop body (P/pretrain_AntoMask.py:418-441) is driven here with
the reference's own modules, torch.optim.AdamW, the reference's get_param_groups
and a ModelEma restated from timm's published behaviour.

Outputs are data only: inputs, masks, expected outputs / checksums.
"""
import copy
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
P = "/root/reference/nnunetv2/training/nnUNetTrainer/variants/pretrain"
sys.path.insert(0, P)
sys.path.insert(0, "/root/reference")

timm = types.ModuleType("timm"); tm = types.ModuleType("timm.models"); tl = types.ModuleType("timm.models.layers")
tl.trunc_normal_ = torch.nn.init.trunc_normal_
tl.DropPath = type("DropPath", (torch.nn.Identity,), {})
tl.to_3tuple = lambda x: (x, x, x)
sys.modules.update({"timm": timm, "timm.models": tm, "timm.models.layers": tl})

import AnatoMask as ref_am          # noqa: E402
import decoder3D as ref_dec         # noqa: E402
import encoder3D as ref_enc         # noqa: E402
import STUNet_head as ref_head      # noqa: E402
from utils.lr_control import get_param_groups  # noqa: E402

from oracle import anatomask_oracle as O  # noqa: E402

torch.set_num_threads(8)
torch.manual_seed(0)


def build_ref(cfg: O.Config):
    pool = [[2, 2, 2]] * 4 + [[1, 1, 1]]
    ks = [[3, 3, 3]] * 6
    head = ref_head.STUNet(1, 1, depth=cfg.depth, dims=cfg.dims, pool_op_kernel_sizes=pool, conv_kernel_sizes=ks,
                           enable_deep_supervision=True)
    enc = ref_enc.SparseEncoder(head, input_size=cfg.input_size, sbn=False)
    dec = ref_dec.LightDecoder(enc.downsample_ratio, sbn=False, width=cfg.width, out_channel=1)
    return ref_am.SparK(sparse_encoder=enc, dense_decoder=dec, mask_ratio=cfg.mask_ratio, densify_norm="in")


def np_volume(B, size, seed):
    """CT-like learnable volume from the legacy MT19937 stream (stable across numpy versions), so the GPU box can
    regenerate the same input without the fixture storing it (only a sample + checksums are stored)."""
    return O.smooth_volume(B, size, int(seed)).numpy()


WEIGHT_SEED = 0        # O.seeded_state(cfg, WEIGHT_SEED): reference-initialiser-scaled RandomState draws (well-conditioned)


def sample(t: torch.Tensor, n=96):
    f = t.detach().reshape(-1)
    idx = torch.linspace(0, f.numel() - 1, min(n, f.numel())).long()
    return f[idx].numpy().copy()


def checks(t: torch.Tensor):
    f = t.detach().double().reshape(-1)
    return np.array([f.sum().item(), f.abs().sum().item(), (f * f).sum().item()], dtype=np.float64)


def ref_sampler_keys(loss_pred, L, len_keep, len_loss, seed):
    """Replays the numpy draws P/AnatoMask.py:110-121 makes under np.random.seed(seed)
    and turns the visible-branch permutation into keys (keys[id] = position)."""
    B = loss_pred.shape[0]
    order = torch.argsort(loss_pred, dim=1)
    keys = np.zeros((B, L), dtype=np.float32)
    easy_len = (L - len_keep) - len_loss
    np.random.seed(seed)
    for i in range(B):
        hard = order[i, L - len_loss:].numpy()
        deleted = np.delete(np.arange(L), hard)
        np.random.shuffle(deleted)
        keys[i, deleted] = np.arange(len(deleted), dtype=np.float32)
        keys[i, hard] = 1e9
        easy = order[i, L - len_loss - easy_len: L - len_loss].numpy()
        deleted2 = np.delete(np.arange(L), easy)
        np.random.shuffle(deleted2)                    # second draw of the reference (easy_mask)
    return keys


class RefEma:
    """timm.utils.ModelEma restated (third-party, see oracle header)."""

    def __init__(self, model, decay):
        self.ema = copy.deepcopy(model).eval()
        self.decay = decay
        for p in self.ema.parameters():
            p.requires_grad_(False)

    def update(self, model):
        with torch.no_grad():
            msd = model.state_dict()
            for k, ema_v in self.ema.state_dict().items():
                ema_v.copy_(ema_v * self.decay + (1. - self.decay) * msd[k].detach())


def main():
    cfg = O.Config([8, 16, 32, 64, 128, 128], [1] * 6, 128, (32, 48, 64), 0.6)
    B = 2
    W0 = O.seeded_state(cfg, WEIGHT_SEED)
    model = build_ref(cfg)
    sd = model.state_dict()
    assert list(sd.keys()) == list(W0.keys()), "key order mismatch"
    for k in sd:
        assert tuple(sd[k].shape) == tuple(W0[k].shape), k
    model.load_state_dict({k: v.clone() for k, v in W0.items()})
    x = torch.from_numpy(np_volume(B, cfg.input_size, 1234))
    out = {"dims": np.array(cfg.dims), "depth": np.array(cfg.depth), "width": np.array(cfg.width),
           "input_size": np.array(cfg.input_size), "mask_ratio": np.array(cfg.mask_ratio), "B": np.array(B),
           "x_seed": np.array(1234), "x_sample": sample(x), "x_checks": checks(x), "weight_seed": np.array(WEIGHT_SEED)}

    # ---------------- F1/F2: one forward (train mode), per-stage taps ----------------
    g = torch.Generator().manual_seed(7)
    mask = model.mask(B, "cpu", generator=g)
    out["fwd_mask"] = mask.numpy()
    model.train()
    ref_enc._cur_active = mask
    masked = x * O.upsample_mask(mask, x.shape[2:]).float()
    feats = model.sparse_encoder(masked)
    for i, f in enumerate(feats):
        out[f"enc{i}_checks"] = checks(f); out[f"enc{i}_sample"] = sample(f)
    bn_before = {k: v.clone() for k, v in model.state_dict().items() if O.is_buffer(k)}
    inp_p, rec_p = model(x, active_b1ff=mask)
    loss, rec_loss = model.forward_loss(inp_p, rec_p, mask)
    out["fwd_rec_checks"] = checks(rec_p); out["fwd_rec_sample"] = sample(rec_p, 256)
    out["fwd_l2"] = rec_loss.detach().numpy(); out["fwd_loss"] = np.array(loss.item())
    out["fwd_teacher_l2"] = (((rec_p - inp_p) ** 2).mean(dim=2) * mask.logical_not().int().view(B, -1)).detach().numpy()
    # ---------------- F3: gradients ----------------
    model.zero_grad()
    loss.backward()
    gn = {}
    for k, p_ in model.named_parameters():
        gn[k] = float(p_.grad.norm()) if p_.grad is not None else -1.0
    out["grad_keys"] = np.array(list(gn.keys()))
    out["grad_norms"] = np.array(list(gn.values()), dtype=np.float64)
    for k in [f"{O.ENC}.0.0.conv1.weight", f"{O.ENC}.0.0.conv3.weight", f"{O.ENC}.0.0.norm1.weight",
              f"{O.ENC}.1.0.conv3.weight", f"{O.ENC}.2.0.norm2.bias", "dense_decoder.proj.weight",
              "mask_tokens.1", f"{O.DEC}.3.conv.4.weight", "densify_norms.2.weight", f"{O.DEC}.3.up_sample.bias"]:
        out["grad::" + k] = dict(model.named_parameters())[k].grad.numpy().copy()
    for k, p_ in model.named_parameters():          # every gradient tensor: <= 2048 evenly spaced elements (cosine / relative L2 per tensor)
        if p_.grad is not None:
            out["gradsample::" + k] = sample(p_.grad, 2048)
    # BN running stats after that single train forward
    for k in [f"{O.DEC}.0.conv.1.running_mean", f"{O.DEC}.3.conv.4.running_var", f"{O.DEC}.3.conv.4.num_batches_tracked"]:
        out["bn1::" + k] = model.state_dict()[k].numpy().copy()
    # eval-mode forward (teacher semantics: BN on running stats), fresh weights
    model.load_state_dict({k: v.clone() for k, v in W0.items()})
    model.eval()
    with torch.no_grad():
        inp_e, rec_e = model(x, active_b1ff=mask)
    out["eval_rec_checks"] = checks(rec_e); out["eval_rec_sample"] = sample(rec_e, 256)
    out["eval_teacher_l2"] = (((rec_e - inp_e) ** 2).mean(dim=2) * mask.logical_not().int().view(B, -1)).numpy()

    # ---------------- F5: generate_mask exactness ----------------
    L, keep = cfg.L, cfg.len_keep
    gl = torch.Generator().manual_seed(11)
    lp = torch.rand(B, L, generator=gl)
    for tag, (ep, tot) in {"a": (500, 999), "b": (998, 999), "c": (0, 999)}.items():
        np.random.seed(100 + ep)
        torch.manual_seed(100 + ep)
        ll = O.len_loss_for(cfg, ep, tot)
        m_ref, _ = model.generate_mask(lp, guide=True, epoch=ep, total_epoch=tot)
        out[f"gm_{tag}_loss"] = lp.numpy(); out[f"gm_{tag}_ep"] = np.array([ep, tot, ll])
        out[f"gm_{tag}_mask"] = m_ref.numpy()
        if ll > 0:
            out[f"gm_{tag}_keys"] = ref_sampler_keys(lp, L, keep, ll, 100 + ep)
        else:                                    # random branch (:99-103): argsort(randn)
            torch.manual_seed(100 + ep)
            noise = torch.randn(B, L)
            out[f"gm_{tag}_keys"] = noise.numpy()

    # ---------------- F6: patchify map ----------------
    ramp = torch.arange(B * np.prod(cfg.input_size), dtype=torch.float32).view(B, 1, *cfg.input_size)
    out["patchify_sample"] = sample(model.patchify(ramp), 512)

    # ---------------- F7: lr schedule ----------------
    from nnunetv2.training.lr_scheduler.LinearWarmupCosine import LinearWarmupCosineAnnealingLR
    dummy = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.AdamW([dummy], lr=1e-4)
    sch = LinearWarmupCosineAnnealingLR(opt, 20, 1000, 1e-6)
    lrs = []
    for e in range(1001):
        lrs.append(opt.param_groups[0]["lr"])
        opt.step(); sch.step()
    out["lr_sched"] = np.array(lrs, dtype=np.float64)
    tot_ep = 1000
    out["ema_decay_sched"] = np.array([0.999 + i / (tot_ep // 4) * (0.9999 - 0.999) if i < tot_ep // 4 else 0.9999
                                       for i in range(tot_ep)])

    # ---------------- F8: pooled InstanceNorm micro-case ----------------
    xin = torch.from_numpy(np.random.RandomState(5).standard_normal((2, 8, 16, 16, 32)).astype(np.float32))
    am = torch.zeros(2, 1, 1, 1, 2, dtype=torch.bool); am[0, 0, 0, 0, 0] = True; am[1, 0, 0, 0, :] = True   # 1 vs 2 active
    ref_enc._cur_active = am
    sin = ref_enc.SparseInstanceNorm(8, 1e-5)
    with torch.no_grad():
        sin.weight.copy_(torch.linspace(0.5, 1.5, 8)); sin.bias.copy_(torch.linspace(-0.2, 0.2, 8))
        yin = sin(xin * O.upsample_mask(am, xin.shape[2:]).float())
    out["in_mask"] = am.numpy(); out["in_checks"] = checks(yin); out["in_sample"] = sample(yin, 256)

    np.savez_compressed(os.path.join(HERE, "forward_tiny.npz"), **out)

    # ---------------- F4: N-step teacher-forced run ----------------
    N, lr, ep, tot = 6, 1e-3, 500, 999
    model = build_ref(cfg)
    model.load_state_dict({k: v.clone() for k, v in W0.items()})
    ema = RefEma(model, decay=0.99)
    groups = get_param_groups(model, nowd_keys={"cls_token", "pos_embed", "mask_token", "gamma"})
    opt = torch.optim.AdamW(params=groups, lr=lr, weight_decay=1e-5, betas=(0.9, 0.999))
    run = {"N": np.array(N), "lr": np.array(lr), "epoch": np.array([ep, tot]), "ema_decay": np.array(0.99),
           "x_seeds": np.arange(N) + 4000}
    gmask = torch.Generator().manual_seed(21)
    losses, gnorms, m1s, ms, kss, recons = [], [], [], [], [], []
    ll = O.len_loss_for(cfg, ep, tot)
    for s in range(N):
        model.train()
        xs = torch.from_numpy(np_volume(B, cfg.input_size, 4000 + s))
        mask1 = model.mask(B, "cpu", generator=gmask)
        with torch.no_grad():
            i1, r1 = ema.ema(xs, active_b1ff=mask1)
            l2 = ((r1 - i1) ** 2).mean(dim=2)
            recon = l2 * mask1.logical_not().int().view(B, -1)
        np.random.seed(9000 + s)
        m, _ = ema.ema.generate_mask(recon, guide=True, epoch=ep, total_epoch=tot)
        keys = ref_sampler_keys(recon, cfg.L, cfg.len_keep, ll, 9000 + s)
        assert torch.equal(O.generate_mask_from_keys(cfg, recon, torch.from_numpy(keys), ll), m)
        ip, rp = model(xs, active_b1ff=m)
        loss, _ = model.forward_loss(ip, rp, m)
        opt.zero_grad()
        loss.backward()
        gnv = torch.nn.utils.clip_grad_norm_(model.parameters(), 12).item()
        opt.step()
        ema.update(model)
        losses.append(loss.item()); gnorms.append(gnv)
        if s == 0:                                   # snapshot after exactly one optimizer step + one EMA update
            for k, v in model.state_dict().items():
                if v.is_floating_point():
                    run["step1delta::" + k] = sample(v - W0[k], 1024)
                    run["step1ema::" + k] = sample(ema.ema.state_dict()[k] - W0[k], 1024)
        m1s.append(mask1.numpy()); ms.append(m.numpy()); kss.append(keys); recons.append(recon.numpy())
        print(f"[ref step {s}] loss {loss.item():.6f} gnorm {gnv:.6f}")
    run.update(losses=np.array(losses), grad_norms=np.array(gnorms), mask1=np.stack(m1s), mask=np.stack(ms),
               keys=np.stack(kss), recon=np.stack(recons), len_loss=np.array(ll))
    fsd, esd = model.state_dict(), ema.ema.state_dict()
    names = list(fsd.keys())
    run["names"] = np.array(names)
    run["final_checks"] = np.stack([checks(fsd[k].float()) for k in names])
    run["ema_checks"] = np.stack([checks(esd[k].float()) for k in names])
    run["delta_norm"] = np.array([float((fsd[k].float() - W0[k].float()).norm()) for k in names])
    for k in names:
        if not fsd[k].is_floating_point():
            run["final::" + k] = fsd[k].numpy().copy()
            run["ema::" + k] = esd[k].numpy().copy()
            continue
        run["finaldelta::" + k] = sample(fsd[k] - W0[k], 1024)
        run["emadelta::" + k] = sample(esd[k] - W0[k], 1024)
    np.savez_compressed(os.path.join(HERE, "train_tiny.npz"), **run)

    # ---------------- F9: does the step learn?  120 reference steps on ONE fixed learnable batch ----------------
    N, lr, ep, tot = 120, 1e-3, 500, 999
    model = build_ref(cfg)
    model.load_state_dict({k: v.clone() for k, v in W0.items()})
    ema = RefEma(model, decay=0.99)
    opt = torch.optim.AdamW(params=get_param_groups(model, nowd_keys={"cls_token", "pos_embed", "mask_token", "gamma"}),
                            lr=lr, weight_decay=1e-5, betas=(0.9, 0.999))
    xs = torch.from_numpy(np_volume(B, cfg.input_size, 77))
    gmask = torch.Generator().manual_seed(31)
    ov = {"N": np.array(N), "lr": np.array(lr), "epoch": np.array([ep, tot]), "ema_decay": np.array(0.99), "x_seed": np.array(77)}
    losses, m1s, kss = [], [], []
    for s in range(N):
        model.train()
        mask1 = model.mask(B, "cpu", generator=gmask)
        with torch.no_grad():
            i1, r1 = ema.ema(xs, active_b1ff=mask1)
            recon = ((r1 - i1) ** 2).mean(dim=2) * mask1.logical_not().int().view(B, -1)
        np.random.seed(12000 + s)
        m, _ = ema.ema.generate_mask(recon, guide=True, epoch=ep, total_epoch=tot)
        keys = ref_sampler_keys(recon, cfg.L, cfg.len_keep, ll, 12000 + s)
        ip, rp = model(xs, active_b1ff=m)
        loss, _ = model.forward_loss(ip, rp, m)
        opt.zero_grad(); loss.backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), 12)
        opt.step(); ema.update(model)
        losses.append(loss.item()); m1s.append(mask1.numpy()); kss.append(keys)
        if s % 10 == 0:
            print(f"[ref overfit {s}] loss {loss.item():.5f}")
    ov.update(losses=np.array(losses), mask1=np.stack(m1s), keys=np.stack(kss))
    np.savez_compressed(os.path.join(HERE, "overfit_tiny.npz"), **ov)
    for f in ("forward_tiny.npz", "train_tiny.npz", "overfit_tiny.npz"):
        print(f, os.path.getsize(os.path.join(HERE, f)) // 1024, "KiB")


if __name__ == "__main__" and len(sys.argv) == 1:
    main()


def finetune_handoff_fixture():
    """F10: the reference's OWN finetune loader (nnunetv2/run/load_pretrained_weights.py:66-106, load_stunet_ssl_weights) applied to a
    checkpoint in the layout anatomask_amd.checkpoint writes ('network_weights' with 'module.'-prefixed SparK keys): which keys of the
    encoder-only STUNet (P/STUNet_head.py) it fills and with what.  Run: python tests/golden/make_fixtures.py finetune"""
    import tempfile
    from nnunetv2.run.load_pretrained_weights import load_stunet_ssl_weights
    cfg = O.Config([8, 16, 32, 64, 128, 128], [1] * 6, 128, (32, 48, 64), 0.6)
    W0 = O.seeded_state(cfg, WEIGHT_SEED)
    net = ref_head.STUNet(1, 1, depth=cfg.depth, dims=cfg.dims, pool_op_kernel_sizes=[[2, 2, 2]] * 4 + [[1, 1, 1]],
                          conv_kernel_sizes=[[3, 3, 3]] * 6, enable_deep_supervision=True)
    before = {k: v.clone() for k, v in net.state_dict().items()}
    with tempfile.TemporaryDirectory() as td:
        f = os.path.join(td, "STUNet_tiny_head_latest.pt")
        torch.save({"network_weights": {"module." + k: v for k, v in W0.items()}, "optimizer_state": {}, "grad_scaler_state": None,
                    "train_loss": [1.0], "current_epoch": 0}, f)
        import contextlib
        import io
        with contextlib.redirect_stdout(io.StringIO()):
            load_stunet_ssl_weights(net, f)
    after = net.state_dict()
    keys = list(after.keys())
    out = {"keys": np.array(keys), "changed": np.array([not torch.equal(before[k], after[k]) for k in keys]),
           "checks": np.stack([checks(after[k].float()) for k in keys]), "weight_seed": np.array(WEIGHT_SEED)}
    np.savez_compressed(os.path.join(HERE, "finetune_tiny.npz"), **out)
    print("finetune_tiny.npz:", len(keys), "keys,", int(out["changed"].sum()), "filled by the reference loader")


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "finetune":
    finetune_handoff_fixture()
