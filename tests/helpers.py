"""Shared helpers for the parity tests (fixture loading, sampling identical to make_fixtures.py)."""
import os

import numpy as np
import torch

from oracle import anatomask_oracle as O

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return dict(np.load(os.path.join(GOLD, name), allow_pickle=False))


def tiny_cfg(f):
    return O.Config(list(f["dims"]), list(f["depth"]), int(f["width"]), tuple(int(v) for v in f["input_size"]),
                    float(f["mask_ratio"]))


def np_volume(B, size, seed):
    """the fixtures' input generator (CT-like learnable volume, legacy MT19937 stream): same call as make_fixtures.py"""
    return O.smooth_volume(B, size, int(seed))


def fixture_weights(cfg, f):
    """the fixtures' weights: reference-initialiser-scaled RandomState draws (oracle.seeded_state), seed stored in the fixture"""
    return O.seeded_state(cfg, int(f["weight_seed"]))


def grad_errors(grads, f, n=2048):
    """per-tensor (relative L2 error, cosine) of `grads` against the reference's gradient samples in fixture `f`
    (`gradsample::<name>`: <= n evenly spaced elements of every gradient tensor).  Tensors whose reference gradient is
    analytically zero (norm < 1e-7: conv biases that feed a norm) are reported with cos = None."""
    out = {}
    for k in f:
        if not k.startswith("gradsample::"):
            continue
        name = k[12:]
        w = f[k].astype(np.float64)
        g = sample(grads[name], n).astype(np.float64)
        nw = np.linalg.norm(w)
        if nw < 1e-7:
            out[name] = (float(np.linalg.norm(g)), None)
            continue
        out[name] = (float(np.linalg.norm(g - w) / nw), float((g * w).sum() / (np.linalg.norm(g) * nw + 1e-300)))
    return out


def sample(t, n=96):
    f = t.detach().float().cpu().reshape(-1)
    idx = torch.linspace(0, f.numel() - 1, min(n, f.numel())).long()
    return f[idx].numpy().copy()


def checks(t):
    f = t.detach().double().cpu().reshape(-1)
    return np.array([f.sum().item(), f.abs().sum().item(), (f * f).sum().item()], dtype=np.float64)


def assert_checks(got, want, rtol, what=""):
    g, w = checks(got), np.asarray(want)
    # sum can cancel: scale its tolerance by the abs-sum
    assert abs(g[0] - w[0]) <= rtol * max(w[1], 1e-30), f"{what} sum {g[0]} vs {w[0]}"
    assert abs(g[1] - w[1]) <= rtol * max(w[1], 1e-30), f"{what} abssum {g[1]} vs {w[1]}"
    assert abs(g[2] - w[2]) <= 2 * rtol * max(w[2], 1e-30), f"{what} sqsum {g[2]} vs {w[2]}"


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


def make_synthetic_folder(folder, seed=7):
    """A small nnU-Net v2 preprocessed folder (both storage forms, volumes smaller AND larger than the patches used in the tests,
    two foreground classes of which one is empty in one case) -- the same call in tests/golden/make_loader_fixtures.py."""
    import pickle
    rs = np.random.RandomState(seed)
    for name, shape in [("case_a", (40, 56, 48)), ("case_b", (20, 70, 33)), ("case_c", (64, 64, 64)), ("case_d", (30, 30, 90)), ("case_e", (52, 41, 37))]:
        data = rs.standard_normal((1, *shape)).astype(np.float32)
        seg = np.zeros((1, *shape), dtype=np.int16)
        c = [s // 3 for s in shape]
        seg[0, c[0]:c[0] + 4, c[1]:c[1] + 5, c[2]:c[2] + 3] = 1
        if name != "case_d":
            seg[0, -6:-2, 2:6, 1:4] = 2
        locs = {k: np.argwhere(seg == k) for k in (1, 2)}          # rows (0, d, h, w) as nnU-Net stores them
        if name in ("case_b", "case_e"):
            np.savez(os.path.join(folder, name + ".npz"), data=data, seg=seg)
        else:
            np.savez(os.path.join(folder, name + ".npz"), data=data[:, :1], seg=seg[:, :1])     # (placeholder: the .npy pair is what is read)
            np.save(os.path.join(folder, name + ".npy"), data); np.save(os.path.join(folder, name + "_seg.npy"), seg)
        with open(os.path.join(folder, name + ".pkl"), "wb") as f:
            pickle.dump({"class_locations": locs, "spacing": [1.0, 1.0, 1.0]}, f)


def tiny_mednext(n_channels=8, exp_r=2, k=3):
    """A dense torch model with the layer structure and parameter names of the reference's MedNeXt encoder (P/MedNeXt_head.py:11-233,
    block_counts = 1, do_res = do_res_up_down = True) written from plain torch.nn layers, for the sparse-layer converter to swap:
    the GPU box has no reference to import.  forward(x, hierarchical=True) -> the 5 stage maps."""
    import torch.nn as nn

    class Block(nn.Module):
        def __init__(self, cin, cout, stride):
            super().__init__()
            self.stride = stride
            self.conv1 = nn.Conv3d(cin, cin, k, stride=stride, padding=k // 2, groups=cin)
            self.norm = nn.GroupNorm(num_groups=cin, num_channels=cin)
            self.conv2 = nn.Conv3d(cin, exp_r * cin, 1)
            self.act = nn.GELU()
            self.conv3 = nn.Conv3d(exp_r * cin, cout, 1)
            if stride == 2:
                self.res_conv = nn.Conv3d(cin, cout, 1, stride=2)

        def forward(self, x):
            h = self.conv3(self.act(self.conv2(self.norm(self.conv1(x)))))
            return x + h if self.stride == 1 else h + self.res_conv(x)

    class Net(nn.Module):
        def __init__(self):
            super().__init__()
            c = n_channels
            self.stem = nn.Conv3d(1, c, 1)
            for i in range(4):
                setattr(self, f"enc_block_{i}", nn.Sequential(Block(c << i, c << i, 1)))
                setattr(self, f"down_{i}", Block(c << i, c << (i + 1), 2))
            self.bottleneck = nn.Sequential(Block(c << 4, c << 4, 1))

        def forward(self, x, hierarchical=True):
            maps = []
            h = self.stem(x)
            for i in range(4):
                h = getattr(self, f"enc_block_{i}")(h)
                maps.append(h)
                h = getattr(self, f"down_{i}")(h)
            maps.append(self.bottleneck(h))
            return maps if hierarchical else maps[-1]

    return Net()


def seeded_params(model):
    """Overwrite every parameter of `model` with a draw that depends only on its NAME and shape (legacy RandomState, stable across
    numpy versions): convolution / linear weights ~ N(0, 1 / fan_in), transposed-conv weights ~ N(0, 1 / (Cout * k^3 / 8)), norm
    weights 1 + 0.1 N, biases 0.1 N, mask tokens 0.02 N.  The reference side (tests/golden/make_spark_mednext_fixture.py) and the
    GPU test both call it, so no weights are stored."""
    import zlib
    import torch
    with torch.no_grad():
        for n, p in model.named_parameters():
            if "dummy" in n:
                continue
            r = torch.from_numpy(np.random.RandomState(zlib.crc32(n.encode()) & 0x7fffffff).standard_normal(tuple(p.shape)).astype(np.float32))
            if "mask_tokens" in n:
                p.copy_(0.02 * r)
            elif p.dim() == 1:
                p.copy_((1.0 if n.endswith("weight") else 0.0) + 0.1 * r)
            elif "up_sample" in n:                       # ConvTranspose3d (Cin, Cout, 4, 4, 4): 8 of its 64 taps reach an output voxel
                p.copy_(r / float(np.sqrt(p.shape[0] * 8)))
            else:
                p.copy_(r / float(np.sqrt(p[0].numel())))


DECODER_IN_WIDTH = 128


def decoder_in_inputs(B=2, f0=(2, 2, 3)):
    """Inputs of tests/golden/make_decoder_in_fixture.py (both sides regenerate them): to_dec[i] of shape (B, WIDTH / 2^i, f0 * 2^i),
    i = 0..3 (seeds 40 + i), and the probe the reconstruction is contracted with (seed 50)."""
    import torch
    maps = [torch.from_numpy(np.random.RandomState(40 + i).standard_normal((B, DECODER_IN_WIDTH >> i, *(v << i for v in f0))).astype(np.float32))
            for i in range(4)]
    probe = torch.from_numpy(np.random.RandomState(50).standard_normal((B, 1, *(v << 4 for v in f0))).astype(np.float32))
    return maps, probe
