"""Data feed on CPU: the folder reader + patch sampler against fixtures produced by RUNNING THE REFERENCE's own loader
(tests/golden/make_loader_fixtures.py -> loader_tiny.npz: bit-exact batches), the enlarged-patch rule, the spatial-augmentation
arithmetic against the batchgenerators procedure evaluated step by step with scipy, and the prefetcher."""
import os
import time

import numpy as np
import pytest
import torch

from anatomask_amd.data import (PatchLoader3D, PrefetchLoader, PreprocessedDataset, ROTATION_FOR_DA, SpatialAugmenter, get_patch_size,
                                _rot_x, _rot_y, _rot_z)
from tests.helpers import checks, load, make_synthetic_folder


@pytest.fixture(scope="module")
def fx():
    return load("loader_tiny.npz")


def test_enlarged_patch_size_matches_reference(fx):
    rot = ROTATION_FOR_DA
    for tag in ("128", "112", "48", "tiny"):
        v = fx[f"patchsize_{tag}"]
        assert list(get_patch_size(tuple(v[:3]), rot, rot, rot, (0.85, 1.25))) == list(v[3:]), tag
    assert list(fx["patchsize_128"][3:]) == [205, 205, 205]


@pytest.mark.parametrize("run", [0, 1])
def test_batches_bit_exact_vs_reference_loader(fx, tmp_path, run):
    """Same folder, same seed (legacy MT19937 stream) -> the reference's nnUNetDataLoader3D and PatchLoader3D select the same cases,
    the same bounding boxes (foreground oversampling of the last 33 %, need_to_pad of the enlarged patch, volumes smaller than the
    patch) and produce identical data / seg arrays."""
    make_synthetic_folder(str(tmp_path), 7)
    cfg = fx[f"r{run}_cfg"]
    B, seed, initial, final = int(cfg[0]), int(cfg[1]), tuple(int(v) for v in cfg[2:5]), tuple(int(v) for v in cfg[5:8])
    ds = PreprocessedDataset(str(tmp_path))
    assert ds.keys() == ["case_a", "case_b", "case_c", "case_d", "case_e"]
    dl = PatchLoader3D(ds, B, initial, 0.33, seed=seed, final_patch_size=final)
    boxes = []
    orig = dl._bbox
    dl._bbox = lambda *a, **k: (lambda r: (boxes.append(np.array(r)), r)[1])(orig(*a, **k))
    for it in range(3):
        b = next(dl)
        assert list(b["keys"]) == [str(k) for k in fx[f"r{run}_b{it}_keys"]], (run, it)
        assert b["data"].shape == (B, 1, *initial) and b["data"].dtype == torch.float32 and b["seg"].dtype == torch.int16
        got = np.stack([checks(b["data"][j]) for j in range(B)])
        np.testing.assert_array_equal(got, fx[f"r{run}_b{it}_data_checks"])
        np.testing.assert_array_equal(b["seg"].numpy().astype(np.int64).sum(axis=(1, 2, 3, 4)), fx[f"r{run}_b{it}_seg_sum"])
        np.testing.assert_array_equal(b["data"].numpy().reshape(B, -1)[:, ::997], fx[f"r{run}_b{it}_data_sample"])
    np.testing.assert_array_equal(np.stack(boxes), fx[f"r{run}_bbox"])


def test_forced_foreground_and_padding(tmp_path):
    make_synthetic_folder(str(tmp_path), 7)
    ds = PreprocessedDataset(str(tmp_path))
    dl = PatchLoader3D(ds, batch_size=6, patch_size=(32, 48, 32), oversample_foreground_percent=0.33, seed=3)
    b = next(dl)
    assert [dl._force_fg(j) for j in range(6)] == [False] * 4 + [True] * 2
    assert all((b["seg"][j] > 0).any() for j in (4, 5))              # the last round(6 * 0.33) = 2 samples contain labelled voxels
    for j, k in enumerate(b["keys"]):
        if k == "case_b":                                            # 20 voxels deep < 32: data padded with 0, seg with -1
            pad_planes = (b["seg"][j, 0] == -1).all(dim=2).all(dim=1)
            assert int(pad_planes.sum()) == 12 and bool((b["data"][j, 0][pad_planes] == 0).all())


def _batchgenerators_coords(final, in_shape, p):
    """augment_spatial's coordinate pipeline, step by step (batchgenerators >= 0.25, published algorithm): zero-centred mesh ->
    rotate_coords_3d -> scale -> + centre of the input patch."""
    mesh = np.stack(np.meshgrid(*[np.arange(n, dtype=np.float64) - (n - 1) / 2.0 for n in final], indexing="ij"))
    if p["modified"]:
        R = _rot_x(p["angles"][0]) @ _rot_y(p["angles"][1]) @ _rot_z(p["angles"][2])
        mesh = np.dot(mesh.reshape(3, -1).T, R).T.reshape(mesh.shape) * p["scale"]
        for d in range(3):
            mesh[d] += in_shape[d] / 2.0 - 0.5
    else:                                                            # center_crop_aug
        for d in range(3):
            mesh[d] += (in_shape[d] - final[d]) // 2 + (final[d] - 1) / 2.0
    return mesh


def test_affine_equals_batchgenerators_coordinate_pipeline():
    final, in_shape = (12, 16, 10), (21, 25, 17)
    aug = SpatialAugmenter(final, seed=4, p_rot=0.7, p_scale=0.7)
    seen = set()
    for _ in range(40):
        p = aug.draw()
        seen.add((p["modified"], any(p["mirror"])))
        A = aug.affine(p, in_shape)
        idx = np.stack(np.meshgrid(*[np.arange(n, dtype=np.float64) for n in final], indexing="ij"))
        for d in range(3):                                           # MirrorTransform acts on the OUTPUT array
            if p["mirror"][d]:
                idx[d] = final[d] - 1 - idx[d]
        want = _batchgenerators_coords(final, in_shape, dict(p, mirror=[False] * 3))
        # mirrored output o reads what the un-mirrored output had at n-1-o
        want_m = want
        for d in range(3):
            if p["mirror"][d]:
                want_m = np.flip(want_m, axis=d + 1)
        got = np.einsum("ij,jdhw->idhw", A[:, :3], np.stack(np.meshgrid(*[np.arange(n, dtype=np.float64) for n in final], indexing="ij"))) \
            + A[:, 3][:, None, None, None]
        assert np.abs(got - want_m).max() < 1e-9
    assert (True, True) in seen and (False, False) in seen or len(seen) >= 3
    a = SpatialAugmenter(final, seed=9)                              # the reference's probabilities: ~36 % of the samples are resampled
    assert 0.25 < np.mean([a.draw()["modified"] for _ in range(2000)]) < 0.47


def test_prefetch_loader_threads(tmp_path):
    make_synthetic_folder(str(tmp_path), 7)
    ds = PreprocessedDataset(str(tmp_path))
    pf = PrefetchLoader(lambda w: PatchLoader3D(ds, 2, (16, 16, 16), seed=100 + w), n_workers=3, num_cached=4)
    try:
        t0 = time.time()
        bs = [next(pf) for _ in range(12)]
        assert time.time() - t0 < 30
        assert all(b["data"].shape == (2, 1, 16, 16, 16) for b in bs)
        assert len({tuple(b["keys"]) for b in bs}) > 1
    finally:
        pf.close()


def test_unpack_dataset_once_and_same_batches(tmp_path):
    """nnU-Net's unpack_dataset step: .npz -> .npy once (then memory-mapped reads); the sampled batches do not change."""
    make_synthetic_folder(str(tmp_path), 7)
    ds = PreprocessedDataset(str(tmp_path))
    before = [next(PatchLoader3D(ds, 2, (16, 24, 16), seed=5))["data"].clone() for _ in range(1)]
    assert ds.unpack() == 2 and ds.unpack() == 0                 # case_b / case_e had no .npy; a second call does nothing
    assert os.path.isfile(os.path.join(str(tmp_path), "case_b.npy")) and os.path.isfile(os.path.join(str(tmp_path), "case_e_seg.npy"))
    data, seg, _ = ds.load_case("case_b")
    assert isinstance(data, np.memmap) and data.shape == (1, 20, 70, 33)
    after = [next(PatchLoader3D(ds, 2, (16, 24, 16), seed=5))["data"] for _ in range(1)]
    assert all(torch.equal(a, b) for a, b in zip(before, after))


def test_cpulist_parsing_and_pinned_prefetch_threads():
    """loader threads pinned to a CPU set (the rank's GPU NUMA node on the GPU box): the kernel's cpulist format, and the affinity the
    worker threads actually run with."""
    import os
    from anatomask_amd.data import PrefetchLoader, parse_cpulist
    assert parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11] and parse_cpulist("5") == [5] and parse_cpulist("") == []
    allowed = sorted(os.sched_getaffinity(0))
    want = allowed[:1]
    seen = []

    def make(w):
        def gen():
            for i in range(3):
                seen.append(sorted(os.sched_getaffinity(0)))
                yield {"data": i}
        return gen()
    pf = PrefetchLoader(make, n_workers=2, num_cached=2, cpus=want)
    got = [next(pf)["data"] for _ in range(4)]
    pf.close()
    assert len(got) == 4 and all(s == want for s in seen)
    assert sorted(os.sched_getaffinity(0)) == allowed          # the calling thread keeps its own mask
