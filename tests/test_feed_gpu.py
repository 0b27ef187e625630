"""GPU: device-side spatial augmentation (csrc/aug_ops.hip) against scipy -- the library batchgenerators' SpatialTransform calls
(map_coordinates, order 3, mode 'constant', cval 0) --, and the whole feed (prefetch threads -> pinned double-buffered H2D ->
device augmentation) driving the trainer without starving it."""
import time

import numpy as np
import pytest
import scipy.ndimage as ndi
import torch

from anatomask_amd.data import DeviceAugmenter, DeviceFeed, PatchLoader3D, PrefetchLoader, PreprocessedDataset, SpatialAugmenter
from oracle import anatomask_oracle as O
from tests.helpers import make_synthetic_folder
from tests.test_data_feed import _batchgenerators_coords

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_spline_prefilter_equals_scipy():
    from anatomask_amd import ops
    rs = np.random.RandomState(0)
    for shape in ((12, 14, 16), (45, 33, 70), (5, 64, 9)):
        v = rs.standard_normal(shape).astype(np.float32)
        want = ndi.spline_filter(v.astype(np.float64), order=3, mode="mirror")
        got = ops.spline_prefilter(torch.from_numpy(v.copy()).to(DEV)).cpu().numpy()
        assert np.abs(got - want).max() < 2e-5 * np.abs(want).max(), shape


@pytest.mark.parametrize("order", [0, 1, 3])
def test_resample_affine_equals_map_coordinates(order):
    from anatomask_amd import ops
    rs = np.random.RandomState(1)
    src = rs.standard_normal((21, 25, 17)).astype(np.float32)
    final = (12, 16, 10)
    aug = SpatialAugmenter(final, seed=3, p_rot=1.0, p_scale=1.0, order=order)
    for trial in range(4):
        p = aug.draw()
        if order == 0:
            p.update(modified=False, angles=(0.0, 0.0, 0.0), scale=1.0)
        A = aug.affine(p, src.shape)
        idx = np.stack(np.meshgrid(*[np.arange(n, dtype=np.float64) for n in final], indexing="ij"))
        coords = np.einsum("ij,jdhw->idhw", A[:, :3], idx) + A[:, 3][:, None, None, None]
        want = ndi.map_coordinates(src.astype(np.float64), coords, order=max(order, 0) if order else 0, mode="constant", cval=0.0)
        s = torch.from_numpy(src.copy()).to(DEV)
        if order == 3:
            ops.spline_prefilter(s)
        got = ops.resample_affine(s, torch.empty(final, device=DEV), A.reshape(-1), order).cpu().numpy()
        # points within fp32 rounding of the [0, n-1] boundary may fall on either side of scipy's inside test
        edge = np.zeros(final, bool)
        for d in range(3):
            edge |= (np.abs(coords[d]) < 1e-3) | (np.abs(coords[d] - (src.shape[d] - 1)) < 1e-3)
        err = np.abs(got - want)[~edge]
        assert err.max() < 2e-4 * max(np.abs(want).max(), 1.0), (order, trial, err.max())


def test_device_augmenter_equals_batchgenerators_pipeline():
    """rotation + scale + mirror of an enlarged batch on the GPU == coordinate mesh -> rotate -> scale -> centre -> map_coordinates(order 3)
    -> flip, sample by sample."""
    rs = np.random.RandomState(2)
    final, enl = (16, 16, 24), (28, 28, 40)
    aug = SpatialAugmenter(final, seed=5, p_rot=0.6, p_scale=0.6)
    x = rs.standard_normal((5, 1, *enl)).astype(np.float32)
    params = [aug.draw() for _ in range(5)]
    out = DeviceAugmenter(aug)(torch.from_numpy(x).to(DEV), params).cpu().numpy()
    assert any(p["modified"] for p in params) and any(not p["modified"] for p in params)
    for b, p in enumerate(params):
        coords = _batchgenerators_coords(final, enl, p)
        want = ndi.map_coordinates(x[b, 0].astype(np.float64), coords, order=3, mode="constant", cval=0.0) if p["modified"] else \
            x[b, 0][tuple(slice((enl[d] - final[d]) // 2, (enl[d] - final[d]) // 2 + final[d]) for d in range(3))]
        for d in range(3):
            if p["mirror"][d]:
                want = np.flip(want, axis=d)
        err = np.abs(out[b, 0] - want)
        if p["modified"]:
            assert np.quantile(err, 0.999) < 5e-4 and err.mean() < 1e-5, (b, err.max())      # (a few voxels sit on the inside/outside edge)
        else:
            assert err.max() == 0.0, b                                                        # crops / flips are exact


def test_feed_drives_trainer_without_starving_it(tmp_path):
    """loader threads -> pinned staging -> copy stream -> device augmentation -> AnatoMaskTrainer.step: per-step wall time within
    1.5x of the same steps on a resident batch (tiny config: the step is short, i.e. the feed has little time to hide in)."""
    from anatomask_amd import modules as M
    from anatomask_amd.trainer import AnatoMaskTrainer
    make_synthetic_folder(str(tmp_path), 7)
    ds = PreprocessedDataset(str(tmp_path))
    final, enl = (32, 48, 64), (40, 56, 72)
    cfg = O.Config([8, 16, 32, 64, 128, 128], [1] * 6, 128, final, 0.6)
    m = M.build_spark(cfg.dims, cfg.depth, cfg.width, cfg.input_size, cfg.mask_ratio, compute_dtype=torch.bfloat16).to(DEV)
    tr = AnatoMaskTrainer(m, lr=1e-4, total_epochs=1000, distributed=False, seed=3)
    aug = DeviceAugmenter(SpatialAugmenter(final, seed=1))
    pf = PrefetchLoader(lambda w: PatchLoader3D(ds, 2, enl, 0.33, seed=50 + w, final_patch_size=final), n_workers=3, num_cached=6)
    try:
        feed = DeviceFeed(pf, DEV)
        xres = torch.randn(2, 1, *final, device=DEV)
        for _ in range(3):
            tr.step(xres, epoch=500); tr.step(aug(next(feed)), epoch=500)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(15):
            out = tr.step(xres, epoch=500)
        torch.cuda.synchronize(); t_res = time.perf_counter() - t0
        t0 = time.perf_counter()
        for _ in range(15):
            out = tr.step(aug(next(feed)), epoch=500)
        torch.cuda.synchronize(); t_feed = time.perf_counter() - t0
        assert torch.isfinite(out["loss"]).all()
        print(f"resident {t_res / 15 * 1e3:.1f} ms/step, fed {t_feed / 15 * 1e3:.1f} ms/step")
        assert t_feed < 1.5 * t_res + 0.05, (t_feed, t_res)
    finally:
        pf.close()
