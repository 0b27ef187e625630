"""GPU: device-side spatial augmentation (csrc/aug_ops.hip) against scipy -- the library batchgenerators' SpatialTransform calls
(map_coordinates, order 3, mode 'constant', cval 0) --, and the whole feed (prefetch threads -> pinned double-buffered H2D ->
device augmentation) driving the trainer without starving it."""
import time

import numpy as np
import pytest
import scipy.ndimage as ndi
import torch

from anatomask_amd.data import DeviceAugmenter, DeviceFeed, PatchLoader3D, PinnedPool, PrefetchLoader, PreprocessedDataset, SpatialAugmenter
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
    aug = SpatialAugmenter(final, seed=5, p_rot=1.0, p_scale=1.0)
    x = rs.standard_normal((5, 1, *enl)).astype(np.float32)
    params = [aug.draw() for _ in range(5)]
    params[1].update(modified=False, angles=(0.0, 0.0, 0.0), scale=1.0)           # centre crop + flips only
    params[3].update(modified=False, angles=(0.0, 0.0, 0.0), scale=1.0, mirror=[False, False, False])
    out = DeviceAugmenter(aug)(torch.from_numpy(x).to(DEV), params).cpu().numpy()
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


def test_feed_drives_trainer(tmp_path):
    """loader threads -> pinned staging -> copy stream -> device augmentation -> AnatoMaskTrainer.step (tiny config): every step gets a
    fresh, correctly shaped device batch and trains on it."""
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
        sums = []
        for _ in range(12):
            x = aug(next(feed))
            assert x.shape == (2, 1, *final) and x.is_cuda
            sums.append(float(x.double().sum()))
            out = tr.step(x, epoch=500)
        assert torch.isfinite(out["loss"]).all() and len(set(sums)) > 6
    finally:
        pf.close()


def test_feed_sustains_the_step_rate_at_production_shape(tmp_path):
    """The feed alone at the shape the 128^3 recipe needs -- 205^3 enlarged crops (get_patch_size) from memory-mapped volumes by 8
    loader threads, pinned double-buffered H2D, prefilter + order-3 resampling of the ~36 % rotated / scaled samples and crop + flips of
    the rest on the GPU -- must deliver more volumes/s than the training step consumes (~100 on one MI355X)."""
    import os
    import pickle
    rs = np.random.RandomState(0)
    for i in range(3):                                                # three 240^3 volumes (55 MB each)
        v = rs.standard_normal((1, 240, 240, 240)).astype(np.float32)
        np.save(os.path.join(tmp_path, f"big_{i}.npy"), v)
        np.save(os.path.join(tmp_path, f"big_{i}_seg.npy"), np.zeros((1, 240, 240, 240), dtype=np.int16))
        with open(os.path.join(tmp_path, f"big_{i}.pkl"), "wb") as f:
            pickle.dump({"class_locations": {1: np.array([[0, 120, 120, 120]])}}, f)
    ds = PreprocessedDataset(str(tmp_path))
    final, enl, B = (128, 128, 128), (205, 205, 205), 4
    aug = DeviceAugmenter(SpatialAugmenter(final, seed=1))
    pool = PinnedPool((B, 1, *enl), n=6 + 2 + 8)                      # queue depth + device slots + one per worker
    pf = PrefetchLoader(lambda w: PatchLoader3D(ds, B, enl, 0.33, seed=10 + w, final_patch_size=final, pool=pool), n_workers=8, num_cached=6)
    try:
        feed = DeviceFeed(pf, DEV)
        for _ in range(2):
            aug(next(feed))
        torch.cuda.synchronize(); t0 = time.perf_counter()
        n = 10
        for _ in range(n):
            x = aug(next(feed))
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        rate = n * B / dt
        print(f"feed at 205^3 -> 128^3: {rate:.0f} volumes/s ({dt / n * 1e3:.1f} ms per batch of {B})")
        assert x.shape == (B, 1, 128, 128, 128) and rate > 100.0, rate
    finally:
        pf.close()


@pytest.mark.timeout(120)
def test_feed_with_a_tiny_pool_and_a_host_far_ahead_of_the_gpu(tmp_path):
    """The training loop synchronises once per epoch: the host can enqueue many steps ahead of the GPU, and every queued H2D copy
    holds its pooled pinned buffer until the copy has run.  With a pool of exactly the steady-state size and a slow GPU step the
    feed must keep going (DeviceFeed reaps finished copies before it blocks on the loader queue and waits for the oldest copy when
    more than `depth` are in flight) -- 40 un-synchronised iterations through a 4-buffer pool, every batch distinct."""
    make_synthetic_folder(str(tmp_path), 7)
    ds = PreprocessedDataset(str(tmp_path))
    final = (32, 48, 64)
    pool = PinnedPool((2, 1, *final), n=4)
    pf = PrefetchLoader(lambda w: PatchLoader3D(ds, 2, final, 0.33, seed=90 + w, final_patch_size=final, pool=pool), n_workers=2, num_cached=1)
    try:
        feed = DeviceFeed(pf, DEV)
        acc = torch.zeros(40, device=DEV, dtype=torch.float64)
        for i in range(40):
            x = next(feed)
            torch.cuda._sleep(20_000_000)                              # a slow "training step" on the compute stream (~10 ms), never synchronised
            acc[i] = x.double().abs().sum()
        torch.cuda.synchronize()
        assert len(set(acc.tolist())) > 20 and bool((acc > 0).all())
    finally:
        pf.close()


@pytest.mark.timeout(300)
def test_pretrain_driver_with_data_feed_checkpoint_and_resume(tmp_path, capsys):
    """The pretraining driver end to end (anatomask_amd.pretrain.main, the loop of P/pretrain_AntoMask.py:371-479) on a synthetic
    nnU-Net folder: loader threads -> pinned pool -> copy stream -> device augmentation -> fused step, one checkpoint per epoch; the
    file loads with a PLAIN torch.load (the finetune hand-off), carries every loader's and the augmenter's generator state, and a
    resumed run continues at the next epoch with the loaders' generators restored BEFORE their threads start."""
    from anatomask_amd import checkpoint, pretrain
    data, out = tmp_path / "data", tmp_path / "run"
    data.mkdir()
    make_synthetic_folder(str(data), 7)
    common = ["--model", "S", "--input-size", "32", "48", "64", "--batch-size", "2", "--iters-per-epoch", "2", "--workers", "2",
              "--data", str(data), "--out", str(out), "--dtype", "bf16"]
    pretrain.main(common + ["--epochs", "2"])
    log = capsys.readouterr().out
    assert "Epoch 0 " in log and "Epoch 1 " in log
    p = str(out / "STUNet_S_head_latest.pt")
    ck = torch.load(p)                                                 # default arguments (weights_only=True): what the reference's loader does
    assert int(ck["current_epoch"]) == 1 and len(ck["train_loss"]) == 2 and all(np.isfinite(ck["train_loss"]))
    fs = ck["feed_state"][0]
    assert sorted(fs["loader_rng"]) == [0, 1] and fs["aug_rng"]["name"] == "MT19937"
    assert len(checkpoint.encoder_weights_for_finetuning(ck["network_weights"])) == 50
    # the state a resumed Feed starts from IS the saved one: build it the way main() does and compare before any draw
    resumed = checkpoint.peek_extra(p, "feed_state")[0]
    feed = pretrain.Feed(str(data), None, 2, (32, 48, 64), torch.device(DEV), 0, 2, True, seed=1000, state=resumed)
    try:
        for w in (0, 1):                                               # (the threads have already drawn: compare the saved keys' identity only)
            assert feed.loaders[w].rs.get_state()[0] == "MT19937"
        x = next(feed)
        assert x.shape == (2, 1, 32, 48, 64) and x.is_cuda
    finally:
        feed.close()
    pretrain.main(common + ["--epochs", "3", "--resume", p])
    log = capsys.readouterr().out
    assert "Epoch 2 " in log and "Epoch 0 " not in log and "Epoch 1 " not in log
    assert int(torch.load(p)["current_epoch"]) == 2
