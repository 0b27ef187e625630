"""Host-side mirror of the reference interface: names, state_dict contract and error conventions.  No GPU needed."""
import pytest
import torch

from anatomask_amd import modules as M
from oracle import anatomask_oracle as O

DIMS, WIDTH, SIZE = [8, 16, 32, 64, 128, 128], 128, (32, 48, 64)


def build(mask_ratio=0.6):
    return M.build_spark(DIMS, [1] * 6, WIDTH, SIZE, mask_ratio)


def test_state_dict_contract_131_keys_and_shapes():
    """Same 131 keys / shapes as the reference model (pinned through the oracle's inventory, tests/test_oracle_golden.py)."""
    sd = build().state_dict()
    ps = O.param_shapes(O.Config(DIMS, [1] * 6, WIDTH, SIZE, 0.6))
    assert len(sd) == 131 and set(sd) == set(ps)
    assert all(tuple(sd[k].shape) == tuple(ps[k]) for k in ps)
    assert "config" in build().state_dict(with_config=True)                       # P/AnatoMask.py:257-262


def test_error_conventions():
    m = build()
    with pytest.raises(RuntimeError, match="HIP engine"):                         # no CPU / torch fallback path exists
        m(torch.randn(1, 1, *SIZE))
    with pytest.raises(AttributeError, match="config mismatch"):                  # P/AnatoMask.py:268-276
        build(0.5).load_state_dict(m.state_dict(with_config=True), strict=True)
    with pytest.raises(RuntimeError):                                             # P/AnatoMask.py:225: extent not divisible by 16
        m.patchify(torch.randn(1, 1, 30, 48, 64))


def test_mask_and_schedules_and_wrappers():
    m = build()
    mk = m.mask(3, "cpu", generator=torch.Generator().manual_seed(0))             # P/AnatoMask.py:75-79
    assert mk.shape == (3, 1, 2, 3, 4) and mk.dtype == torch.bool and int(mk.sum()) == 3 * m.len_keep
    assert m.len_keep == round(24 * 0.4)
    assert m.sparse_encoder.downsample_ratio == 16 and m.dense_decoder.width == WIDTH
    ema = M.ModelEma(m, 0.999, device=None)
    assert isinstance(ema.ema, M.SparK) and ema.decay == 0.999 and not ema.ema.training      # teacher is always eval
    ddp = M.LocalDDP(m)
    assert next(iter(ddp.state_dict())).startswith("module.")                     # P/pretrain_AntoMask.py:201-207
    groups = M.get_param_groups(m, nowd_keys={"mask_token"})
    assert len(groups) == 2 and sum(len(g["params"]) for g in groups) == sum(1 for _ in m.parameters())
    assert M.ema_decay_for_epoch(0, 1000) == pytest.approx(0.999) and M.ema_decay_for_epoch(999, 1000) == pytest.approx(0.9999)


def test_feed_rng_state_is_weights_only_safe_and_round_trips(tmp_path):
    """The checkpoint's `feed_state` (loader / augmenter RandomStates) must survive the reference's plain `torch.load(fname)`
    (nnunetv2/run/load_pretrained_weights.py; weights_only=True by default from torch 2.6): tensors and python scalars only."""
    import numpy as np
    from anatomask_amd.pretrain import default_workers, rng_state_from_plain, rng_state_to_plain
    rs = np.random.RandomState(1234)
    rs.standard_normal(7)                                                         # has_gauss / cached_gaussian populated
    st = {"feed_state": {0: {"loader_rng": {0: rng_state_to_plain(rs)}, "aug_rng": rng_state_to_plain(np.random.RandomState(5))}},
          "network_weights": {"module.x": torch.zeros(2)}, "grad_scaler_state": None, "train_loss": [1.0], "current_epoch": 0}
    p = str(tmp_path / "ck.pt")
    torch.save(st, p)
    ck = torch.load(p)                                                            # default arguments, as the reference calls it
    want = rs.uniform(size=5), rs.standard_normal(3)
    rs2 = np.random.RandomState(0)
    rng_state_from_plain(rs2, ck["feed_state"][0]["loader_rng"][0])
    assert np.array_equal(rs2.uniform(size=5), want[0]) and np.array_equal(rs2.standard_normal(3), want[1])
    assert 1 <= default_workers(8) <= 8 and default_workers(1) >= default_workers(8)


def test_read_checkpoint_never_unpickles_code(tmp_path):
    """checkpoint.read_checkpoint: a file is data.  A format-2 file of this package (numpy RandomState tuples in `feed_state`) is read
    through the restricted unpickler with plain-ndarray reconstruction allowed and loses its feed state; a pickle that names any other
    global (here: one whose unpickling would run `os.system`) is refused and its payload is NOT executed."""
    import os
    import pickle

    import numpy as np
    import pytest

    from anatomask_amd import checkpoint
    old = str(tmp_path / "format2.pt")
    torch.save({"feed_state": {0: {"loader_rng": {0: np.random.RandomState(3).get_state()}}}, "anatomask_amd_version": 2,
                "network_weights": {"module.x": torch.ones(2)}, "current_epoch": 4}, old)
    with pytest.warns(UserWarning, match="loader state is dropped"):
        ck = checkpoint.read_checkpoint(old)
    assert "feed_state" not in ck and int(ck["current_epoch"]) == 4 and torch.equal(ck["network_weights"]["module.x"], torch.ones(2))

    witness = str(tmp_path / "executed")

    class Evil:
        def __reduce__(self):
            return (os.system, (f"touch {witness}",))
    bad = str(tmp_path / "evil.pt")
    torch.save({"network_weights": {}, "payload": Evil(), "anatomask_amd_version": 2}, bad)
    with pytest.raises(pickle.UnpicklingError):
        checkpoint.read_checkpoint(bad)
    assert not os.path.exists(witness)
    # a file that CLAIMS format 3 but needs numpy globals is not ours either
    liar = str(tmp_path / "liar.pt")
    torch.save({"feed_state": np.arange(3), "anatomask_amd_version": 3}, liar)
    with pytest.raises(pickle.UnpicklingError):
        checkpoint.read_checkpoint(liar)


def test_easy_mask_is_the_band_below_the_hard_patches():
    """generate_mask's second output (P/AnatoMask.py:116-134): the `L - len_keep - len_loss` patches ranked just below the hard band are
    hidden, the other len_keep + len_loss are visible; in the random regime it is the mask itself.  (Host-side: SparK._easy_mask.)"""
    m = build()
    L = m.fmap_h * m.fmap_w * m.fmap_d
    g = torch.Generator().manual_seed(3)
    loss = torch.rand(2, L, generator=g)
    ll = m.len_loss_for(L, m.len_keep, 120, 200)
    assert 0 < ll < L - m.len_keep
    mask = torch.zeros(2, 1, m.fmap_h, m.fmap_w, m.fmap_d, dtype=torch.bool)
    easy = m._easy_mask(loss, mask, ll).view(2, L)
    assert (easy.sum(1) == m.len_keep + ll).all()
    order = loss.argsort(1)
    for b in range(2):
        hidden = set(order[b, m.len_keep:L - ll].tolist())             # ascending ranks [L - ll - easy_len, L - ll), easy_len = L - len_keep - ll
        assert hidden == set((~easy[b]).nonzero().flatten().tolist())
    assert torch.equal(m._easy_mask(loss, mask, 0), mask)                # random regime


def test_forward_learning_loss_matches_its_definition():
    """P/AnatoMask.py:204-219 (unused by the drivers): per-image normalised target, MSE against the prediction."""
    m = build()
    g = torch.Generator().manual_seed(9)
    tgt, pred = torch.rand(3, 40, generator=g), torch.randn(3, 40, generator=g, requires_grad=True)
    want = ((pred - (tgt - tgt.mean(1, keepdim=True)) / (tgt.var(1, keepdim=True) + 1e-6) ** .5) ** 2).mean()
    got = m.forward_learning_loss(pred, tgt)
    assert torch.allclose(got, want) and got.requires_grad
