"""CPU, world_size 2 over gloo: the N>1 host logic of the trainer (bucket ranges per parameter group, async all-reduce
launched from the backward hook, averaging) without touching the HIP kernels.  The step's compute is replaced by writing
known per-rank gradients into the flat gradient buffer exactly where backward would."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from anatomask_amd import engine


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from anatomask_amd import modules as M
        from anatomask_amd.trainer import AnatoMaskTrainer
        torch.manual_seed(rank)                                  # deliberately different init per rank
        m = M.build_spark([8, 16, 32, 64, 128, 128], [1] * 6, 128, (32, 48, 64))
        # host-only stand-in for the flat buffers _ensure_flat() builds on the device
        named = list(m.named_parameters())
        order = [(n, p) for n, p in named if n not in m._dead] + [(n, p) for n, p in named if n in m._dead]
        offs, tot = {}, 0
        for n, p in order:
            if n in m._dead and "live_end" not in offs:
                offs["live_end"] = tot
            offs[n] = tot; tot += (p.numel() + 3) // 4 * 4
        m._offs, m._live_end, m._pnames = offs, offs["live_end"], [n for n, _ in named]
        m._flat = torch.cat([torch.cat([p.detach().flatten(), torch.zeros((-p.numel()) % 4)]) for _, p in order])
        m._gflat = torch.zeros(tot)
        tr = AnatoMaskTrainer.__new__(AnatoMaskTrainer)
        tr.model, tr.distributed, tr.pg, tr.world, tr._works, tr.exchange_log = m, True, None, world, [], []
        tr._pieces = tr._t_begin = tr._t_end = None
        tr.BUCKET_BYTES, tr.FLUSH_BYTES = 256 << 10, 96 << 10     # small limits so that this 3 M-parameter model splits and merges
        from anatomask_amd.engine import Spec
        m.spec = Spec([8, 16, 32, 64, 128, 128], [1] * 6, 128, (32, 48, 64))
        AnatoMaskTrainer._build_ranges(tr)
        # 1. start-up broadcast makes the replicas identical (P/pretrain_AnatoMask_DDP.py:239-240)
        dist.broadcast(m._flat, 0)
        ref = m._flat.clone(); dist.broadcast(ref, 0)
        assert torch.equal(m._flat, ref)
        # 2. the groups tile the live region exactly; tags in backward completion order proj, dec3..0, densify, stage4.0 .. stage0.0
        r = tr._ranges
        order = ["proj"] + [f"dec{i}" for i in reversed(range(4))] + ["densify"] + [f"stage{s}.0" for s in reversed(range(5))]
        assert set(r) == set(order)
        spans = sorted(r.values())
        assert spans[0][0] == 0 and spans[-1][1] == m._live_end
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        assert r["dec0"][0] == offs["dense_decoder.dec.0.up_sample.weight"] and r["densify"][1] == m._live_end
        assert r["stage0.0"][0] == 0 and r["stage4.0"][1] == r["dec0"][0] and r["proj"][1] == r["densify"][0]
        # 3. async all-reduce from the backward hook == mean over ranks; the exchange starts before backward ends, no collective
        #    exceeds BUCKET_BYTES, every live element is sent exactly once
        m._gflat[:m._live_end] = float(rank + 1)
        m._gflat[m._live_end:] = 123.0                           # dead tensors are never exchanged
        sent_after = {}
        for tag in order:
            tr._after_group(tag)
            sent_after[tag] = len(tr.exchange_log)
        assert sent_after["dec0"] > 0, "nothing was sent before the encoder's backward"
        tr._finish_exchange()
        log = tr.exchange_log
        assert all((b - a) * 4 <= tr.BUCKET_BYTES for a, b in log) and len(log) > len(order) // 2
        cover = sorted(log)
        assert cover[0][0] == 0 and cover[-1][1] == m._live_end and all(x[1] == y[0] for x, y in zip(cover, cover[1:]))
        assert not tr._pending and not tr._works
        # the buffer holds the SUM over ranks; DDP's 1/world is a factor of the fused optimizer kernel (am_adamw_ema grad_scale)
        want = sum(range(1, world + 1)) / world
        assert AnatoMaskTrainer.grad_scale.fget(tr) == 1.0 / world
        assert torch.allclose(m._gflat[:m._live_end] * AnatoMaskTrainer.grad_scale.fget(tr), torch.full((m._live_end,), want))
        assert torch.all(m._gflat[m._live_end:] == 123.0)
        q.put((rank, "ok"))
    except Exception as e:  # noqa: BLE001
        q.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_gradient_exchange_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = [q.get(timeout=100) for _ in ps]
    for p in ps:
        p.join(20)
    assert all(v == "ok" for _, v in res), res
