"""Why the step falls off a cliff between B=24 and B=32: python tools/batch_cliff.py [batches...]
Per batch size: ms/step (median of HIP-event step times), the caching allocator's peak reserved / allocated bytes, the number of
allocation RETRIES (a retry = hipMalloc failed, every cached block was released and the device synchronised) and of hipMalloc calls
inside the timed steps; with the side stream off as well (record_stream keeps blocks alive until the side stream's kernels end)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anatomask_amd import engine, modules as M  # noqa: E402
from anatomask_amd.trainer import AnatoMaskTrainer  # noqa: E402

batches = [int(v) for v in sys.argv[1:]] or [16, 24, 32]
dev = torch.device("cuda:0")
kw = M.STUNET_CONFIGS["B"]
print(f"device memory: {torch.cuda.get_device_properties(dev).total_memory / 2 ** 30:.0f} GiB", flush=True)
for side in (True, False):
    engine._USE_SIDE = side
    for B in batches:
        torch.manual_seed(0)
        model = M.build_spark(kw["dims"], kw["depth"], kw["width"], (128,) * 3, 0.6, compute_dtype=torch.bfloat16).to(dev)
        tr = AnatoMaskTrainer(model, lr=1e-4, total_epochs=1000, seed=1, distributed=False)
        tr.set_epoch(500)
        x = torch.randn(B, 1, 128, 128, 128, device=dev)
        for _ in range(3):
            tr.step(x, epoch=500)
        torch.cuda.synchronize()
        torch.cuda.reset_peak_memory_stats()
        s0 = torch.cuda.memory_stats()
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(7)]
        evs[0].record()
        for i in range(6):
            tr.step(x, epoch=500)
            evs[i + 1].record()
        torch.cuda.synchronize()
        s1 = torch.cuda.memory_stats()
        per = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(6))
        print(f"side_stream={side} B={B}: {per[3]:.1f} ms/step median (min {per[0]:.1f}, max {per[-1]:.1f}) = {B / per[3] * 1e3:.1f} volumes/s | peak reserved "
              f"{s1['reserved_bytes.all.peak'] / 2 ** 30:.1f} GiB, peak allocated {s1['allocated_bytes.all.peak'] / 2 ** 30:.1f} GiB | retries "
              f"{s1['num_alloc_retries'] - s0['num_alloc_retries']}, hipMalloc calls {s1['num_device_alloc'] - s0['num_device_alloc']}, "
              f"hipFree calls {s1['num_device_free'] - s0['num_device_free']} in 6 steps", flush=True)
        del tr, model, x
        torch.cuda.empty_cache()
