"""Soak run of the fused trainer at the headline configuration: N AnatoMask steps of STUNet-B 128^3 bf16 over a pool of learnable synthetic volumes (smooth
random fields + noise), learning-rate warm-up, EMA decay ramp: loss every 25 steps, step time, allocator state -- no NaN, no drift of memory or step time,
and the loss has to come down.    python tools/soak.py [steps] [batch]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anatomask_amd import modules as M  # noqa: E402
from anatomask_amd.trainer import AnatoMaskTrainer  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dev = torch.device("cuda:0")
kw = M.STUNET_CONFIGS["B"]
torch.manual_seed(0)
model = M.build_spark(kw["dims"], kw["depth"], kw["width"], (128,) * 3, 0.6, compute_dtype=torch.bfloat16).to(dev)
tr = AnatoMaskTrainer(model, lr=1e-4, total_epochs=100, seed=1)
g = torch.Generator(device=dev).manual_seed(5)
pool = []
for i in range(8):                                            # 8 batches of B volumes, revisited: something to learn
    z = torch.randn(B, 1, 16, 16, 16, device=dev, generator=g)
    pool.append(torch.nn.functional.interpolate(z, size=(128,) * 3, mode="trilinear", align_corners=False) + 0.1 * torch.randn(B, 1, 128, 128, 128, device=dev, generator=g))
lrs = M.linear_warmup_cosine_lrs(steps, base_lr=3e-4, warmup=20)
hist, t0 = [], time.perf_counter()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
ev[0].record()
for s in range(steps):
    tr.set_epoch(min(99, s * 100 // steps))
    out = tr.step(pool[s % len(pool)], epoch=min(99, s * 100 // steps), lr=lrs[s])
    ev[s + 1].record()
    hist.append(out["loss"])
torch.cuda.synchronize()
wall = time.perf_counter() - t0
loss = torch.cat(hist).float().cpu()
ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(steps)]
assert torch.isfinite(loss).all(), "non-finite loss"
print(f"STUNet-B 128^3 bf16, B={B}, {steps} AnatoMask steps over {len(pool)} revisited batches, lr warm-up + cosine to 3e-4: wall {wall:.1f} s")
for s in range(0, steps, 25):
    print(f"step {s:4d}: loss {loss[s:s + 25].mean():.4f} (mean of 25)   step time {sum(ms[s:s + 25]) / len(ms[s:s + 25]):7.2f} ms")
st = torch.cuda.memory_stats(dev)
print(f"first 25 / last 25 steps: loss {loss[:25].mean():.4f} -> {loss[-25:].mean():.4f}; step time {sum(ms[5:30]) / 25:.2f} -> {sum(ms[-25:]) / 25:.2f} ms; "
      f"reserved {torch.cuda.memory_reserved(dev) / 2**30:.1f} GiB, allocation retries {st.get('num_alloc_retries', 0)}, gradient norm of the last step {out['grad_norm'].item():.4f}")
assert loss[-25:].mean() < 0.9 * loss[:25].mean(), "the loss did not come down"
