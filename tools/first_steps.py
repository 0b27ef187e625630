"""Wall time of each of the first steps from a cold process (allocator growth, one-time kernel attribute calls): python tools/first_steps.py [batch]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anatomask_amd import modules as M  # noqa: E402
from anatomask_amd.trainer import AnatoMaskTrainer  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dev = torch.device("cuda:0")
kw = M.STUNET_CONFIGS["B"]
model = M.build_spark(kw["dims"], kw["depth"], kw["width"], (128,) * 3, 0.6, compute_dtype=torch.bfloat16).to(dev)
tr = AnatoMaskTrainer(model, lr=1e-4, total_epochs=1000, seed=1)
x = torch.randn(B, 1, 128, 128, 128, device=dev)
for i in range(10):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    tr.step(x, epoch=500)
    torch.cuda.synchronize()
    st = torch.cuda.memory_stats()
    print(f"step {i}: {(time.perf_counter() - t0) * 1e3:8.1f} ms  reserved {torch.cuda.memory_reserved() / 2**30:6.1f} GiB  allocated peak {torch.cuda.max_memory_allocated() / 2**30:6.1f} GiB  "
          f"alloc_retries {st.get('num_alloc_retries', 0)}  device mallocs {st.get('num_device_alloc', 0)} frees {st.get('num_device_free', 0)}", flush=True)
