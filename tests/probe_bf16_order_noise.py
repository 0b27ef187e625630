"""Not a test (pytest does not collect it): how much of the bf16-storage distance from fp32 is SUMMATION-ORDER noise?

    python tools/with_lib.py build_ab/libanatomask_hip_ablate.so tests/probe_bf16_order_noise.py TAG      (tools build; the AM_* switches
    select older kernels for single layers = the same function in another summation order)

Runs the step of test_config2_stunet_b_128_bf16_step_vs_oracle, prints the per-tensor distances from the fp32 oracle for the worst tensors and
saves every gradient tensor to /tmp/order_noise_TAG.pt; with two or more TAGs on the command line after `cmp` it prints the pairwise
HIP-vs-HIP distances instead."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
STEM = "sparse_encoder.sp_cnn.conv_blocks_context.0.0.conv1.weight"


def main():
    if sys.argv[1] == "cmp":
        G = {t: torch.load(f"/tmp/order_noise_{t}.pt") for t in sys.argv[2:]}
        tags = list(G)
        for i, a in enumerate(tags):
            for b in tags[i + 1:]:
                rel = {k: float((G[a][k] - G[b][k]).norm() / (G[b][k].norm() + 1e-300)) for k in G[a] if k != "_fp32"}
                print(f"HIP[{a}] vs HIP[{b}]: rel-L2 median {np.median(list(rel.values())):.3e}, stem weight {rel[STEM]:.3e}, max {max(rel.values()):.3e}")
        return
    from tests import test_configs_gpu as T
    from anatomask_amd.trainer import AnatoMaskTrainer
    cfg, W0, x, mask1, keys, o = T._oracle_step_b128()
    m = T._build(cfg, W0)
    tr = AnatoMaskTrainer(m, lr=1e-4, ema_decay=0.999, total_epochs=1000, distributed=False)
    tr.step(x.to(T.DEV), epoch=0, mask1=mask1, keys=keys)
    rows = T._per_tensor_errors(m, o["grads"], f"[{sys.argv[1]}] bf16 storage vs fp32 oracle")
    stem = [r for r in rows if r[0] == STEM][0]
    print(f"[{sys.argv[1]}] stem weight rel {stem[2]:.4f} 1-cos {1 - stem[3]:.4f}")
    torch.save({r[0]: m._G[r[0]].detach().double().cpu().reshape(-1) for r in rows}, f"/tmp/order_noise_{sys.argv[1]}.pt")
    if len(sys.argv) > 2 and sys.argv[2] == "table":           # per tensor: HIP vs fp32, ideal emulation vs fp32, HIP vs emulation
        emu = T._emulated_bf16_grads(cfg, W0, x, o["mask"])
        for k, n_el, rel, cos in rows:
            w, a = o["grads"][k].double().reshape(-1), emu[k].double().reshape(-1)
            g = m._G[k].detach().double().cpu().reshape(-1)
            print(f"[tab] {k.replace('sparse_encoder.sp_cnn.conv_blocks_context', 'enc').replace('dense_decoder.', 'dec.'):44s} {n_el:9d}  hip {rel:.3f}  emu {float((a - w).norm() / w.norm()):.3f}  "
                  f"hip-emu {float((g - a).norm() / a.norm()):.3f}  |g|/|w| {float(g.norm() / w.norm()):.3f} |a|/|w| {float(a.norm() / w.norm()):.3f}")


if __name__ == "__main__":
    main()
