import os, sys, torch
sys.path.insert(0, os.getcwd())
from anatomask_amd import ops
dev="cuda:0"; B,C,S=16,64,128
w = torch.randn(C, C, 3, 3, 3, device=dev) * 0.02
wp = ops.pack_weight(w, torch.bfloat16, False, False)
fl = 2.0 * B * S ** 3 * C * C * 27
def timed(fn, iters=10):
    fn(); fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / iters
base = torch.randn(B, S, S, S, C, device=dev)
y = torch.empty(B, S, S, S, C, device=dev, dtype=torch.bfloat16)
for rep in range(2):
    for name, x in (("randn", base), ("lrelu(randn)", torch.nn.functional.leaky_relu(base, 0.01)), ("randn*1e-3", base * 1e-3), ("zeros", base * 0), ("ones", base * 0 + 1), ("relu(randn)", torch.relu(base))):
        xb = x.to(torch.bfloat16)
        t = timed(lambda: ops.conv3d(ops.CONV_FWD, xb, wp, None, (S, S, S), 3, 1, out=y, want_partials=True))
        print(f"{name:14s} {t:.3f} ms {fl / t / 1e9:.0f} TF", flush=True)
        del xb
