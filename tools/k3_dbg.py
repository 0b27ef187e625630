import os, sys, torch
sys.path.insert(0, "/root/repo")
sys.argv = ["x", "none"]
exec(open("/root/repo/tools/k3_check.py").read().split("quick = len(sys.argv)")[0])
check(1, 64, 64, 64, 64, 64, bias=False)
check(1, 64, 64, 64, 64, 64, stats=True, bias=False)
check(2, 64, 64, 64, 64, 64, bias=False)
check(1, 64, 64, 64, 64, 64, fused=True)
check(1, 64, 64, 128, 64, 64, bias=False)
check(1, 64, 64, 64, 128, 64, bias=False)
check(1, 32, 64, 64, 64, 128, bias=False)
print(FAILED)
