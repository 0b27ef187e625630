"""Self-launch of the one-process-per-GPU job (what `torchrun` does for P/pretrain_AnatoMask_DDP.py:192-240).

`python bench.py --gpus N` / `python -m anatomask_amd.pretrain --gpus N` without a launcher: the parent process -- which must not
have touched the GPU (a process that initialised HIP must never fork/exec GPU children on this pool) -- starts
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P <script> <args>` as a CHILD
process, forwards its stdout/stderr and exits with its return code.  Under a launcher (WORLD_SIZE set) this is a no-op.
"""
import os
import socket
import subprocess
import sys
from typing import List, Optional


def free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launched() -> bool:
    return "WORLD_SIZE" in os.environ and "RANK" in os.environ


def launch_command(n: int, script: str, argv: List[str], port: Optional[int] = None) -> List[str]:
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
            "--master-port", str(port or free_port()), script, *argv]


def self_launch(n: int, script: str, argv: List[str]) -> int:
    """Run `script argv` as n ranks; returns the job's exit code.  Call BEFORE any GPU call in this process."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS=os.environ.get("OMP_NUM_THREADS", "4"))
    return subprocess.run(launch_command(n, script, argv), env=env).returncode
