"""One process per GPU: starts the ranks of a multi-GPU run as CHILD processes of a parent that never touches the GPU.

`python bench.py --gpus N` (N > 1, no WORLD_SIZE in the environment) goes through `spawn_ranks`: the parent has not
imported torch or loaded libqsim at that point, it starts `python -m torch.distributed.run` (which in turn starts the N
ranks), relays their output and exits with their code.  Nothing here replaces a running program (no exec): on this
platform a process that has initialised the GPU must not exec, and the simplest way to be sure is never to do it.
"""
from __future__ import annotations

import glob
import os
import socket
import subprocess
import sys
from typing import List, Optional, Sequence, Tuple


def free_port() -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def count_gpus() -> Optional[int]:
    """GPUs the kernel driver exposes, read from sysfs (no HIP call, no torch): KFD topology nodes with SIMDs.
    None when the topology cannot be read (then the ranks themselves decide)."""
    nodes = glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties")
    if not nodes:
        return None
    count = 0
    for path in nodes:
        try:
            with open(path) as f:
                for line in f:
                    if line.startswith("simd_count"):
                        count += int(line.split()[1]) > 0
                        break
        except OSError:
            return None
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None and v.strip() != "":
            count = min(count, len([x for x in v.split(",") if x.strip() != ""]))
    return count


def rank_command(nproc: int, script: str, script_args: Sequence[str], port: Optional[int] = None) -> List[str]:
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}",
            "--master-addr", "127.0.0.1", "--master-port", str(port or free_port()), script, *script_args]


def spawn_ranks(nproc: int, script: str, script_args: Sequence[str], env: Optional[dict] = None,
                timeout: Optional[float] = None) -> Tuple[int, str]:
    """Runs `script` as nproc ranks (children of this process).  Returns (exit code, the children's stdout); their
    stderr goes straight to ours.  The caller has not touched the GPU and does not afterwards."""
    cmd = rank_command(nproc, script, script_args)
    child_env = dict(os.environ if env is None else env)
    child_env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL needs between processes on this image
    try:
        p = subprocess.run(cmd, stdout=subprocess.PIPE, text=True, env=child_env, timeout=timeout)
    except subprocess.TimeoutExpired as e:
        return 124, (e.stdout or "") if isinstance(e.stdout, str) else ""
    return p.returncode, p.stdout
