"""Child script for tests/test_distributed_cpu.py::test_launcher_*: one rank of a gloo job started by
gpu_quantum_simulator_amd.launch.spawn_ranks (the function bench.py --gpus N uses).  Prints one line from rank 0."""
import json
import os
import sys

import torch
import torch.distributed as dist


def main():
    dist.init_process_group("gloo")
    t = torch.tensor([float(dist.get_rank() + 1)], dtype=torch.float64)
    dist.all_reduce(t)
    if dist.get_rank() == 0:
        print(json.dumps({"metric": "probe", "world": dist.get_world_size(), "sum": float(t.item()),
                          "argv": sys.argv[1:], "local_rank": os.environ.get("LOCAL_RANK")}), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
