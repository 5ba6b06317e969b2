"""(visited, blocks, ms) of every tile pass over several circuits and scheduler variants: data for the pass-time model of the planning step."""
import os, sys
from ctypes import byref, c_double, c_int, c_ubyte
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from gpu_quantum_simulator_amd import Circuit, Simulator, circuits, _lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
precision = int(sys.argv[2]) if len(sys.argv) > 2 else 64
seeds = [20240117 + n, 1, 2, 3, 4, 5]
circs = [Circuit.from_gates(n, circuits.random_gates(n, 1000, s, "all")) for s in seeds]
circs.append(Circuit.from_gates(n, circuits.random_gates(n, 1000, 7, "clifford_t")))
variants = [{}, {"QSIM_SCHED_CAP": "28"}, {"QSIM_SCHED_CAP": "32"}, {"QSIM_SCHED_CAP": "40"}, {"QSIM_SCHED_CAP": "48"}, {"QSIM_SCHED_CAP": "16"},
            {"QSIM_SCHED_MERGEQ": "3"}, {"QSIM_SCHED_MERGEQ": "4"}, {"QSIM_SCHED_MERGEQ": "5"}, {"QSIM_SCHED_MERGE": "0"},
            {"QSIM_SCHED_MERGEQ": "4", "QSIM_SCHED_CAP": "48"}, {"QSIM_SCHED_MERGEQ": "5", "QSIM_SCHED_CAP": "24"}]
lib = _lib.load()
print("visited,blocks,ms,high_mask,forms")  # forms: one hex byte per block (qsim_launch_log_blocks)
with Simulator(n, profile=2, precision=precision) as sim:
    for env in variants:
        for k in list(os.environ):
            if k.startswith("QSIM_SCHED_"):
                del os.environ[k]
        os.environ.update(env)
        for c in circs:
            sim.reset(); sim.run(c); sim.sync()
            sim.reset_stats()
            sim.reset(); sim.run(c); sim.sync()
            for i, (k, o, hm, ms) in enumerate(sim.launch_log()):
                if k != "tile":
                    continue
                v = c_double()
                lib.qsim_launch_log_visited(sim._h, i, byref(v))
                codes, cnt = (c_ubyte * 64)(), c_int()
                lib.qsim_launch_log_blocks(sim._h, i, codes, 64, byref(cnt))
                forms = "".join(f"{codes[j]:02x}" for j in range(min(cnt.value, 64)))
                print(f"{v.value:.6g},{o},{ms:.4f},{hm:#x},{forms}", flush=True)
