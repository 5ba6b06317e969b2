"""Training data for the pass-geometry cost model: memory-only time (blocks skipped) of ONE tile pass whose nine high tile
bits are exactly a chosen set, for many random sets out of index bits 3..n-1 (n=30, fp64, L=3).  The state is made dense
once; every sample is a circuit of nine h gates on the chosen qubits, run without a reset (one pass, tile = the set).
Writes gpurun_out/geom_probe3.csv: "<bits>,<ms>"."""
import os
import sys
os.environ["QSIM_SCHED_LOCAL"] = "0"
os.environ["QSIM_SCHED_LOOKAHEAD"] = "0"
sys.path.insert(0, '.')
import numpy as np
from gpu_quantum_simulator_amd import Circuit, Simulator, circuits

n, L, H = 30, 3, 9
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 11
N = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
rng = np.random.default_rng(seed)
h = np.array([[1, 1], [1, -1]]) / np.sqrt(2)
rows, bad = [], 0
with Simulator(n, fuse=3, profile=True) as sim:
    sim.run(Circuit.from_gates(n, circuits.random_gates(n, 200, 5, "all")))  # a dense state to move around
    sim.sync()
    sim.set_option(10, 1)  # QSIM_OPT_DEBUG_SKIP_OPS
    for it in range(N):
        mode = it % 4
        if mode == 0:
            qs = rng.choice(np.arange(L, n), H, replace=False)
        elif mode == 1:   # what schedules tend to look like: a few low bits, the rest spread
            lo = rng.choice(np.arange(L, 14), int(rng.integers(1, 5)), replace=False)
            qs = np.concatenate([lo, rng.choice(np.arange(14, n), H - len(lo), replace=False)])
        elif mode == 2:   # upper half only
            qs = rng.choice(np.arange(12, n), H, replace=False)
        else:             # clustered: a contiguous run plus random others
            start = int(rng.integers(L, n - 4)); run = np.arange(start, min(n, start + int(rng.integers(2, 6))))
            rest = np.array([q for q in range(L, n) if q not in run])
            qs = np.concatenate([run, rng.choice(rest, H - len(run), replace=False)])
        qs = sorted(int(x) for x in qs)
        c = Circuit.empty(n)
        for q in qs: c.append_1q(h, q)
        for rep in range(2):
            sim.reset_stats()
            sim.run(c); sim.flush(); sim.sync()
        log = sim.launch_log()
        tiles = [(hm, ms) for k_, nops, hm, ms in log if k_ == "tile"]
        want = sum(1 << q for q in qs)
        if len(tiles) == 1 and tiles[0][0] == want:
            rows.append((qs, tiles[0][1]))
        else:
            bad += 1
        if it % 250 == 0:
            print(it, len(rows), bad, flush=True)
os.makedirs("gpurun_out", exist_ok=True)
with open(f"gpurun_out/geom_probe3_{seed}.csv", "w") as f:
    for bits, ms in rows:
        f.write(" ".join(map(str, bits)) + f",{ms:.4f}\n")
print("samples", len(rows), "rejected", bad)
