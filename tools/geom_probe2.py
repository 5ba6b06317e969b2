"""Memory-only time of a tile pass as a function of WHICH index bits are tile bits (n=30, bits 0..2 + 9 high bits), blocks
skipped.  The second pass of every probe circuit is logged with its high-bit mask: gpurun_out/geom_probe2.csv."""
import os
import sys
os.environ["QSIM_SCHED_LOCAL"] = "0"      # the probe wants exactly the qubit sets it asks for
os.environ["QSIM_SCHED_LOOKAHEAD"] = "0"
sys.path.insert(0, '.')
import numpy as np
from gpu_quantum_simulator_amd import Circuit, Simulator

n, L, H = 30, 3, 9
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 7)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 300
h = np.array([[1, 1], [1, -1]]) / np.sqrt(2)
rows = []
with Simulator(n, fuse=3, profile=True, debug_skip_ops=1) as sim:
    for it in range(N):
        mode = it % 3
        if mode == 0:
            qs = sorted(int(x) for x in rng.choice(np.arange(L, n), H, replace=False))
        elif mode == 1:   # biased towards the upper half
            qs = sorted(int(x) for x in rng.choice(np.arange(12, n), H, replace=False))
        else:             # a few of the suspicious bits 20..23 plus random others
            k = int(rng.integers(0, 4))
            a = [int(x) for x in rng.choice(np.arange(20, 24), k, replace=False)]
            rest = [q for q in range(L, n) if not 20 <= q <= 23]
            qs = sorted(a + [int(x) for x in rng.choice(rest, H - k, replace=False)])
        others = [q for q in range(L, n) if q not in qs]
        rng.shuffle(others)
        others = sorted(int(x) for x in others[:H])
        c = Circuit.empty(n)
        for q in others: c.append_1q(h, q)   # pass 1 (generates the state)
        for q in qs: c.append_1q(h, q)       # pass 2
        sim.reset(); sim.run(c); sim.flush(); sim.sync(); sim.reset_stats()
        sim.reset(); sim.run(c); sim.flush(); sim.sync()
        for j, (k_, nops, hm, ms) in enumerate(sim.launch_log()):
            bits = [b for b in range(40) if hm >> b & 1]
            if k_ == "tile" and len(bits) == H and j > 0:   # j == 0 only writes
                rows.append((bits, ms))
        if it % 25 == 0:
            print(it, len(rows), flush=True)
os.makedirs("gpurun_out", exist_ok=True)
with open("gpurun_out/geom_probe2.csv", "w") as f:
    for bits, ms in rows:
        f.write(" ".join(map(str, bits)) + f",{ms:.4f}\n")
print("samples", len(rows))
