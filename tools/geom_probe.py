"""Memory-only time of a tile pass as a function of WHICH index bits are tile bits (n=30, 9 high bits + bits 0..2).
Writes gpurun_out/geom_probe.csv: the bit set and the pass time with the blocks skipped."""
import sys
sys.path.insert(0, '.')
import numpy as np
from gpu_quantum_simulator_amd import Circuit, Simulator

n, L, H = 30, 3, 9
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 120
rows = []
with Simulator(n, fuse=3, profile=True, debug_skip_ops=1) as sim:
    for it in range(N):
        qs = sorted(int(x) for x in rng.choice(np.arange(L, n), H, replace=False))
        others = [q for q in range(L, n) if q not in qs][:H]
        c = Circuit.empty(n)
        h = np.array([[1, 1], [1, -1]]) / np.sqrt(2)
        for q in others: c.append_1q(h, q)   # pass 1 (generates the state): its 9 slots are taken by `others`
        for q in qs: c.append_1q(h, q)       # pass 2 = the probe: tile bits = qs
        sim.reset(); sim.run(c); sim.flush(); sim.sync(); sim.reset_stats()
        sim.reset(); sim.run(c); sim.flush(); sim.sync()
        log = sim.launch_log()
        k, nops, hm, ms = log[-1]
        bits = [b for b in range(40) if hm >> b & 1]
        rows.append((bits, ms, len(log)))
        print(it, bits, f"{ms:.3f}", len(log), flush=True)
with open("gpurun_out/geom_probe.csv", "w") as f:
    for bits, ms, nl in rows:
        f.write(" ".join(map(str, bits)) + f",{ms:.4f},{nl}\n")
