"""Pairwise structure of the pass-geometry cost (n=30, fp64, L=3, blocks skipped): for every pair of index bits (a, b) and
every role group (lanes = tile-local bits 3..5, waves = 6..8, registers = 9..11) a few passes whose order puts a and b in
that group, the other seven tile bits drawn at random.  Orders are imposed through the geometry table
(qsim_tune_table_load).  Writes gpurun_out/geom_probe5.csv: "<bits in tile-local order>,<ms>,<role>,<a>,<b>"."""
import os
import sys
os.environ["QSIM_SCHED_LOCAL"] = "0"
os.environ["QSIM_SCHED_LOOKAHEAD"] = "0"
sys.path.insert(0, '.')
import itertools
import numpy as np
from gpu_quantum_simulator_amd import Circuit, Simulator, circuits, _lib

n, L, H = 30, 3, 9
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 31
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rng = np.random.default_rng(seed)
h = np.array([[1, 1], [1, -1]]) / np.sqrt(2)
samples, seen = [], set()
for role in range(3):
    for a, b in itertools.combinations(range(L, n), 2):
        for _ in range(reps):
            for attempt in range(20):
                rest = [q for q in range(L, n) if q not in (a, b)]
                others = [int(x) for x in rng.choice(rest, H - 2, replace=False)]
                key = tuple(sorted([a, b] + others))
                if key not in seen:
                    break
            seen.add(key)
            grp = [a, b, others[0]]
            rng.shuffle(grp)
            o = others[1:]
            groups = [o[0:3], o[3:6]]
            groups.insert(role, [int(x) for x in grp])
            order = groups[0] + groups[1] + groups[2]
            samples.append((order, role, a, b))
os.makedirs("gpurun_out", exist_ok=True)
wis = "gpurun_out/geom_probe5_orders.txt"
with open(wis, "w") as f:
    for order, role, a, b in samples:
        mask = sum(1 << q for q in order)
        f.write(f"{n} 0 12 3 {mask:x} 1.0 1.0 " + " ".join(map(str, order)) + "\n")
lib = _lib.load()
lib.qsim_tune_table_clear()
print("orders loaded:", lib.qsim_tune_table_load(wis.encode()), "of", len(samples), flush=True)
rows = []
with Simulator(n, fuse=3, profile=True) as sim:
    sim.run(Circuit.from_gates(n, circuits.random_gates(n, 200, 5, "all")))
    sim.sync()
    sim.set_option(_lib.OPT_DEBUG_SKIP_OPS, 1)
    for it, (order, role, a, b) in enumerate(samples):
        c = Circuit.empty(n)
        for q in sorted(order): c.append_1q(h, q)
        for rep in range(2):
            sim.reset_stats()
            sim.run(c); sim.flush(); sim.sync()
        log, orders = sim.launch_log(), sim.launch_log_orders()
        tiles = [(o, ms) for (k_, nops, hm, ms), o in zip(log, orders) if k_ == "tile"]
        if len(tiles) == 1 and tiles[0][0] == order:
            rows.append((order, tiles[0][1], role, a, b))
        if it % 500 == 0: print(it, len(rows), flush=True)
with open("gpurun_out/geom_probe5.csv", "w") as f:
    for order, ms, role, a, b in rows:
        f.write(" ".join(map(str, order)) + f",{ms:.4f},{role},{a},{b}\n")
print("samples", len(rows), "of", len(samples))
