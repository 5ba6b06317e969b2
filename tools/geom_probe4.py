"""What the ORDER of a pass's high tile bits is worth (n=30, fp64, L=3, nine high bits, blocks skipped).
Part A: a few fixed sets (the bench schedule's passes), many random orders each -> spread per set.
Part B: random sets x random orders -> gpurun_out/geom_probe4_<seed>.csv ("<bits in tile-local order>,<ms>") for a cost model."""
import os
import sys
os.environ["QSIM_SCHED_LOCAL"] = "0"
os.environ["QSIM_SCHED_LOOKAHEAD"] = "0"
sys.path.insert(0, '.')
import numpy as np
from gpu_quantum_simulator_amd import Circuit, Simulator, circuits, _lib

n, L, H = 30, 3, 9
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 21
NB = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
NA = int(sys.argv[3]) if len(sys.argv) > 3 else 48
rng = np.random.default_rng(seed)
h = np.array([[1, 1], [1, -1]]) / np.sqrt(2)
FIXED = [[3, 6, 10, 13, 16, 24, 25, 27, 29], [4, 5, 7, 8, 9, 12, 19, 23, 24], [3, 8, 12, 13, 15, 17, 18, 20, 26],
         [3, 9, 11, 13, 15, 16, 17, 21, 26], [8, 16, 19, 22, 25, 26, 27, 28, 29], [3, 9, 10, 11, 14, 15, 20, 21, 23],
         [4, 5, 8, 12, 15, 17, 21, 26, 27], [6, 7, 9, 16, 18, 20, 23, 26, 29], [17, 18, 23, 24, 25, 26, 27, 28, 29],
         [13, 14, 15, 19, 20, 21, 22, 27, 28]]


def one_pass(sim, qs):
    c = Circuit.empty(n)
    for q in qs: c.append_1q(h, q)
    for rep in range(2):
        sim.reset_stats()
        sim.run(c); sim.flush(); sim.sync()
    log, orders = sim.launch_log(), sim.launch_log_orders()
    tiles = [(o, ms) for (k_, nops, hm, ms), o in zip(log, orders) if k_ == "tile"]
    if len(tiles) != 1 or sorted(tiles[0][0]) != sorted(qs):
        return None
    return tiles[0]


with Simulator(n, fuse=3, profile=True) as sim:
    sim.run(Circuit.from_gates(n, circuits.random_gates(n, 200, 5, "all")))
    sim.sync()
    sim.set_option(_lib.OPT_DEBUG_SKIP_OPS, 1)
    print("part A: fixed sets, ascending order first, then random orders", flush=True)
    for qs in FIXED:
        sim.set_option(_lib.OPT_DEBUG_TILE_ORDER, 0)
        base = one_pass(sim, qs)
        res = []
        for k in range(1, NA + 1):
            sim.set_option(_lib.OPT_DEBUG_TILE_ORDER, 1000 * seed + k)
            r = one_pass(sim, qs)
            if r: res.append(r)
        res.sort(key=lambda t: t[1])
        ms = np.array([t[1] for t in res])
        print(f"set {qs}: ascending {base[1]:.3f} | random orders min {ms.min():.3f} p25 {np.percentile(ms,25):.3f} median {np.median(ms):.3f} max {ms.max():.3f}", flush=True)
        print("   best orders:", [(o, round(t, 3)) for o, t in res[:3]], flush=True)
        print("   worst orders:", [(o, round(t, 3)) for o, t in res[-2:]], flush=True)
    rows = []
    for it in range(NB):
        mode = it % 3
        if mode == 0:
            qs = rng.choice(np.arange(L, n), H, replace=False)
        elif mode == 1:
            lo = rng.choice(np.arange(L, 14), int(rng.integers(1, 6)), replace=False)
            qs = np.concatenate([lo, rng.choice(np.arange(14, n), H - len(lo), replace=False)])
        else:
            qs = rng.choice(np.arange(10, n), H, replace=False)
        qs = sorted(int(x) for x in qs)
        sim.set_option(_lib.OPT_DEBUG_TILE_ORDER, 0 if it % 5 == 0 else 7 + it)
        r = one_pass(sim, qs)
        if r: rows.append(r)
        if it % 500 == 0: print("part B", it, len(rows), flush=True)
os.makedirs("gpurun_out", exist_ok=True)
with open(f"gpurun_out/geom_probe4_{seed}.csv", "w") as f:
    for order, ms in rows:
        f.write(" ".join(map(str, order)) + f",{ms:.4f}\n")
print("samples", len(rows))
