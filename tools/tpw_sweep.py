"""Tiles per workgroup of k_tile (QSIM_OPT_GRID_CAP caps the grid: tiles per workgroup = tiles / cap; 0 = the default: 64 where that leaves 2048 workgroups, down to 8 otherwise; 8 until late in round 4) against the steady-state
step of the n = 30 bench circuit, planning as bench.py does it; one process, alternating settings.  Usage: python tools/tpw_sweep.py [n]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from gpu_quantum_simulator_amd import Circuit, Simulator, circuits, _lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
c = Circuit.from_gates(n, circuits.random_gates(n, 1000, 20240117 + n, "all"))
ntiles = 1 << (n - 12)
with Simulator(n) as sim:
    sim.tune(c, 48, 8000.0)
    for rep in range(2):
        for tpw in ({24: (0, 16, 8, 4, 2), 25: (0, 16, 8, 4, 2), 26: (0, 32, 16, 8, 4), 28: (0, 8, 16, 32, 64), 30: (0, 8, 32, 64, 128), 32: (0, 8, 64, 128, 256, 512)}.get(n, (0, 8, 16, 32, 64))):
            sim.set_option(_lib.OPT_GRID_CAP, ntiles // tpw if tpw else 0)
            for _ in range(2):
                sim.reset(); sim.run(c); sim.sync()
            t0 = time.time()
            steps = (10 if n <= 30 else 4) * (8 if n <= 26 else 1)
            for _ in range(steps):
                sim.reset(); sim.run(c); sim.sync()
            print(f"n {n} tiles per workgroup {tpw:3d}: {(time.time() - t0) * 1e3 / steps:.2f} ms/step", flush=True)
