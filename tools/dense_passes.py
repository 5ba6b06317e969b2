"""Launch log of the full-sweep comparison run (sparse start off, schedule and tile-bit orders planned for a dense start):
blocks and time of every pass.  Usage: python tools/dense_passes.py [n] [sparse 0/1]"""
import os, sys
from ctypes import byref, c_double, c_int, c_uint64
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from gpu_quantum_simulator_amd import Circuit, Simulator, circuits, _lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
sparse = int(sys.argv[2]) if len(sys.argv) > 2 else 0
c = Circuit.from_gates(n, circuits.random_gates(n, 1000, 20240117 + n, "all"))
lib = _lib.load()
with Simulator(n, fuse=3, profile=True, sparse_start=sparse) as sim:
    sim.choose_schedule(c)
    rep = sim.tune(c, 32, 6000.0, dense_start=not sparse)
    print("planning", {k: rep[k] for k in ("seconds", "tile_passes", "candidates_timed")}, flush=True)
    for _ in range(2):
        sim.reset(); sim.run(c); sim.flush(); sim.sync()
    sim.reset_stats()
    sim.reset(); sim.run(c); sim.flush(); sim.sync()
    cnt = lib.qsim_launch_log(sim._h, -1, None, None, None, None)
    tot = 0.0
    for i in range(cnt):
        k, o, hm, ms, vis = c_int(), c_int(), c_uint64(), c_double(), c_double()
        lib.qsim_launch_log(sim._h, i, byref(k), byref(o), byref(hm), byref(ms))
        lib.qsim_launch_log_visited(sim._h, i, byref(vis))
        tot += ms.value
        print(f"{_lib.K_NAMES[k.value]:6s} blocks={o.value:2d} visited={vis.value:.3g} {ms.value:7.3f} ms  high={[b for b in range(n) if hm.value >> b & 1]}", flush=True)
    print(f"total {tot:.2f} ms over {cnt} launches", flush=True)
