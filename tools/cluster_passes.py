"""Per-shard launch log of a virtual cluster (P shards on ONE device): which kernels every shard runs between exchanges
and how long each takes.  Usage: python tools/cluster_passes.py [n] [P]"""
import sys
from ctypes import byref, c_double, c_int, c_uint64, c_void_p
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from gpu_quantum_simulator_amd import Circuit, Cluster, circuits, _lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
P = int(sys.argv[2]) if len(sys.argv) > 2 else 8
c = Circuit.from_gates(n, circuits.random_gates(n, 1000, 20240117 + n, "all"))
lib = _lib.load()
lib.qsim_cluster_shard.restype = c_void_p
with Cluster(n, P, devices=[0] * P, profile=1) as cl:
    cl.plan(c)
    cl.run(c)
    shards = [c_void_p(lib.qsim_cluster_shard(cl._h, r)) for r in range(P)]
    for s in shards:
        lib.qsim_reset_stats(s)
    cl.run(c)
    tot = 0.0
    for r, s in enumerate(shards):
        cnt = lib.qsim_launch_log(s, -1, None, None, None, None)
        line, sub = [], 0.0
        for i in range(cnt):
            k, o, hm, ms = c_int(), c_int(), c_uint64(), c_double()
            lib.qsim_launch_log(s, i, byref(k), byref(o), byref(hm), byref(ms))
            vis = c_double()
            lib.qsim_launch_log_visited(s, i, byref(vis))
            line.append(f"{_lib.K_NAMES[k.value]}:{o.value}:{ms.value:.2f}" + (f"@{vis.value:.3g}" if vis.value != 1.0 else ""))
            sub += ms.value
        tot += sub
        print(f"shard {r}: {sub:7.2f} ms  " + " ".join(line), flush=True)
    print(f"sum of kernel times over shards: {tot:.2f} ms", flush=True)
