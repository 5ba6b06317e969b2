"""One virtual-cluster configuration, a few iterations (for kernel traces).  Usage: python tools/cluster_one.py n P iters"""
import sys, time
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from gpu_quantum_simulator_amd import Circuit, Cluster, circuits
n, P, iters = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
c = Circuit.from_gates(n, circuits.random_gates(n, 1000, 20240117 + n, "all"))
with Cluster(n, P, devices=[0] * P) as cl:
    cl.plan(c)
    cl.run(c)
    t0 = time.perf_counter()
    for _ in range(iters):
        cl.run(c)
    dt = (time.perf_counter() - t0) / iters
    print(f"{P} virtual: {dt*1e3:.2f} ms/iter  packs fused/separate = {cl.pack_counts()}", flush=True)
