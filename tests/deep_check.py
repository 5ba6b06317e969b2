"""Very deep circuits (tens of thousands of gate statements) on a small register, single state and virtual shards, against
the oracle: plan-cache limits, scheduler windows and the shard planner's search cut-off (20 000 gates) all lie below this depth.
Run by hand on the GPU box (it calls the oracle, hence under tests/; not collected by pytest).  Usage: python tests/deep_check.py [n] [depth]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from gpu_quantum_simulator_amd import Circuit, Cluster, Simulator, circuits
from oracle import oracle  # the checker
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 30000
path = circuits.random_circuit_file(f"/tmp/deep_{n}_{depth}.qasm", n, depth, 4711, "all")
t0 = time.time()
_, want, _, _ = oracle.run_qasm(path)
print(f"oracle {time.time() - t0:.1f} s", flush=True)
c = Circuit.from_file(path)
worst = 0.0
for fuse in (3, 0):
    with Simulator(n, fuse=fuse) as sim:
        for rep in range(2):
            sim.reset(); sim.run(c); sim.flush()
            err = float(np.max(np.abs(sim.read() - want)))
            worst = max(worst, err)
            print(f"single fuse {fuse} run {rep}: max error {err:.3g} launches {sim.stats()['launches']}", flush=True)
for P in (2, 4, 8):
    for plan in (False, True):
        with Cluster(n, P, devices=[0] * P) as cl:
            if plan:
                cl.plan(c)
            cl.run(c)
            err = float(np.max(np.abs(cl.read() - want)))
            worst = max(worst, err)
            print(f"{P} shards planned={plan}: max error {err:.3g} exchanges {cl.exchange_stats()[0]} re-layouts fused/separate {cl.pack_counts()}", flush=True)
assert worst < 1e-10, worst
print("deep check ok", worst)
