"""Steady-state step of the n = 30 bench circuit with the package found under <root> (another build of this tree, e.g. a git
worktree copied to tools/ab/old): planning as bench.py does it, then 20 timed steps.  Usage: python tools/ab_step.py <root> [n]"""
import os, sys, time
root = os.path.abspath(sys.argv[1])
sys.path.insert(0, root)
from gpu_quantum_simulator_amd import Circuit, Simulator, circuits
import gpu_quantum_simulator_amd as pkg
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
c = Circuit.from_gates(n, circuits.random_gates(n, 1000, 20240117 + n, "all"))
with Simulator(n, profile=True) as sim:
    sim.tune(c, 48, 8000.0)
    for _ in range(3):
        sim.reset(); sim.run(c); sim.sync()
    t0 = time.time()
    for _ in range(20):
        sim.reset(); sim.run(c); sim.sync()
    print(f"{os.path.relpath(os.path.dirname(pkg.__file__))}: {(time.time() - t0) * 50:.2f} ms/step", flush=True)
