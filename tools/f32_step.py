"""Planned step of the n = 30 bench circuit in one precision (for A/B runs with QSIM_LIB).  Usage: python tools/f32_step.py [32|64] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from gpu_quantum_simulator_amd import Circuit, Simulator, circuits
n = 30
precision = int(sys.argv[1]) if len(sys.argv) > 1 else 32
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 20240117 + n
c = Circuit.from_gates(n, circuits.random_gates(n, 1000, seed, "all"))
with Simulator(n, precision=precision) as sim:
    sim.tune(c, 48, 6000.0)
    for _ in range(3):
        sim.reset(); sim.run(c); sim.sync()
    t0 = time.time()
    for _ in range(10):
        sim.reset(); sim.run(c); sim.sync()
    print(f"{os.environ.get('QSIM_LIB', 'tree')}: fp{precision} seed {seed} {(time.time() - t0) * 100:.2f} ms/step", flush=True)
