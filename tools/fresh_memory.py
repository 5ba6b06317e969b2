"""Per-pass time of the first run on a freshly allocated state against the second run (same schedule): what freshly allocated
memory costs a cold run.  python tools/fresh_memory.py [n]"""
import sys, time
sys.path.insert(0, '.')
from gpu_quantum_simulator_amd import Circuit, Simulator, circuits
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
c = Circuit.from_gates(n, circuits.random_gates(n, 1000, 20240117 + n, "all"))
with Simulator(8, fuse=3) as warm:  # loads the code object, creates queues: not what is measured here
    warm.run(Circuit.from_gates(8, circuits.random_gates(8, 50, 1, "all"))); warm.sync()
with Simulator(n, fuse=3, pingpong=0, profile=True) as sim:
    logs = []
    for rep in range(3):
        sim.reset(); sim.reset_stats()
        t0 = time.perf_counter(); sim.run(c); sim.flush(); sim.sync(); dt = time.perf_counter() - t0
        logs.append((dt, sim.launch_log()))
    for i in range(len(logs[0][1])):
        print(" ".join(f"{lg[1][i][3]:7.3f}" for lg in logs), " blocks", logs[0][1][i][1])
    print("wall ms:", " ".join(f"{1e3*lg[0]:.1f}" for lg in logs), " kernel ms:", " ".join(f"{sum(x[3] for x in lg[1]):.1f}" for lg in logs))
