"""ms per step (n qubits, 1000 gates) under scheduler overrides, several circuits; no planning step.  Usage: cap_sweep.py [n]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from gpu_quantum_simulator_amd import Circuit, Simulator, circuits
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
seeds = [20240117 + n, 1, 2, 3, 4]
circs = [Circuit.from_gates(n, circuits.random_gates(n, 1000, s, "all")) for s in seeds]
variants = [{}, {"QSIM_SCHED_CAP": "28"}, {"QSIM_SCHED_CAP": "32"}, {"QSIM_SCHED_CAP": "40"}, {"QSIM_SCHED_NOCOMMUTE": "1"},
            {"QSIM_SCHED_CAP": "32", "QSIM_SCHED_NOCOMMUTE": "1"}, {"QSIM_SCHED_ROLLOUT": "4"}, {"QSIM_SCHED_CAP": "32", "QSIM_SCHED_ROLLOUT": "4"},
            {"QSIM_SCHED_CHEAP": "2"}, {"QSIM_SCHED_CAP": "32", "QSIM_SCHED_CHEAP": "2"}]
with Simulator(n, profile=True) as sim:
    for env in variants:
        for k in list(os.environ):
            if k.startswith("QSIM_SCHED_"):
                del os.environ[k]
        os.environ.update(env)
        row, tot = [], 0.0
        for c in circs:
            def body():
                sim.reset(); sim.run(c); sim.sync()
            body()
            sim.reset_stats()
            t0 = time.perf_counter(); body(); body(); dt = (time.perf_counter() - t0) / 2
            st = sim.stats()
            log = sim.launch_log()
            nb = sum(o for k, o, hm, ms in log if k == "tile") / 2
            row.append(f"{dt*1e3:6.1f}ms/{st['launches']//2}p/{nb:.0f}b")
            tot += dt
        print(f"{str(env):60s} total {tot*1e3:7.1f} ms  " + " ".join(row), flush=True)
