"""ms per step (n qubits, 1000 gates) over several circuits: as scheduled by default, and after the planning step's schedule
choice (qsim_choose_schedule: a few dozen scheduler settings ranked by the pass-time model).  Usage: plan_sweep.py [n] [ncircuits]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from gpu_quantum_simulator_amd import Circuit, Simulator, circuits, _lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
k = int(sys.argv[2]) if len(sys.argv) > 2 else 10
seeds = [20240117 + n] + list(range(1, k))
tot = [0.0, 0.0, 0.0]
with Simulator(n, profile=True) as sim:
    for s in seeds:
        c = Circuit.from_gates(n, circuits.random_gates(n, 1000, s, "all"))
        row = []
        for planned in (0, 1, 2):
            _lib.load().qsim_tune_table_clear()
            t_plan = 0.0
            if planned == 1:
                t0 = time.perf_counter(); sim.choose_schedule(c); t_plan = time.perf_counter() - t0
            if planned == 2:
                t0 = time.perf_counter(); sim.choose_schedule(c); sim.tune(c, 48, 8000.0); t_plan = time.perf_counter() - t0
            def body():
                sim.reset(); sim.run(c); sim.sync()
            body()
            sim.reset_stats()
            t0 = time.perf_counter(); body(); body(); dt = (time.perf_counter() - t0) / 2
            st = sim.stats()
            nb = sum(o for kk, o, hm, ms in sim.launch_log() if kk == "tile") / 2
            row.append(f"{dt*1e3:6.1f} ms {st['launches']//2:2d}p {nb:.0f}b" + (f" (planning {t_plan:.2f} s)" if planned else ""))
            tot[planned] += dt
        print(f"seed {s:9d}: default {row[0]}   model {row[1]}   measured {row[2]}", flush=True)
print(f"total: default {tot[0]*1e3:.1f} ms, schedule chosen by the model {tot[1]*1e3:.1f} ms, measured choice + tile-bit orders {tot[2]*1e3:.1f} ms")
