"""Steady-state step time of the bench circuit against the number of model-ranked schedules qsim_tune_circuit runs (QSIM_TUNE_SCHEDULES):
one child process per setting (the measured choice is kept per process).  Usage: python tools/sched_tries.py [n] [counts,...]"""
import os, subprocess, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    from gpu_quantum_simulator_amd import Circuit, Simulator, circuits
    n = int(sys.argv[2])
    seed = int(sys.argv[3])
    c = Circuit.from_gates(n, circuits.random_gates(n, 1000, seed, "all"))
    with Simulator(n) as sim:
        t0 = time.time()
        rep = sim.tune(c, 48, 8000.0)
        t_plan = time.time() - t0
        for _ in range(3):
            sim.reset(); sim.run(c); sim.sync()
        t0 = time.time()
        for _ in range(10):
            sim.reset(); sim.run(c); sim.sync()
        print(f"schedules {os.environ.get('QSIM_TUNE_SCHEDULES', '4'):>3} seed {seed}: {(time.time() - t0) * 100:.2f} ms/step, planning {t_plan:.1f} s", flush=True)
    sys.exit(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
counts = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "1,4,8,16,32").split(",")]
for seed in (20240117 + n, 1, 2):
    for k in counts:
        env = dict(os.environ, QSIM_TUNE_SCHEDULES=str(k))
        subprocess.run([sys.executable, os.path.abspath(__file__), "child", str(n), str(seed)], env=env, check=False)
