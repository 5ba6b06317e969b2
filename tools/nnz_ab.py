"""Steady-state step time with merged blocks of at most 4 against at most 8 entries per row (QSIM_SCHED_NNZ; 8 is the default since late
round 4): one child process per (circuit seed, setting), planning as bench.py does it; the launch log's block forms say how many blocks
a step has and how many of them are wide.  Usage: python tools/nnz_ab.py [n] [precision]"""
import os, subprocess, sys
HERE = os.path.abspath(__file__)
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import time
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), ".."))
    from gpu_quantum_simulator_amd import Circuit, Simulator, circuits
    n, seed, precision = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    c = Circuit.from_gates(n, circuits.random_gates(n, 1000, seed, "all"))
    with Simulator(n, profile=2, precision=precision) as sim:
        sim.tune(c, 48, 8000.0)
        for _ in range(3):
            sim.reset(); sim.run(c); sim.sync()
        sim.reset_stats()
        sim.reset(); sim.run(c); sim.sync()
        forms = [f for f in sim.launch_log_blocks() if f]
        blocks = sum(len(f) for f in forms)
        wide = sum(1 for f in forms for b in f if b[0] == 8)
        sim.set_option(2, 0)  # QSIM_OPT_PROFILE off: no events, no host work per launch in the timed steps
        for _ in range(2):
            sim.reset(); sim.run(c); sim.sync()
        t0 = time.time()
        for _ in range(10):
            sim.reset(); sim.run(c); sim.sync()
        print(f"n {n} fp{precision} seed {seed} nnz {os.environ.get('QSIM_SCHED_NNZ', '8')}: {(time.time() - t0) * 100:.2f} ms/step, {len(forms)} passes, {blocks} blocks, {wide} wide", flush=True)
    sys.exit(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
precision = int(sys.argv[2]) if len(sys.argv) > 2 else 64
for seed in (20240117 + n, 1, 2, 3, 4, 5):
    for nnz in ("4", "8"):
        env = dict(os.environ, QSIM_SCHED_NNZ=nnz)
        subprocess.run([sys.executable, HERE, "child", str(n), str(seed), str(precision)], env=env, check=False)
