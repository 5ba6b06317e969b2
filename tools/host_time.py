import sys, time
sys.path.insert(0, '.')
from gpu_quantum_simulator_amd import Circuit, Simulator, circuits, _lib
n = 30
c = Circuit.from_gates(n, circuits.random_gates(n, 1000, 20240117 + n, "all"))
with Simulator(n, fuse=3, pingpong=0, plan_cache=0) as sim:
    for rep in range(4):
        sim.reset()
        t0 = time.perf_counter(); sim.run_nosync(c) if hasattr(sim, "run_nosync") else None
        lib = _lib.load()
        lib.qsim_run_circuit(sim._h, c._h, 0, -1)
        t1 = time.perf_counter()
        lib.qsim_flush(sim._h)
        t2 = time.perf_counter()
        sim.sync()
        t3 = time.perf_counter()
        print(f"rep {rep}: enqueue {1e3*(t1-t0):.1f} ms, flush (schedule+launch) {1e3*(t2-t1):.1f} ms, wait {1e3*(t3-t2):.1f} ms", flush=True)
