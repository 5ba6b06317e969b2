import sys, time
sys.path.insert(0, '.')
from gpu_quantum_simulator_amd import Circuit, Simulator, circuits
for n in (10, 12, 14, 16, 18, 20, 22):
    c = Circuit.from_gates(n, circuits.random_gates(n, 1000, 20240117 + n, "all"))
    for opts in ({}, {"tile_bits": 10}, {"tile_bits": 8, "tile_low_bits": 3}, {"fuse": 2}):
        o = dict(opts); fuse = o.pop("fuse", 3)
        with Simulator(n, fuse=fuse, **o) as sim:
            def body():
                sim.reset(); sim.run(c); sim.sync()
            body()
            t0 = time.perf_counter()
            for _ in range(5): body()
            dt = (time.perf_counter() - t0) / 5
            print(f"n={n:2d} {str(opts):40s} {dt*1e3:8.3f} ms/iter  {1000/dt:10.0f} gate-applies/s  launches={sim.stats()['launches']//6}", flush=True)
