import sys, json
sys.path.insert(0, '.')
from gpu_quantum_simulator_amd import Circuit, Simulator, circuits
n = 30
c = Circuit.from_gates(n, circuits.random_gates(n, 1000, 20240117 + n, "all"))
cfgs = [(11, 4, 256), (11, 3, 256), (11, 2, 256), (11, 4, 512), (12, 4, 512), (12, 4, 1024), (12, 3, 512), (12, 2, 512), (12, 5, 512), (13, 4, 1024), (13, 3, 1024), (10, 3, 256), (10, 2, 256)]
for (B, L, T) in cfgs:
    with Simulator(n, fuse=3, profile=True, tile_bits=B, tile_low_bits=L, tile_threads=T, tile_max_ops=64) as sim:
        def body():
            sim.reset(); sim.run(c); sim.flush()
        body(); sim.sync(); sim.reset_stats()
        for _ in range(2): body()
        sim.sync()
        st = sim.stats()
        ms = sum(v["ms"] for k, v in st["kernels"].items() if k != "init") / 2
        nl = sum(v["launches"] for k, v in st["kernels"].items() if k != "init") / 2
        print(f"B={B} L={L} T={T}: {ms:8.2f} ms/iter, {nl:.0f} passes, {ms/nl:.2f} ms/pass -> {1000/((ms+3)*1e-3):.0f} gate-applies/s", flush=True)
