import sys
sys.path.insert(0, '.')
from gpu_quantum_simulator_amd import Circuit, Simulator, circuits
import os
n = int(os.environ.get('GEO_N', 30))
prec = int(os.environ.get('GEO_PRECISION', 64))
c = Circuit.from_gates(n, circuits.random_gates(n, 1000, 20240117 + n, "all"))
cfgs = [tuple(int(x) for x in a.split(',')) for a in sys.argv[1:]] or [(11, 4, 256, 0), (12, 4, 512, 0)]
for cfg in cfgs:
    B, L, T = cfg[:3]
    G = cfg[3] if len(cfg) > 3 else 0
    with Simulator(n, fuse=3, profile=True, tile_bits=B, tile_low_bits=L, tile_threads=T, tile_max_ops=int(os.environ.get('GEO_MAXOPS', 64)), grid_cap=max(G, 0), precision=prec) as sim:
        def body():
            sim.reset(); sim.run(c); sim.flush()
        body(); sim.sync(); sim.reset_stats()
        for _ in range(2): body()
        sim.sync()
        st = sim.stats()
        ms = sum(v["ms"] for k, v in st["kernels"].items() if k != "init") / 2
        nl = sum(v["launches"] for k, v in st["kernels"].items() if k != "init") / 2
        print(f"B={B} L={L} T={T} G={G}: {ms:8.2f} ms/iter, {nl:.0f} passes, {ms/nl:.2f} ms/pass -> {1000/((ms+3)*1e-3):.0f} gate-applies/s  norm2={sim.norm2():.6f}", flush=True)
