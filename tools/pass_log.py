import sys
sys.path.insert(0, '.')
from gpu_quantum_simulator_amd import Circuit, Simulator, circuits
n = 30
c = Circuit.from_gates(n, circuits.random_gates(n, 1000, 20240117 + n, "all"))
B, L, T = (int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "12,4,512").split(','))
with Simulator(n, fuse=3, profile=True, tile_bits=B, tile_low_bits=L, tile_threads=T, tile_max_ops=64) as sim:
    def body():
        sim.reset(); sim.run(c); sim.flush()
    body(); sim.sync(); sim.reset_stats()
    body(); sim.sync()
    for k, nops, hm, ms in sim.launch_log():
        bits = [b for b in range(40) if hm >> b & 1]
        print(f"{k:6s} ops={nops:2d} ms={ms:7.3f} high={bits}")
