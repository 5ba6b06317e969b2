import sys
sys.path.insert(0, '.')
import numpy as np
from gpu_quantum_simulator_amd import Simulator, gate_matrix
n = 30
H = gate_matrix("h")
ph = np.exp(1j * np.array([0.1, 0.7, 1.3, 2.1]))
mono = np.zeros((4, 4), dtype=complex)
for r, c in enumerate((1, 0, 3, 2)):
    mono[r, c] = ph[r]
cases = {"hi{25,28}": [(28, 25)], "hi{22,25,28}": [(28, 25), (25, 22)], "hi{19..28}": [(28, 25), (25, 22), (22, 19)],
         "hi{9,14}": [(14, 9)], "hi{9,14,19}": [(14, 9), (19, 14)], "lo{2,3}+hi{20}": [(3, 2), (20, 3)]}
for name, pairs in cases.items():
    for pad in (-1, 6, 8, 10, 12, 14, 16, 20):
        with Simulator(n, fuse=3, profile=True, tile_bits=12, tile_low_bits=4, tile_pad_from=pad) as sim:
            for q in range(n):
                sim.apply_1q(H, q)
            sim.sync()
            def body():
                for hi, lo in pairs:
                    sim.apply_2q(mono, hi, lo)
                sim.flush()
            body(); sim.sync(); sim.reset_stats()
            for _ in range(5): body()
            sim.sync()
            log = sim.launch_log()
            ms = sum(l[3] for l in log) / 5
            bits = [b for b in range(40) if log[0][2] >> b & 1]
            print(f"{name:16s} pad_from={pad:3d}: {ms:7.3f} ms  kernel={log[0][0]} high={bits}", flush=True)
