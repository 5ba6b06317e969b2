"""Does it matter how the second buffer of the out-of-place passes is aligned against the first?  A pass reads index x of one buffer and
writes index x of the other; with equal alignment both land on the same channel / bank group of the HBM.  The spare buffer is lent
(qsim_set_spare_buffer) at several byte offsets inside one torch allocation; step time of the n = 30 bench circuit with its schedule
chosen and the tile-bit orders ascending (no measured orders: they would be measured against one offset).
Usage: python tools/spare_offset.py [n]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from gpu_quantum_simulator_amd import Circuit, Simulator, circuits
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
c = Circuit.from_gates(n, circuits.random_gates(n, 1000, 20240117 + n, "all"))
nbytes = 16 << n
pad = 64 << 20
buf = torch.empty(nbytes + pad, dtype=torch.uint8, device="cuda")
base = (buf.data_ptr() + (2 << 20) - 1) & ~((2 << 20) - 1)  # 2 MiB aligned
with Simulator(n) as sim:
    print(f"state at {sim.device_ptr:#x}, lent buffer base {base:#x}", flush=True)
    sim.choose_schedule(c)
    MiB, KiB = 1 << 20, 1 << 10
    offsets = [0] + [k * MiB for k in (1, 2, 3, 4, 5, 8, 9, 15, 16, 17, 18, 19, 20, 21, 24, 25, 32, 33, 34, 48, 49, 51)] + \
              [16 * MiB + k * KiB for k in (128, 256, 512)] + [k * KiB for k in (128, 256, 512)]
    if len(sys.argv) > 2:
        offsets = [int(float(x) * MiB) for x in sys.argv[2].split(",")]
    for rep in range(2):
        for off in offsets:
            sim.sync()
            sim.reset()
            sim.set_spare_buffer(base + off)
            for _ in range(2):
                sim.reset(); sim.run(c); sim.sync()
            t0 = time.time()
            for _ in range(8):
                sim.reset(); sim.run(c); sim.sync()
            print(f"offset {off:>10d} B: {(time.time() - t0) * 125:.2f} ms/step", flush=True)
    sim.set_spare_buffer(None)
