"""Cost of one more block in a tile pass (n=30): passes built from N copies of the same kind of block, merging off."""
import os, sys
os.environ["QSIM_SCHED_MERGE"] = "0"
sys.path.insert(0, '.')
import numpy as np
from gpu_quantum_simulator_amd import Circuit, Simulator

n = 30
rng = np.random.default_rng(1)
def ru(d):
    a = rng.standard_normal((d, d)) + 1j * rng.standard_normal((d, d))
    q, r = np.linalg.qr(a)
    return q * (np.diag(r) / np.abs(np.diag(r)))

H = np.array([[1, 1], [1, -1]]) / np.sqrt(2)
kinds = {
    "dense 2q (G2)": lambda c, i: c.append_2q(ru(4), 10 + (i % 3) * 2 + 1, 10 + (i % 3) * 2),
    "dense 1q (G1)": lambda c, i: c.append_1q(ru(2), 10 + i % 9),
    "diag 1q": lambda c, i: c.append_1q(np.diag(np.exp(1j * rng.uniform(0, 6, 2))), 10 + i % 9),
    "cx in tile (SP k2 t1)": lambda c, i: c.append_cx(10 + i % 9, 10 + (i + 1) % 9),
    "cx low bits": lambda c, i: c.append_cx(i % 3, (i + 1) % 3),
    "h+cx cluster (SP k2 t2)": lambda c, i: (c.append_1q(H, 10 + (i + 1) % 9), c.append_cx(10 + i % 9, 10 + (i + 1) % 9)),
}
with Simulator(n, fuse=3, profile=True, tile_max_ops=64) as sim:
    for name, add in kinds.items():
        res = []
        for N in (2, 6, 10, 14):
            c = Circuit.empty(n)
            for q in range(10, 19):  # pin the tile set: one gate on every qubit of it first
                c.append_1q(H, q)
            for i in range(N):
                add(c, i)
            def body():
                sim.reset(); sim.run(c); sim.flush()
            body(); sim.sync(); sim.reset_stats(); body(); body(); sim.sync()
            st = sim.stats()["kernels"]
            ms = sum(v["ms"] for k, v in st.items() if k != "init") / 2
            nl = sum(v["launches"] for k, v in st.items() if k != "init") / 2
            res.append((N, nl, ms))
        slope = (res[-1][2] - res[1][2]) / (res[-1][0] - res[1][0])
        print(f"{name:28s} " + " ".join(f"N={N}:{nl:.0f}p/{ms:6.2f}ms" for N, nl, ms in res) + f"  slope {slope:.3f} ms/block", flush=True)
