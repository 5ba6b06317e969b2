#!/usr/bin/env python3
"""GPU micro-benchmarks (HIP-event timed inside libqsim): achieved algorithmic GB/s per kernel and target bit.
Usage: python tools/sweep.py [--qubits 30] [--what 1q,cx,2q,tile,circuit]"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from gpu_quantum_simulator_amd import Circuit, Simulator, circuits, gate_matrix


def timed(sim, fn, reps):
    fn()  # warm
    sim.sync()
    sim.reset_stats()
    for _ in range(reps):
        fn()
    sim.sync()
    st = sim.stats()
    ms = sum(v["ms"] for k, v in st["kernels"].items() if k != "init")
    by = sum(v["bytes"] for k, v in st["kernels"].items() if k != "init")
    kinds = [k for k, v in st["kernels"].items() if v["launches"] and k != "init"]
    return ms / reps, by / (ms * 1e-3) / 1e9 if ms else 0.0, kinds


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--qubits", type=int, default=30)
    ap.add_argument("--what", default="1q,diag,cx,2q,tile,circuit")
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--grid-cap", type=int, default=0)
    a = ap.parse_args()
    n = a.qubits
    what = set(a.what.split(","))
    H, T = gate_matrix("h"), gate_matrix("t")
    rng = np.random.default_rng(0)
    q4, _ = np.linalg.qr(rng.standard_normal((4, 4)) + 1j * rng.standard_normal((4, 4)))
    rows = []
    with Simulator(n, fuse=0, profile=True, grid_cap=a.grid_cap) as sim:
        # make the state dense first
        for q in range(n):
            sim.apply_1q(H, q)
        sim.sync()
        if "1q" in what:
            for q in sorted({0, 1, 3, 5, 6, 7, 8, 10, 12, 14, 16, 20, 24, n - 2, n - 1}):
                ms, gbs, k = timed(sim, lambda: sim.apply_1q(H, q), a.reps)
                rows.append(("dense1q", q, ms, gbs, k))
        if "diag" in what:
            for q in (0, 2, 6, 12, 20, n - 1):
                ms, gbs, k = timed(sim, lambda: sim.apply_1q(T, q), a.reps)
                rows.append(("phase", q, ms, gbs, k))
        if "cx" in what:
            for c, t in ((n - 1, n - 2), (20, 10), (10, 20), (7, 3), (3, 7), (1, 0), (0, n - 1), (n - 1, 0)):
                ms, gbs, k = timed(sim, lambda: sim.apply_cx(c, t), a.reps)
                rows.append(("cx", (c, t), ms, gbs, k))
        if "2q" in what:
            for hi, lo in ((n - 1, n - 2), (20, 10), (12, 6), (n - 1, 6), (20, 3), (5, 2), (7, 0)):
                ms, gbs, k = timed(sim, lambda: sim.apply_2q(q4, hi, lo), a.reps)
                rows.append(("dense2q", (hi, lo), ms, gbs, k))
    for r in rows:
        print(f"{r[0]:8s} {str(r[1]):10s} {r[2]:9.3f} ms  {r[3]:8.1f} GB/s  {r[4]}", flush=True)

    if "pack" in what:
        import torch
        with Simulator(n, fuse=0, profile=True, grid_cap=a.grid_cap) as sim:
            for q in range(n):
                sim.apply_1q(H, q)
            sim.sync()
            dst = torch.empty((1 << n, 2), dtype=torch.float64, device="cuda")
            for bits in ((n - 3, n - 2, n - 1), (0, 1, 2), (1, 12, 20), (5,), (n - 1,), (3, 4, 5, 6)):
                ms, gbs, k = timed(sim, lambda: sim.pack_bits(bits, dst.data_ptr()), 5)
                print(f"pack     {str(bits):14s} {ms:9.3f} ms  {gbs:8.1f} GB/s  {k}", flush=True)
            del dst
    if "tileops" in what:
        # cost of one fused block inside a tile pass, by sparsity class: chains of 2q gates on overlapping pairs of
        # five high qubits (each closes the previous cluster), all inside ONE tile pass
        pairs = [(n - 1, n - 2), (n - 2, n - 3), (n - 3, n - 4), (n - 4, n - 5), (n - 1, n - 5), (n - 1, n - 3), (n - 2, n - 4),
                 (n - 3, n - 5), (n - 1, n - 4), (n - 2, n - 5)]
        ph = np.exp(1j * np.array([0.1, 0.7, 1.3, 2.1]))
        mono = np.zeros((4, 4), dtype=complex)
        for r, c in enumerate((1, 0, 3, 2)):
            mono[r, c] = ph[r]
        cxm = np.eye(4)[[0, 1, 3, 2]].astype(complex)
        pair = cxm @ np.kron(np.diag([1, np.exp(0.3j)]), H)
        for tb, tl in ((12, 4), (11, 4)):
            for label, M in (("dense", q4), ("pair", pair), ("mono", mono), ("cx", cxm)):
                for nops in (1, 2, 3, 4, 6, 8, 12, 16):
                    with Simulator(n, fuse=3, profile=True, tile_bits=tb, tile_low_bits=tl, tile_max_ops=64, grid_cap=a.grid_cap) as sim:
                        for q in range(n):
                            sim.apply_1q(H, q)
                        sim.sync()

                        def body():
                            # spread the pairs over mid/high qubits (like a real pass) instead of the top five only
                            qs = [n - 2, n - 5, n - 8, n - 11, n - 14, n - 17, n - 20, n - 23]
                            for j in range(nops):
                                a_, b_ = qs[j % 8], qs[(j + 1 + (j // 8)) % 8]
                                if a_ == b_:
                                    b_ = qs[(j + 3) % 8]
                                sim.apply_2q(M, max(a_, b_), min(a_, b_))
                            sim.flush()
                        ms, gbs, k = timed(sim, body, 5)
                        st = sim.stats()
                        print(f"tileops B={tb} L={tl} {label:5s} ops={nops:2d}: {ms:8.3f} ms {gbs:8.1f} GB/s launches/iter={st['launches']/5:.1f}", flush=True)
    if "tile" in what:
        # tile kernel with k dense 4x4 ops on fixed high qubits, for several geometries
        for tb, tl in ((12, 7), (12, 6), (11, 7), (11, 6), (13, 7), (10, 6)):
            for nops in (1, 2, 4, 8, 16):
                with Simulator(n, fuse=3, profile=True, tile_bits=tb, tile_low_bits=tl, tile_max_ops=64, grid_cap=a.grid_cap) as sim:
                    for q in range(n):
                        sim.apply_1q(H, q)
                    sim.sync()

                    def body():
                        for j in range(nops):
                            hi, lo = (n - 1, n - 3) if j % 2 == 0 else (n - 3, 2)
                            sim.apply_2q(q4, hi, lo)
                            sim.apply_1q(H, n - 1 if j % 2 else 2)  # break the pair so clusters stay separate
                            sim.apply_cx(n - 2, 4)
                        sim.flush()
                    ms, gbs, k = timed(sim, body, 5)
                    st = sim.stats()
                    print(f"tile B={tb} L={tl} src_ops={nops:2d}: {ms:8.3f} ms/iter {gbs:8.1f} GB/s launches/iter={st['launches']/5:.1f} {k}", flush=True)
    if "circuit" in what:
        gates = circuits.random_gates(n, 1000, 20240117 + n, "all")
        c = Circuit.from_gates(n, gates)
        for fuse, opts in ((2, {}), (3, {}), (3, {"tile_bits": 11, "tile_low_bits": 6}), (3, {"tile_bits": 12, "tile_low_bits": 6}),
                           (3, {"tile_bits": 13, "tile_low_bits": 7}), (3, {"tile_bits": 13, "tile_low_bits": 6})):
            with Simulator(n, fuse=fuse, profile=True, grid_cap=a.grid_cap, **opts) as sim:
                def body():
                    sim.reset()
                    sim.run(c)
                    sim.flush()
                ms, gbs, k = timed(sim, body, 2)
                st = sim.stats()
                print(f"circuit fuse={fuse} {opts}: {ms:9.2f} ms kernel/iter -> {1000/ (ms*1e-3):9.1f} gate-applies/s, {gbs:7.1f} GB/s, launches/iter={st['launches']/2:.0f}", flush=True)
                print("   ", json.dumps({k2: round(v['ms']/2, 2) for k2, v in st['kernels'].items() if v['launches']}), flush=True)


if __name__ == "__main__":
    main()
