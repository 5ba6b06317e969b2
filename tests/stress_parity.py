"""Randomised parity stress, run by hand on the GPU box (python tests/stress_parity.py CASES SEED; not collected by pytest; it lives
under tests/ because it calls the oracle): many seeded circuits x engine options (tile shape, ops
per pass, shuffled tile-bit orders, plan cache on/off, out-of-place passes, sparse start, fp32, virtual-shard clusters)
against the oracle."""
import os
import sys
import tempfile
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
from gpu_quantum_simulator_amd import Circuit, Cluster, Simulator, circuits
from oracle import oracle

N = int(sys.argv[1]) if len(sys.argv) > 1 else 150
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 2024)
worst = {"f64": 0.0, "f32": 0.0, "cluster": 0.0}
with tempfile.TemporaryDirectory() as d:
    for case in range(N):
        n = int(rng.integers(3, 21))
        depth = int(rng.integers(30, 700))
        vocab = "all" if rng.random() < 0.6 else "clifford_t"
        path = circuits.random_circuit_file(os.path.join(d, "c.qasm"), n, depth, 9000 + case, vocab)
        _, want, _, _ = oracle.run_qasm(path)
        c = Circuit.from_file(path)
        tile_bits = int(rng.integers(8, 14))
        opts = {"tile_bits": tile_bits, "tile_low_bits": int(rng.integers(max(2, tile_bits - 10), min(6, tile_bits - 2) + 1)),
                "tile_max_ops": int(rng.integers(1, 40)), "debug_tile_order": int(rng.integers(0, 6)),
                "plan_cache": int(rng.integers(0, 2)), "pingpong": int(rng.choice([0, 2])), "sparse_start": int(rng.integers(0, 2)),
                "debug_plan_key": int(rng.choice([0, 0, 77]))}
        kind = rng.random()
        if os.environ.get("STRESS_KIND") == "cluster":
            kind = 0.9  # only sharded cases (same random stream otherwise)
        if kind < 0.7:
            with Simulator(n, fuse=3, **opts) as sim:
                for rep in range(2):
                    sim.reset(); sim.run(c)
                    err = float(np.max(np.abs(sim.read() - want)))
                    worst["f64"] = max(worst["f64"], err)
                    assert err < 1e-10, (case, n, depth, vocab, opts, rep, err)
        elif kind < 0.85:
            opts.pop("debug_plan_key")
            with Simulator(n, fuse=3, precision=32, **opts) as sim:
                sim.run(c)
                err = float(np.max(np.abs(sim.read() - want)))
                worst["f32"] = max(worst["f32"], err)
                assert err < 5e-5, (case, n, depth, vocab, opts, err)
        else:
            # round 3: registers large enough for the fused re-layout (shards of >= 2^12 amplitudes), a stale unrelated run in
            # every buffer first, optional per-shard planning, the small-last-pass hand-over on and off, forced key collisions
            P = int(rng.choice([2, 4, 8]))
            n = int(rng.integers(14, 20))
            path = circuits.random_circuit_file(os.path.join(d, "c.qasm"), n, depth, 9000 + case, vocab)
            _, want, _, _ = oracle.run_qasm(path)
            c = Circuit.from_file(path)
            stale = Circuit.from_gates(n, circuits.random_gates(n, 200, 70000 + case, "all"))
            os.environ["QSIM_SHARD_TAIL"] = str(int(rng.choice([0, 8, 24])))
            with Cluster(n, P, devices=[0] * P, pingpong=opts["pingpong"], tile_bits=int(rng.choice([9, 12]))) as cl:
                cl.run(stale)
                if rng.random() < 0.5:
                    cl.plan(c, int(rng.choice([1, 1, 4])), 300.0)  # schedule choice per shard; sometimes with timing (measured choice + tile-bit orders)
                for rep in range(2):
                    cl.run(c)
                    err = float(np.max(np.abs(cl.read() - want)))
                    worst["cluster"] = max(worst["cluster"], err)
                    assert err < 1e-10, (case, n, depth, vocab, P, rep, err)
                fused = cl.pack_counts()
                worst["fused_relayouts"] = worst.get("fused_relayouts", 0) + fused[0]
                worst["separate_relayouts"] = worst.get("separate_relayouts", 0) + fused[1]
            os.environ.pop("QSIM_SHARD_TAIL", None)
        if case % 25 == 0:
            print(case, worst, flush=True)
print("stress ok", N, worst)
