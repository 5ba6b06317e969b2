"""Replays tests/stress_parity.py's random stream up to one case and runs only that case, verbosely.  Usage: stress_repro.py CASE SEED"""
import os, sys, tempfile
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
from gpu_quantum_simulator_amd import Circuit, Cluster, Simulator, circuits
from oracle import oracle
target, seed = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed)
with tempfile.TemporaryDirectory() as d:
    for case in range(target + 1):
        n = int(rng.integers(3, 21)); depth = int(rng.integers(30, 700)); vocab = "all" if rng.random() < 0.6 else "clifford_t"
        tile_bits = int(rng.integers(8, 14))
        opts = {"tile_bits": tile_bits, "tile_low_bits": int(rng.integers(max(2, tile_bits - 10), min(6, tile_bits - 2) + 1)),
                "tile_max_ops": int(rng.integers(1, 40)), "debug_tile_order": int(rng.integers(0, 6)),
                "plan_cache": int(rng.integers(0, 2)), "pingpong": int(rng.choice([0, 2])), "sparse_start": int(rng.integers(0, 2)),
                "debug_plan_key": int(rng.choice([0, 0, 77]))}
        kind = rng.random()
        if os.environ.get("STRESS_KIND") == "cluster":
            kind = 0.9  # only sharded cases (same random stream otherwise)
        if kind < 0.85:
            continue
        P = int(rng.choice([2, 4, 8])); n = int(rng.integers(14, 20))
        tail = int(rng.choice([0, 8, 24])); tb = int(rng.choice([9, 12])); do_plan = rng.random() < 0.5
        plan_cand = int(rng.choice([1, 1, 4])) if do_plan else 1
        if case != target:
            continue
        print("case", case, "n", n, "depth", depth, vocab, "P", P, "tail", tail, "tile_bits", tb, "plan", do_plan, "pingpong", opts["pingpong"], flush=True)
        path = circuits.random_circuit_file(os.path.join(d, "c.qasm"), n, depth, 9000 + case, vocab)
        _, want, _, _ = oracle.run_qasm(path)
        c = Circuit.from_file(path)
        stale = Circuit.from_gates(n, circuits.random_gates(n, 200, 70000 + case, "all"))
        os.environ["QSIM_SHARD_TAIL"] = str(tail)
        for variant in ("as is", "no stale run", "no plan", "tail 0"):
            if variant == "tail 0":
                os.environ["QSIM_SHARD_TAIL"] = "0"
            with Cluster(n, P, devices=[0] * P, pingpong=opts["pingpong"], tile_bits=tb) as cl:
                if variant != "no stale run":
                    cl.run(stale)
                if do_plan and variant != "no plan":
                    cl.plan(c, plan_cand, 300.0)
                for rep in range(2):
                    cl.run(c)
                    err = float(np.max(np.abs(cl.read() - want)))
                    print(variant, "rep", rep, "err", err, "packs", cl.pack_counts(), flush=True)
