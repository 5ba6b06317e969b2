#!/usr/bin/env python3
"""bench.py — BASELINE.json's metric on its config: gate-applies/s (+ achieved HBM GB/s) for a seeded
random circuit, n = 30 qubits fp64, 1000 gate statements (configs[3]; fits one MI355X: 16 GiB state).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = the whole circuit applied to a fresh |0...0> (init kernel + every fused pass), inputs
resident in HBM (the gate list is parsed once, before the timed region).  N > 1 shards the SAME 2^30
state over N ranks (strong scaling; top log2 N qubits global, exchanged over RCCL) — see
gpu_quantum_simulator_amd/distributed.py.

Rank 0 prints ONE JSON line.  Besides the contract's keys it carries
  roofline     — dominant kernel class: algorithmic bytes / HIP-event time measured live on the engine's
                 stream during the timed steps, against the 8 TB/s HBM3E peak;
  cpu_baseline — the reference's own loops (oracle/_ref, kind "reference") or the restatement
                 (oracle/liboracle.so, kind "port") timed on this host, 1 thread, on the first few
                 gates of the same circuit (bounded to ~15 s).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--qubits", type=int, default=30)
    ap.add_argument("--depth", type=int, default=1000)
    ap.add_argument("--vocabulary", default="all", choices=["all", "clifford_t"])
    ap.add_argument("--seed", type=int, default=None, help="default 20240117 + qubits (SURVEY §8d)")
    ap.add_argument("--fuse", type=int, default=3)
    ap.add_argument("--tile-bits", type=int, default=None)
    ap.add_argument("--tile-low-bits", type=int, default=None)
    ap.add_argument("--tile-max-ops", type=int, default=None)
    ap.add_argument("--grid-cap", type=int, default=None)
    ap.add_argument("--probe", type=int, default=None, metavar="Q",
                    help="single-qubit roofline probe instead of the random circuit: `depth` h gates on qubit Q, fusion off")
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"],
                    help="strong: the same n on every N (default, the metric is quoted at n=30); weak: n = qubits + log2(N)")
    ap.add_argument("--force-sharded", action="store_true",
                    help="with --gpus 1: still go through torch.distributed + ShardedSimulator (rehearsal of the N>1 code path)")
    ap.add_argument("--precision", type=int, default=64, choices=[64, 32],
                    help="64 = the headline / parity configuration; 32 = fp32 state like the reference's CUDA variants "
                         "(an extra measurement, single GPU only, reported with dtype f32)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    return ap.parse_args()


KERNEL_SYMBOL = {"tile": "k_tile<12, 512>", "gate1": "k_gate1_hi<4, false>",
                 "gate1_lo": "k_gate1_lo<4, false>", "gate2": "k_gate2_hh<2, false>"}  # inside namespace qsim::f64


def pmc_traffic(kernel_class, is_default_workload):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc summary of this same
    command (profiles/rNN/bench_n30_pmc_summary.json: FETCH_SIZE x2 on gfx950 + WRITE_SIZE, KiB -> bytes).
    Counters cannot be read from inside the process, so this is null for any other workload."""
    if not is_default_workload:
        return None
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "bench_n30_pmc_summary.json")))
    if not files:
        return None
    with open(files[-1]) as f:
        summary = json.load(f)
    want = KERNEL_SYMBOL.get(kernel_class)
    for name, entry in summary.get("kernels", {}).items():
        if want and name.endswith("::" + want) and "f32" not in name:
            return entry["hbm_traffic_bytes_per_launch"]
    return None


def cpu_baseline(n, gates, budget_s):
    """The reference's hot loops on this host's cores (1 thread), bounded sample of the same circuit."""
    import ctypes

    import numpy as np
    from oracle import oracle  # the checker, used here only as the timed CPU baseline

    oracle.build(with_reference=False)
    use_ref = oracle.have_reference()
    dp = ctypes.POINTER(ctypes.c_double)
    state = np.zeros(1 << n, dtype=np.complex128)
    state[0] = 1.0
    state[1:] = 0.0  # touch every page before the clock starts
    sp = state.view(np.float64).ctypes.data_as(dp)
    from gpu_quantum_simulator_amd import gate_matrix
    if use_ref:
        R = oracle.reference_lib()
        one_q = lambda u, q: R.execute_single_qubit_gate(sp, n, u.view(np.float64).ctypes.data_as(dp), q)
        cx = lambda c, t: R.execute_cnot(sp, n, c, t)
    else:
        L = oracle.lib()
        one_q = lambda u, q: L.oracle_apply_1q(sp, n, u.view(np.float64).ctypes.data_as(dp), q)
        cx = lambda c, t: L.oracle_apply_cx(sp, n, c, t)
    done = 0
    t0 = time.perf_counter()
    for g in gates:
        if g[0] == "cx":
            cx(g[1], g[2])
        else:
            tok = f"rz({g[1]!r})" if g[0] == "rz" else g[0]
            u = np.ascontiguousarray(gate_matrix(tok).T.reshape(4))  # symmetric anyway
            one_q(u, g[-1])
        done += 1
        if time.perf_counter() - t0 >= budget_s:
            break
    dt = time.perf_counter() - t0
    return {"value": done / dt, "unit": "gate-applies/s", "cores": 1, "host_cores": os.cpu_count(),
            "kind": "reference" if use_ref else "port",
            "sample": f"first {done} gate statements of the same n={n} circuit, {dt:.1f} s, state resident in host RAM"}


def main():
    args = parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit(f"--gpus {args.gpus} needs one process per GPU: launch with python -m torch.distributed.run "
                     f"--nproc-per-node {args.gpus} bench.py --gpus {args.gpus} ...")
        sys.exit(f"WORLD_SIZE={world} does not match --gpus {args.gpus}")

    import numpy as np
    import torch

    from gpu_quantum_simulator_amd import Circuit, Simulator, circuits

    n = args.qubits + (int(round(__import__("math").log2(world))) if args.scaling == "weak" else 0)
    seed = args.seed if args.seed is not None else 20240117 + n
    if args.probe is not None:
        gates = circuits.probe_gates(n, args.probe, args.depth)
        fuse = 0
        workload = f"single-qubit probe: {args.depth} x h q[{args.probe}], n={n}, fusion off"
    else:
        gates = circuits.random_gates(n, args.depth, seed, args.vocabulary)
        fuse = args.fuse
        workload = f"random circuit ({args.vocabulary}), n={n}, depth {args.depth}, seed {seed}"
    opts = {k: v for k, v in (("tile_bits", args.tile_bits), ("tile_low_bits", args.tile_low_bits),
                              ("tile_max_ops", args.tile_max_ops), ("grid_cap", args.grid_cap)) if v is not None}

    dist = None
    if args.precision == 32 and (world > 1 or args.force_sharded):
        sys.exit("--precision 32 is single-GPU only (shards and clusters are fp64)")
    if world > 1 or args.force_sharded:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        if "MASTER_ADDR" not in os.environ:
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29511", RANK="0", WORLD_SIZE="1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        from gpu_quantum_simulator_amd.distributed import ShardedSimulator
        sim = ShardedSimulator(n, gates, device=local_rank, fuse=fuse, profile=True, **opts)
        run_step = sim.run_step
    else:
        torch.cuda.set_device(local_rank)
        circuit = Circuit.from_gates(n, gates)
        sim = Simulator(n, local_rank, fuse=fuse, profile=True, precision=args.precision, **opts)

        def run_step():
            sim.reset()
            sim.run(circuit)
            sim.flush()

    def fence():
        sim.sync()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        run_step()
    fence()
    sim.reset_stats()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run_step()
    fence()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    stats = sim.stats()
    norm2 = sim.norm2()

    # north_star target "single-qubit gate apply at n=30": a short dense-1q probe on the same (now dense) state
    probe = None
    if dist is None and args.probe is None and n >= 8:
        from gpu_quantum_simulator_amd import _lib as _qlib, gate_matrix
        H = gate_matrix("h")
        sim.set_option(_qlib.OPT_FUSE, 0)  # one launch per gate on the live (dense, random) state; H^6 = I
        probe = {}
        for q in sorted({0, min(12, n - 1), n - 1}):
            sim.sync()
            sim.reset_stats()
            for _ in range(6):
                sim.apply_1q(H, q)
            sim.sync()
            k = {kk: vv for kk, vv in sim.stats()["kernels"].items() if vv["launches"] and kk != "init"}
            name = next(iter(k))
            gbs = k[name]["bytes"] / (k[name]["ms"] * 1e-3) / 1e9
            probe[f"q{q}"] = {"kernel": name, "achieved": gbs, "frac": gbs / HBM_PEAK_GBPS, "avg_launch_ms": k[name]["ms"] / 6}
        sim.set_option(_qlib.OPT_FUSE, fuse)

    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        value = args.depth * args.steps / elapsed
        kernels = {k: v for k, v in stats["kernels"].items() if v["launches"] and k != "init"}
        dom = max(kernels, key=lambda k: kernels[k]["ms"]) if kernels else None
        roof = None
        if dom and kernels[dom]["ms"] > 0:
            achieved = kernels[dom]["bytes"] / (kernels[dom]["ms"] * 1e-3) / 1e9
            roof = {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBPS,
                    "traffic": pmc_traffic(dom, args.precision == 64 and args.probe is None and n == 30 and args.depth == 1000 and fuse == 3
                                           and args.gpus == 1 and not opts and args.vocabulary == "all"),
                    "launches": kernels[dom]["launches"],
                    "avg_launch_ms": kernels[dom]["ms"] / kernels[dom]["launches"],
                    "algorithmic_bytes_per_launch": kernels[dom]["bytes"] / kernels[dom]["launches"]}
        total_kernel_ms = sum(v["ms"] for v in stats["kernels"].values())
        out = {
            "metric": "gate-applies/sec", "value": value, "unit": "gate-applies/s", "n_gpus": args.gpus,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": args.scaling, "vs_baseline": None, "dtype": "f64" if args.precision == 64 else "f32", "data": "synthetic",
            "config": {"workload": workload, "qubits": n, "gate_statements": args.depth, "fuse": fuse,
                       "state_bytes": (16 if args.precision == 64 else 8) * (1 << n), "parallelism": f"shard{args.gpus}", **opts},
            "hbm_gbps_all_kernels": stats["algorithmic_bytes"] / (total_kernel_ms * 1e-3) / 1e9 if total_kernel_ms else None,
            "launches_per_step": stats["launches"] / args.steps,
            "kernel_ms_per_step": {k: v["ms"] / args.steps for k, v in stats["kernels"].items() if v["launches"]},
            "norm2": norm2,
            "roofline": roof,
            "roofline_1q_probe": probe,
        }
        if dist is not None:
            xs = sim.exchange_seconds / args.steps
            xb = sim.exchange_bytes / args.steps
            out["exchange"] = {"per_step": sim.plan.exchanges, "qubits_swapped": [len(s[1]) for s in sim.plan.steps if s[0] == "exchange"],
                               "bytes_sent_per_rank_per_step": xb, "seconds_per_step": xs,
                               "xgmi_gbps_per_rank": (xb / xs / 1e9) if xs > 0 else None,
                               "note": "seconds include the pack kernel and host-side stream hand-offs (rank 0's clock)"}
        if not args.no_cpu_baseline and args.gpus == 1:
            out["cpu_baseline"] = cpu_baseline(n, gates, args.cpu_seconds)
        print(json.dumps(out), flush=True)
    sim.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
