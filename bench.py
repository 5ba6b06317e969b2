#!/usr/bin/env python3
"""bench.py — BASELINE.json's metric on its config: gate-applies/s (+ achieved HBM GB/s) for a seeded
random circuit, n = 30 qubits fp64, 1000 gate statements (configs[3]; fits one MI355X: 16 GiB state).

    python bench.py --gpus N --steps K --warmup W          (N > 1: starts its own N rank processes)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = the whole circuit applied to a fresh |0...0> (init + every fused pass), inputs resident in HBM (the
gate list is parsed once, before the timed region).  N > 1 shards the SAME 2^n state over N ranks, one process per
GPU (strong scaling; top log2 N qubits global, exchanged with RCCL send/recv) — gpu_quantum_simulator_amd/distributed.py.

Rank 0 prints ONE JSON line.  Besides the contract's keys it carries
  roofline     — dominant kernel class: algorithmic bytes / HIP-event time measured live on the engine's own
                 stream during the timed steps, against the 8 TB/s HBM3E peak; `traffic` = PMC bytes per launch
                 from the committed rocprofv3 summary of this same command (`traffic_source` names it);
  cpu_baseline — the CPU restatement of quantum_simulator.c's loops (oracle/liboracle.so, kind "port",
                 byte-identical to the compiled reference in the container tests) timed on this host,
                 1 thread, on the first gates of the same circuit (bounded to ~12 s); N = 1 only;
  sizes        — the same measurement at n = 24, 28, 32 (north_star / tester.bash:5-48 protocol): gate-applies/s,
                 the tile kernel's GB/s and fraction of peak, and a bounded CPU sample beside each (N = 1 only).
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def parse_args(argv=None):
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
    ap.add_argument("--tile-threads", type=int, default=None)
    ap.add_argument("--pingpong", type=int, default=None, choices=[0, 1, 2],
                    help="QSIM_OPT_PINGPONG: tile passes out of place between two buffers (default: the library's, 1 = from 8 GiB of state)")
    ap.add_argument("--probe", type=int, default=None, metavar="Q",
                    help="single-qubit roofline probe instead of the random circuit: `depth` h gates on qubit Q, fusion off")
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"],
                    help="strong: the same n on every N (default, the metric is quoted at n=30); weak: n = qubits + log2(N)")
    ap.add_argument("--force-sharded", action="store_true",
                    help="with --gpus 1: still go through torch.distributed + ShardedSimulator (rehearsal of the N>1 code path)")
    ap.add_argument("--precision", type=int, default=64, choices=[64, 32],
                    help="64 = the headline / parity configuration; 32 = fp32 state like the reference's CUDA variants "
                         "(an extra measurement, single GPU only, reported with dtype f32)")
    ap.add_argument("--sizes", default="24,28,32",
                    help="other register sizes measured after the headline run and reported under `sizes` ('' = none)")
    ap.add_argument("--size-steps", type=int, default=3)
    ap.add_argument("--no-tune", action="store_true",
                    help="skip the geometry planning step (qsim_tune_circuit) and run every pass with its tile bits in ascending order")
    ap.add_argument("--wisdom", default=None, metavar="PATH",
                    help="load measured pass geometries from PATH before planning and save the table there afterwards "
                         "(a second run, e.g. under rocprofv3, then launches no planning passes)")
    ap.add_argument("--tune-candidates", type=int, default=48)
    ap.add_argument("--tune-ms", type=float, default=8000.0, help="wall-time budget of the planning step per register size")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-full-sweeps", action="store_true",
                    help="skip the extra steps with QSIM_OPT_SPARSE_START off (profile runs: every launch then is one of the timed kind)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    return ap.parse_args(argv)


KERNEL_SYMBOL = {"tile": "k_tile<12, 512, false>", "gate1": "k_gate1_hi<4, false>",
                 "gate1_lo": "k_gate1_lo<4, false>", "gate2": "k_gate2_hh<2, false>"}  # inside namespace qsim::f64


def pmc_traffic(kernel_class, is_default_workload):
    """(HBM bytes per launch of the dominant kernel, source file) from the committed rocprofv3 --pmc summary of this
    same command (profiles/rNN/bench_n30_pmc_summary.json: FETCH_SIZE x2 on gfx950 + WRITE_SIZE, KiB -> bytes).
    Counters cannot be read from inside the process, so this is (None, None) for any other workload; the summary must
    be refreshed in the commit that changes the kernel — `traffic_source` says which profile the number is from."""
    if not is_default_workload:
        return None, None
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "bench_n30_pmc_summary.json")))
    if not files:
        return None, None
    with open(files[-1]) as f:
        summary = json.load(f)
    want = KERNEL_SYMBOL.get(kernel_class)
    for name, entry in summary.get("kernels", {}).items():
        if want and name.endswith("::" + want) and "f32" not in name:
            return entry["hbm_traffic_bytes_per_launch"], os.path.relpath(files[-1], ROOT)
    return None, None


def host_ram_bytes():
    try:
        return os.sysconf("SC_PAGE_SIZE") * os.sysconf("SC_PHYS_PAGES")
    except (ValueError, OSError):
        return 0


def cpu_baseline(n, gates, budget_s):
    """quantum_simulator.c's hot loops (the oracle's restatement of :81-106) on this host, 1 thread, on a bounded
    sample of the same circuit."""
    import ctypes

    import numpy as np
    from oracle import oracle  # the checker, used here only as the timed CPU baseline

    need = 16 << n
    if need + (8 << 30) > host_ram_bytes():
        return {"value": None, "unit": "gate-applies/s", "cores": 1, "host_cores": os.cpu_count(), "kind": "port",
                "sample": f"not run: the 2^{n} state ({need >> 30} GiB) does not fit this host's RAM"}
    oracle.build(with_reference=False)
    dp = ctypes.POINTER(ctypes.c_double)
    state = np.zeros(1 << n, dtype=np.complex128)
    state[0] = 1.0
    state[1:] = 0.0  # touch every page before the clock starts
    sp = state.view(np.float64).ctypes.data_as(dp)
    from gpu_quantum_simulator_amd import gate_matrix
    L = oracle.lib()
    done = 0
    t0 = time.perf_counter()
    for g in gates:
        if g[0] == "cx":
            L.oracle_apply_cx(sp, n, g[1], g[2])
        else:
            tok = f"rz({g[1]!r})" if g[0] == "rz" else g[0]
            u = np.ascontiguousarray(gate_matrix(tok).T.reshape(4))  # symmetric anyway
            L.oracle_apply_1q(sp, n, u.view(np.float64).ctypes.data_as(dp), g[-1])
        done += 1
        if time.perf_counter() - t0 >= budget_s:
            break
    dt = time.perf_counter() - t0
    del state
    return {"value": done / dt, "unit": "gate-applies/s", "cores": 1, "host_cores": os.cpu_count(), "kind": "port",
            "sample": f"first {done} gate statements of the same n={n} circuit, {dt:.1f} s, 1 thread, state resident in host RAM"}


def exchange_model(depth, vocabulary, ms_per_step_n30, link_gbps=50.0, pack_gbps=5000.0):
    """What the planner's cost model predicts for the multi-GPU configs of BASELINE.json, computed from the plans alone
    (host work, no GPU): exchanges, qubits swapped, bytes per rank, exchange time, and the step time that follows when
    the local passes scale with the shard size from the measured 1-GPU n=30 step."""
    from gpu_quantum_simulator_amd import circuits, gate_matrix
    from gpu_quantum_simulator_amd.distributed import ShardPlan, normalize_gates
    rows = []
    for n, P in ((30, 2), (30, 4), (30, 8), (33, 8)):
        gates = normalize_gates(circuits.random_gates(n, depth, 20240117 + n, vocabulary), gate_matrix)
        plan = ShardPlan(n, P.bit_length() - 1, gates, 0)
        nbytes, secs = plan.predict(link_gbps, pack_gbps)
        local_ms = ms_per_step_n30 * (2.0 ** (n - 30)) / P
        rows.append({"qubits": n, "ranks": P, "exchanges": plan.exchanges,
                     "qubits_swapped": [len(st[1]) for st in plan.steps if st[0] == "exchange"],
                     "bytes_sent_per_rank": nbytes, "predicted_exchange_ms": 1e3 * secs,
                     "predicted_local_ms": local_ms, "predicted_step_ms": local_ms + 1e3 * secs,
                     "vs_ideal": (local_ms + 1e3 * secs) / (ms_per_step_n30 * (2.0 ** (n - 30)) / P)})
    return {"assumptions": {"link_gbps_per_direction": link_gbps, "pack_gbps": pack_gbps,
                            "note": "xGMI link ~76.8 GB/s per direction peak (7 x ~153 GB/s bidirectional per GPU), 65 % assumed; "
                                    "no overlap of exchange and local passes; local passes scale with the shard size"},
            "configs": rows}


def launch_ranks(args):
    """--gpus N > 1 without a launcher: become the parent of N rank processes.  Nothing here touches the GPU."""
    from gpu_quantum_simulator_amd import launch
    present = launch.count_gpus()
    if present is not None and present < args.gpus:
        sys.stderr.write(f"bench.py: {args.gpus} GPUs requested, {present} present\n")
        return 2
    rc, out = launch.spawn_ranks(args.gpus, os.path.abspath(__file__), sys.argv[1:])
    lines = [ln for ln in out.splitlines() if ln.startswith('{"metric"')]
    for ln in out.splitlines():
        if ln not in lines:
            sys.stderr.write(ln + "\n")
    if lines:
        print(lines[-1], flush=True)
    elif rc == 0:
        rc = 1
    return rc


class Bench:
    def __init__(self, args):
        self.args = args
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.dist = None

    def setup(self):
        import torch
        args = self.args
        ndev = torch.cuda.device_count()  # does not initialise the GPU on this image
        if ndev < max(args.gpus, 1) or self.local_rank >= ndev:
            sys.stderr.write(f"bench.py: {args.gpus} GPUs requested, {ndev} present\n")
            sys.exit(2)
        torch.cuda.set_device(self.local_rank)
        if self.world > 1 or args.force_sharded:
            import torch.distributed as dist
            if "MASTER_ADDR" not in os.environ:
                os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29511", RANK="0", WORLD_SIZE="1")
            dist.init_process_group("nccl", device_id=torch.device("cuda", self.local_rank))
            self.dist = dist
        self.torch = torch

    def fence(self, sim):
        sim.sync()
        self.torch.cuda.synchronize()
        if self.dist is not None:
            self.dist.barrier()
            self.torch.cuda.synchronize()

    def measure(self, n, depth, vocabulary, seed, steps, warmup, fuse, opts, probe_q=None, with_1q_probe=False):
        """Builds the simulator for one register size, runs `warmup` + `steps` timed steps, returns the numbers."""
        from gpu_quantum_simulator_amd import Circuit, Simulator, circuits
        args, dist, torch = self.args, self.dist, self.torch
        if probe_q is not None:
            gates = circuits.probe_gates(n, probe_q, depth)
            fuse = 0
            workload = f"single-qubit probe: {depth} x h q[{probe_q}], n={n}, fusion off"
        else:
            gates = circuits.random_gates(n, depth, seed, vocabulary)
            workload = f"random circuit ({vocabulary}), n={n}, depth {depth}, seed {seed}"
        if dist is not None:
            from gpu_quantum_simulator_amd.distributed import ShardedSimulator
            sim = ShardedSimulator(n, gates, device=self.local_rank, fuse=fuse, profile=True, **opts)
            run_step = sim.run_step
        else:
            circuit = Circuit.from_gates(n, gates)
            sim = Simulator(n, self.local_rank, fuse=fuse, profile=True, precision=args.precision, **opts)

            def run_step():
                sim.reset()
                sim.run(circuit)
                sim.flush()

        # Planning, outside the timed region (like parsing): one untuned step is timed for the record, then every pass of
        # the schedule is measured under candidate orders of its tile bits and the best order per geometry is kept
        # (qsim_tune_circuit).  Same passes, same blocks, same results — only the walk order changes.
        tuning = None
        if dist is not None and probe_q is None and not args.no_tune and fuse >= 3:
            tuning = sim.tune(args.tune_candidates, args.tune_ms)  # every rank plans its own shard's passes
            self.fence(sim)
        if dist is None and probe_q is None and fuse >= 3:
            sim.choose_schedule(circuit)  # planning, part one: which of the scheduler's two cluster orders this circuit gets
        if dist is None and probe_q is None and not args.no_tune and fuse >= 3:
            run_step()
            self.fence(sim)
            t0 = time.perf_counter()
            run_step()
            self.fence(sim)
            untuned_ms = 1e3 * (time.perf_counter() - t0)
            tuning = sim.tune(circuit, args.tune_candidates, args.tune_ms)
            # with a loaded table (--wisdom) that step already ran in the measured orders: nothing "untuned" to report
            tuning["untuned_ms_per_step"] = untuned_ms if tuning["already_known"] == 0 else None
        for _ in range(warmup):
            run_step()
        self.fence(sim)
        sim.reset_stats()
        t0 = time.perf_counter()
        for _ in range(steps):
            run_step()
        self.fence(sim)
        elapsed = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        stats = sim.stats()
        norm2 = sim.norm2()

        # For the record: the same steps with every pass sweeping the whole register (QSIM_OPT_SPARSE_START = 0).  By default
        # the first passes after a reset only visit the tiles inside the state's support — same
        # amplitudes, fewer bytes; `value` is the default path, this is what it would be without that.
        full_sweeps = None
        if dist is None and probe_q is None and fuse >= 3 and not args.no_full_sweeps:
            from gpu_quantum_simulator_amd import _lib as _qlib
            sim.set_option(_qlib.OPT_SPARSE_START, 0)
            run_step()
            self.fence(sim)
            k = max(1, min(steps, 3))
            t1 = time.perf_counter()
            for _ in range(k):
                run_step()
            self.fence(sim)
            dt = time.perf_counter() - t1
            sim.set_option(_qlib.OPT_SPARSE_START, 1)
            full_sweeps = {"ms_per_step": 1e3 * dt / k, "value": depth * k / dt, "steps": k}

        # north_star target "single-qubit gate apply at n=30": a short dense-1q probe on the same (now dense) state
        probe = None
        if with_1q_probe and dist is None and probe_q is None and n >= 8:
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

        res = {"workload": workload, "n": n, "gates": gates, "elapsed": elapsed, "steps": steps, "stats": stats,
               "norm2": norm2, "probe": probe, "fuse": fuse, "tuning": tuning, "full_sweeps": full_sweeps}
        if dist is not None:
            xs, xb = sim.exchange_seconds / steps, sim.exchange_bytes / steps
            res["exchange"] = {"per_step": sim.plan.exchanges,
                               "qubits_swapped": [len(s[1]) for s in sim.plan.steps if s[0] == "exchange"],
                               "bytes_sent_per_rank_per_step": xb, "seconds_per_step": xs,
                               "xgmi_gbps_per_rank": (xb / xs / 1e9) if xs > 0 else None,
                               "backend": getattr(sim, "exchange_backend", None),
                               "predicted": getattr(sim, "exchange_prediction", None),
                               "note": "seconds: pack + send/recv of rank 0 (rccl-native: HIP events on the engine's stream; "
                                       "torch.distributed: host clock incl. the stream hand-offs)"}
        sim.close()
        del sim
        return res

    @staticmethod
    def roofline_of(stats):
        kernels = {k: v for k, v in stats["kernels"].items() if v["launches"] and k != "init"}
        dom = max(kernels, key=lambda k: kernels[k]["ms"]) if kernels else None
        if not dom or kernels[dom]["ms"] <= 0:
            return None
        achieved = kernels[dom]["bytes"] / (kernels[dom]["ms"] * 1e-3) / 1e9
        return {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBPS, "traffic": None, "launches": kernels[dom]["launches"],
                "avg_launch_ms": kernels[dom]["ms"] / kernels[dom]["launches"],
                "algorithmic_bytes_per_launch": kernels[dom]["bytes"] / kernels[dom]["launches"]}


def main():
    args = parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args))
    b = Bench(args)
    if b.world != args.gpus:
        sys.exit(f"WORLD_SIZE={b.world} does not match --gpus {args.gpus}")
    if args.precision == 32 and (b.world > 1 or args.force_sharded):
        sys.exit("--precision 32 is single-GPU only (shards and clusters are fp64)")
    b.setup()
    if args.wisdom and os.path.exists(args.wisdom):
        from gpu_quantum_simulator_amd import _lib as _qlib
        _qlib.load().qsim_tune_table_load(args.wisdom.encode())

    n = args.qubits + (int(round(math.log2(b.world))) if args.scaling == "weak" else 0)
    seed = args.seed if args.seed is not None else 20240117 + n
    opts = {k: v for k, v in (("tile_bits", args.tile_bits), ("tile_low_bits", args.tile_low_bits),
                              ("tile_max_ops", args.tile_max_ops), ("grid_cap", args.grid_cap),
                              ("tile_threads", args.tile_threads), ("pingpong", args.pingpong)) if v is not None}
    head = b.measure(n, args.depth, args.vocabulary, seed, args.steps, args.warmup, args.fuse, opts,
                     probe_q=args.probe, with_1q_probe=True)

    # other register sizes, same generator and defaults (north_star: n = 24/28/30/32)
    sizes = []
    if args.sizes and args.probe is None and args.precision == 64:
        want = [int(x) for x in args.sizes.split(",") if x.strip()]
        weak = args.qubits + int(round(math.log2(b.world)))  # the same shard size as the 1-GPU headline (n=33 on 8 GPUs)
        if b.world > 1 and weak not in want:
            want.append(weak)
        for m in want:
            if m == n or m - int(round(math.log2(b.world))) < 14:
                continue
            r = b.measure(m, args.depth, args.vocabulary, 20240117 + m, args.size_steps, 1, args.fuse, opts)
            sizes.append(r)

    if args.wisdom and b.rank == 0:
        from gpu_quantum_simulator_amd import _lib as _qlib
        _qlib.load().qsim_tune_table_save(args.wisdom.encode())
    if b.rank == 0:
        stats = head["stats"]
        ms_per_step = 1e3 * head["elapsed"] / args.steps
        value = args.depth * args.steps / head["elapsed"]
        roof = Bench.roofline_of(stats)
        if roof:
            default = (args.precision == 64 and args.probe is None and n == 30 and args.depth == 1000 and head["fuse"] == 3
                       and args.gpus == 1 and not opts and args.vocabulary == "all")
            roof["traffic"], roof["traffic_source"] = pmc_traffic(roof["kernel"], default)
        total_kernel_ms = sum(v["ms"] for v in stats["kernels"].values())
        out = {
            "metric": "gate-applies/sec", "value": value, "unit": "gate-applies/s", "n_gpus": args.gpus,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": args.scaling, "vs_baseline": None, "dtype": "f64" if args.precision == 64 else "f32", "data": "synthetic",
            "config": {"workload": head["workload"], "qubits": n, "gate_statements": args.depth, "fuse": head["fuse"],
                       "state_bytes": (16 if args.precision == 64 else 8) * (1 << n), "parallelism": f"shard{args.gpus}", **opts},
            "n_ranks_seen": b.dist.get_world_size() if b.dist is not None else 1,
            "hbm_gbps_all_kernels": stats["algorithmic_bytes"] / (total_kernel_ms * 1e-3) / 1e9 if total_kernel_ms else None,
            "launches_per_step": stats["launches"] / args.steps,
            "kernel_ms_per_step": {k: v["ms"] / args.steps for k, v in stats["kernels"].items() if v["launches"]},
            "norm2": head["norm2"],
            "roofline": roof,
            "roofline_1q_probe": head["probe"],
            "geometry_planning": head["tuning"],
        }
        if head.get("full_sweeps"):
            tile = stats["kernels"].get("tile")
            swept = (tile["bytes"] / (tile["launches"] * 2.0 * (16 if args.precision == 64 else 8) * (1 << n))) if tile and tile["launches"] else None
            out["sparse_start"] = {"note": "after a reset, tile passes visit only the tiles inside the state's support (the first pass "
                                           "writes one tile, the full sweeps start once every qubit has been inside a tile); "
                                           "same amplitudes, fewer bytes.  with_full_sweeps: the same steps with the option off",
                                   "tile_bytes_vs_full_sweeps": swept, "with_full_sweeps": head["full_sweeps"]}
        if "exchange" in head:
            out["exchange"] = head["exchange"]
        cpu = not args.no_cpu_baseline and args.gpus == 1 and b.dist is None
        if cpu:
            out["cpu_baseline"] = cpu_baseline(n, head["gates"], args.cpu_seconds)
        if cpu and args.probe is None and n == 30:
            out["exchange_model"] = exchange_model(args.depth, args.vocabulary, ms_per_step)
        if sizes:
            table = []
            for r in sizes:
                rf = Bench.roofline_of(r["stats"])
                row = {"qubits": r["n"], "state_bytes": 16 << r["n"], "workload": r["workload"],
                       "value": args.depth * r["steps"] / r["elapsed"], "unit": "gate-applies/s",
                       "ms_per_step": 1e3 * r["elapsed"] / r["steps"], "steps": r["steps"],
                       "launches_per_step": r["stats"]["launches"] / r["steps"], "norm2": r["norm2"],
                       "geometry_planning": r["tuning"], "with_full_sweeps": r.get("full_sweeps"),
                       "roofline": None if rf is None else {k: rf[k] for k in ("kernel", "achieved", "frac", "avg_launch_ms")}}
                if (16 << r["n"]) <= (256 << 20) and row["roofline"]:
                    row["roofline"]["note"] = "the state fits the 256 MiB Infinity Cache: passes run from cache, the HBM roofline does not bound them"
                if "exchange" in r:
                    row["exchange"] = r["exchange"]
                if cpu:
                    # ~10 s of CPU work per size: about 170 gates at n=24, 10 at n=28, 1 at n=32 (one gate there takes longer)
                    row["cpu_baseline"] = cpu_baseline(r["n"], r["gates"], 10.0 if r["n"] < 31 else 1.0)
                table.append(row)
            out["sizes"] = table
        print(json.dumps(out), flush=True)
    if b.dist is not None:
        b.dist.barrier()
        b.dist.destroy_process_group()


if __name__ == "__main__":
    main()
