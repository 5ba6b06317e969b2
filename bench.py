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
    ap.add_argument("--no-precision32", action="store_true", help="skip the extra fp32 record of the default run")
    ap.add_argument("--no-one-shot", action="store_true", help="skip the cold figures (bin/qsim as a child process)")
    ap.add_argument("--no-full-sweeps", action="store_true",
                    help="skip the extra steps with QSIM_OPT_SPARSE_START off (profile runs: every launch then is one of the timed kind)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    return ap.parse_args(argv)


KERNEL_SYMBOL = {"tile": "k_tile<12, 512, false, false>", "gate1": "k_gate1_hi<4, false>",
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


def cpu_baseline(n, gates, budget_s, min_gates=3):
    """quantum_simulator.c's hot loops (:81-106) on this host, 1 thread, on a bounded sample of the same circuit.  `kind`
    "reference": execute_single_qubit_gate / execute_cnot of the REAL quantum_simulator.c, compiled in the build container from
    the sources where they lie (oracle/Makefile ref -> oracle/_ref/libqsref.so: a build output that travels with the snapshot,
    git-ignored); "port": the oracle's restatement of the same loops (byte-identical results, tests/test_oracle_golden.py) where
    that file is absent."""
    import ctypes

    import numpy as np
    from oracle import oracle  # the checker, used here only as the timed CPU baseline

    kind = "reference" if oracle.have_reference() else "port"
    need = 16 << n
    if need + (8 << 30) > host_ram_bytes():
        return {"value": None, "unit": "gate-applies/s", "cores": 1, "host_cores": os.cpu_count(), "kind": kind,
                "sample": f"not run: the 2^{n} state ({need >> 30} GiB) does not fit this host's RAM"}
    oracle.build(with_reference=False)
    dp = ctypes.POINTER(ctypes.c_double)
    state = np.zeros(1 << n, dtype=np.complex128)
    state[0] = 1.0
    state[1:] = 0.0  # touch every page before the clock starts
    sp = state.view(np.float64).ctypes.data_as(dp)
    from gpu_quantum_simulator_amd import gate_matrix
    if kind == "reference":
        R = oracle.reference_lib()
        apply_cx, apply_1q = R.execute_cnot, R.execute_single_qubit_gate
    else:
        L = oracle.lib()
        apply_cx, apply_1q = L.oracle_apply_cx, L.oracle_apply_1q
    done = 0
    t0 = time.perf_counter()
    for g in gates:
        if g[0] == "cx":
            apply_cx(sp, n, g[1], g[2])
        else:
            tok = f"rz({g[1]!r})" if g[0] == "rz" else g[0]
            u = np.ascontiguousarray(gate_matrix(tok).T.reshape(4))  # symmetric anyway
            apply_1q(sp, n, u.view(np.float64).ctypes.data_as(dp), g[-1])
        done += 1
        if done >= min_gates and time.perf_counter() - t0 >= budget_s:
            break
    dt = time.perf_counter() - t0
    del state
    return {"value": done / dt, "unit": "gate-applies/s", "cores": 1, "host_cores": os.cpu_count(), "kind": kind,
            "sample": f"first {done} gate statements of the same n={n} circuit, {dt:.1f} s, 1 thread, state resident in host RAM"
                      + ("; the compiled quantum_simulator.c itself (oracle/_ref)" if kind == "reference" else "; the oracle's restatement of quantum_simulator.c:81-106")}


def exchange_model(depth, vocabulary, tile_gbps, ms_per_step_n30, link_gbps=50.0, pack_gbps=5000.0):
    """What the plans predict for the multi-GPU configs of BASELINE.json, from host work alone (no GPU): per config the
    passes rank 0's engine will schedule between the exchanges — counted in sweeps of the shard, the sparse first passes
    by the fraction they visit (ShardPlan.local_sweeps: the very scheduler the engine runs) — priced at the tile kernel's
    rate measured in THIS run, and the exchanges priced per link: a k-qubit swap puts 2^-k of the shard on each of 2^k - 1
    links at once, and the blocks of ranks that hold nothing yet stay home.  The re-layout of an exchange rides on the last
    tile pass in front of it (qsim_flush_pack), so it costs no sweep; `pack_gbps` only prices the exchanges whose last pass
    cannot take it (none in these plans)."""
    from gpu_quantum_simulator_amd import circuits, gate_matrix
    from gpu_quantum_simulator_amd.distributed import ShardPlan, normalize_gates
    rows = []
    for n, P in ((30, 2), (30, 4), (30, 8), (33, 8)):
        gates = normalize_gates(circuits.random_gates(n, depth, 20240117 + n, vocabulary), gate_matrix)
        p = P.bit_length() - 1
        worst = None
        for rank in sorted({0, P - 1}):  # rank 0 carries the sparse start, the others join at the first exchange
            ls = ShardPlan(n, p, gates, rank).local_sweeps()
            if worst is None or ls["sweeps"] > worst["sweeps"]:
                worst = ls
        shard_bytes = 16.0 * (1 << (n - p))
        local_ms = 1e3 * worst["sweeps"] * 2.0 * shard_bytes / (tile_gbps * 1e9)
        sent = sum(x["blocks_sent"] * x["block_bytes"] for x in worst["exchanges"])
        xms = sum((1e3 * x["block_bytes"] / (link_gbps * 1e9)) if (x["blocks_sent"] or x["blocks_received"]) else 0.0 for x in worst["exchanges"])
        ideal = ms_per_step_n30 * (2.0 ** (n - 30)) / P
        overlap = _price_chunked_exchanges(worst, n - p, shard_bytes, tile_gbps, link_gbps)
        rows.append({"qubits": n, "ranks": P, "exchanges": len(worst["exchanges"]),
                     "qubits_swapped": [x["qubits"] for x in worst["exchanges"]],
                     "local_passes": worst["passes"], "local_sweeps_of_the_shard": worst["sweeps"],
                     "bytes_sent_per_rank": sent, "predicted_exchange_ms": xms,
                     "predicted_local_ms": local_ms, "predicted_step_ms": local_ms + xms,
                     "ideal_ms": ideal, "vs_ideal": (local_ms + xms) / ideal,
                     "overlappable_sweeps": overlap["overlappable_sweeps"], "overlap_per_exchange": overlap["per_exchange"],
                     "predicted_step_ms_overlapped": local_ms + xms - overlap["saved_ms"],
                     "vs_ideal_overlapped": (local_ms + xms - overlap["saved_ms"]) / ideal})
    return {"assumptions": {"link_gbps_per_direction": link_gbps, "tile_kernel_gbps_measured_in_this_run": tile_gbps,
                            "note": "xGMI link ~76.8 GB/s per direction peak (7 x ~153 GB/s bidirectional per GPU), 65 % assumed; "
                                    "no overlap of exchange and local passes in predicted_step_ms; local term = the shard's own schedule "
                                    "(dense after the first exchange, sparse before it) at the measured tile-kernel rate; ideal = this "
                                    "run's 1-GPU n=30 step x 2^(n-30) / ranks; predicted_step_ms_overlapped = the same with up to three "
                                    "passes on either side of every exchange run chunk by chunk beside the transfer (priced, not built: "
                                    "_price_chunked_exchanges)"},
            "configs": rows}


def _price_chunked_exchanges(ls, m, shard_bytes, tile_gbps, link_gbps, max_passes=3, max_chunk_bits=3):
    """Host only: what running the passes around an exchange CHUNK BY CHUNK beside the transfer would save (not built: priced).
    A k-qubit exchange on 2^k ranks puts one block on each of 2^k - 1 links AT ONCE, so sending block b while block b + 1 is
    computed hides nothing — every link still carries its block from the moment the last block is ready.  What pipelines is a
    chunk index over c index bits that are neither swapped out nor inside the tile of the passes involved: the last i passes of
    the segment in front run chunk by chunk (the packing pass writes chunk x of EVERY block, then all links send their piece of
    chunk x), and the first j passes behind start on chunk x once every peer's piece of it has arrived — also passes on the
    incoming qubits, because a chunk holds all values of those.  Three-stage pipeline over C = 2^c chunks:
        t/C + x/C + h/C + (C - 1) max(t, x, h)/C     instead of     t + x + h.
    Chosen per exchange: i, j <= max_passes and the c <= max_chunk_bits free bits, whichever saves most under the model
    (the schedules are the shards' own, as qsim_flush builds them: no pass was moved to make room)."""
    steps = ls["steps"]
    sweep_ms = 1e3 * 2.0 * shard_bytes / (tile_gbps * 1e9)
    out = {"overlappable_sweeps": 0.0, "saved_ms": 0.0, "per_exchange": []}
    for ex in ls["exchanges"]:
        i_step = ex["step"]
        x_ms = (1e3 * ex["block_bytes"] / (link_gbps * 1e9)) if (ex["blocks_sent"] or ex["blocks_received"]) else 0.0
        before = steps[i_step - 1]["per_pass"] if i_step > 0 and isinstance(steps[i_step - 1], dict) else []
        after = steps[i_step + 1]["per_pass"] if i_step + 1 < len(steps) and isinstance(steps[i_step + 1], dict) else []
        lsel = ex["local_positions_out"]
        k = len(lsel)
        rest = [b for b in range(m) if b not in lsel]           # old local positions that stay, ascending
        new_pos = {b: idx for idx, b in enumerate(rest)}         # ... and where they sit afterwards (the incoming qubits take the top k)
        best = {"saved_ms": 0.0, "tail_passes": 0, "head_passes": 0, "chunk_bits": 0, "tail_sweeps": 0.0, "head_sweeps": 0.0}
        for i in range(0, min(max_passes, len(before)) + 1):
            for j in range(0, min(max_passes, len(after)) + 1):
                if i + j == 0 or x_ms == 0.0:
                    continue
                busy = 0
                for ps in before[len(before) - i:]:
                    busy |= ps["tile_mask"]
                free = [b for b in rest if not (busy >> b) & 1]
                for ps in after[:j]:
                    free = [b for b in free if not (ps["tile_mask"] >> new_pos[b]) & 1]
                c = min(max_chunk_bits, len(free))
                if c == 0:
                    continue
                t = sum(ps["sweeps"] for ps in before[len(before) - i:]) * sweep_ms
                h = sum(ps["sweeps"] for ps in after[:j]) * sweep_ms
                C = float(1 << c)
                piped = (t + x_ms + h) / C + (C - 1.0) * max(t, x_ms, h) / C
                saved = (t + x_ms + h) - piped
                if saved > best["saved_ms"]:
                    best = {"saved_ms": saved, "tail_passes": i, "head_passes": j, "chunk_bits": c,
                            "tail_sweeps": t / sweep_ms, "head_sweeps": h / sweep_ms, "exchange_ms": x_ms}
        out["per_exchange"].append(best)
        out["saved_ms"] += best["saved_ms"]
        out["overlappable_sweeps"] += best["tail_sweeps"] + best["head_sweeps"]
    return out


def _selfcheck_once(dist, device, shard_factory, qubits, depth, seed):
    import tempfile

    import numpy as np
    from gpu_quantum_simulator_amd import circuits
    from gpu_quantum_simulator_amd.distributed import ShardedSimulator
    world, rank = dist.get_world_size(), dist.get_rank()
    n = max(qubits, (world.bit_length() - 1) + 12)
    gates = circuits.random_gates(n, depth, seed, "all")
    sim = ShardedSimulator(n, gates, device=device, shard_factory=shard_factory)
    for _ in range(2):  # the second step starts from a used state and replays cached plans
        sim.run_step()
    got = sim.gather_logical()
    norm2 = sim.norm2()
    rec = {"qubits": n, "gates": depth, "exchanges": sim.plan.exchanges, "exchange_backend": sim.exchange_backend,
           "tolerance": 1e-10, "norm2": norm2}
    if getattr(sim.shard, "comm", None) is not None:
        rec["relayouts_fused_separate"] = list(sim.shard.comm.pack_counts())
    ok = 1
    if rank == 0:
        from oracle import oracle  # the checker
        oracle.build(with_reference=False)
        with tempfile.TemporaryDirectory() as d:
            path = circuits.write_qasm(os.path.join(d, "selfcheck.qasm"), n, gates)
            _, want, _, _ = oracle.run_qasm(path)
        rec["max_abs_err"] = float(np.max(np.abs(got - want)))
        if not (rec["max_abs_err"] < 1e-10 and abs(norm2 - 1.0) < 1e-10):
            ok = 0
    sim.close()
    return rec, ok


def run_selfcheck(dist, device, shard_factory=None, require_native=True, qubits=20, depth=400, seed=777, hang_seconds=300.0):
    """Before a multi-rank run is timed: a small sharded circuit on the SAME ranks against the oracle (the checker leg,
    outside any timed region).  Every exchange path the timed run will take — planner, fused re-layout, the library's own
    RCCL send/recv, support bookkeeping — has to reproduce quantum_simulator.c's amplitudes within 1e-10, on every rank's
    data, or the bench stops: SystemExit(3) on every rank, no number printed.  Should the library's own RCCL exchange give
    wrong amplitudes while the torch.distributed form of the same exchange gives right ones, the run goes on with that
    form and says so (`exchange_backend`, `native_exchange_failed`); a run that stops moving for hang_seconds is ended
    with exit code 4 rather than left to the caller's time limit."""
    import threading

    import torch
    world, rank = dist.get_world_size(), dist.get_rank()
    done = threading.Event()

    def watchdog():
        if not done.wait(hang_seconds):
            sys.stderr.write(f"bench.py: rank {rank}: the sharded self-check did not finish within {hang_seconds:.0f} s (an exchange is stuck); giving up\n")
            sys.stderr.flush()
            os._exit(4)

    threading.Thread(target=watchdog, daemon=True).start()

    def agree(ok):
        flag = torch.tensor([ok], dtype=torch.int32)
        if dist.get_backend() == "nccl":
            flag = flag.cuda()
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        return bool(int(flag.item()))

    try:
        rec, ok = _selfcheck_once(dist, device, shard_factory, qubits, depth, seed)
        passed = agree(ok)
        if not passed and rec["exchange_backend"] == "rccl-native":
            first = rec
            os.environ["QSIM_EXCHANGE"] = "torch"  # every rank takes this branch together (the verdict was agreed on)
            rec, ok = _selfcheck_once(dist, device, shard_factory, qubits, depth, seed)
            passed = agree(ok)
            rec["native_exchange_failed"] = {k: first.get(k) for k in ("max_abs_err", "norm2", "relayouts_fused_separate")}
        if passed and require_native and world > 1 and rec["exchange_backend"] != "rccl-native" and "native_exchange_failed" not in rec:
            rec["note"] = "the library's own RCCL communicator could not be created: exchanges go through torch.distributed"
        rec["passed"] = passed
    finally:
        done.set()
    if not rec["passed"]:
        if rank == 0:
            sys.stderr.write("bench.py: sharded self-check FAILED: " + json.dumps(rec) + "\n")
        dist.barrier()
        raise SystemExit(3)
    return rec


def one_shot_cli(n, gates, runs=5):
    """The reference's own protocol (quantum_simulator.c:143,244-248; tester.bash:5-14: five runs, their mean): ONE process, ONE
    circuit, the printed seconds run from the header to the last gate.  `bin/qsim <file> 1` as a child process of this one —
    fresh process, no wisdom file, no planning step: allocation, |0...0>, the schedule choice that fits beside the allocation,
    scheduling, block preparation, uploads and every pass are inside the printed time (only the creation of the HIP context,
    process start-up, is not).  `runs` children one after the other; printed_seconds / value are the MEDIAN run's, with the
    minimum and the mean beside them (one sample moved 2 x between boxes: hipMalloc of 16 GiB takes 0.04-0.25 s)."""
    import statistics
    import subprocess
    import tempfile
    from gpu_quantum_simulator_amd import _lib, circuits
    samples = []
    with tempfile.TemporaryDirectory() as d:
        path = circuits.write_qasm(os.path.join(d, "one_shot.qasm"), n, gates)
        env = {k: v for k, v in os.environ.items() if not k.startswith("QSIM_")}
        env["QSIM_STATS"] = "1"
        for _ in range(runs):
            t0 = time.perf_counter()
            p = subprocess.run([_lib.CLI_PATH, path, "1"], capture_output=True, text=True, env=env, timeout=900)
            one = {"exit_code": p.returncode, "process_wall_seconds": time.perf_counter() - t0}
            try:
                one["printed_seconds"] = float(p.stdout.split()[0])
                stats = [ln for ln in p.stderr.splitlines() if ln.startswith("{")]
                if stats:
                    st = json.loads(stats[-1])
                    one["launches"] = st.get("launches")
                    one["breakdown_s"] = {k: st.get(k) for k in ("parse_s", "allocate_s", "schedule_and_launch_s", "wait_s")}
            except (ValueError, IndexError):
                one["error"] = (p.stdout + p.stderr)[-400:]
            samples.append(one)
    rec = {"command": "bin/qsim <file> 1", "runs": runs, "exit_code": max(x["exit_code"] for x in samples)}
    good = sorted((x for x in samples if "printed_seconds" in x), key=lambda x: x["printed_seconds"])
    if good:
        med = good[len(good) // 2]
        rec.update({"printed_seconds": med["printed_seconds"], "value": len(gates) / med["printed_seconds"], "unit": "gate-applies/s",
                    "printed_seconds_min": good[0]["printed_seconds"], "value_of_the_fastest_run": len(gates) / good[0]["printed_seconds"],
                    "printed_seconds_mean": statistics.fmean(x["printed_seconds"] for x in good),
                    "printed_seconds_all": [x["printed_seconds"] for x in samples if "printed_seconds" in x],
                    "process_wall_seconds": med["process_wall_seconds"], "launches": med.get("launches"),
                    "breakdown_s": med.get("breakdown_s")})
    else:
        rec["error"] = samples[-1].get("error", "no run printed a time")
    return rec


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

    def measure(self, n, depth, vocabulary, seed, steps, warmup, fuse, opts, probe_q=None, with_1q_probe=False, precision=None, cold=False):
        """Builds the simulator for one register size, runs `warmup` + `steps` timed steps, returns the numbers.
        cold: before anything is planned, the very first step on the fresh state is timed on its own (empty plan cache,
        no schedule choice, no measured geometries: what a one-shot run pays), then the second one (cached plan)."""
        from gpu_quantum_simulator_amd import Circuit, Simulator, circuits
        args, dist, torch = self.args, self.dist, self.torch
        precision = precision or args.precision
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
            sim = Simulator(n, self.local_rank, fuse=fuse, profile=True, precision=precision, **opts)

            def run_step():
                sim.reset()
                sim.run(circuit)
                sim.flush()

        cold_ms = None
        if cold and dist is None:
            # a dozen h gates first: the engine allocates its second buffer and HIP loads the tile kernel on the first tile pass —
            # 0.04-0.5 s of hipMalloc for 16 GiB depending on the box, which would bury what the figure is about (scheduling and
            # running the circuit with nothing planned); the CLI figure of `one_shot` includes every allocation
            from gpu_quantum_simulator_amd import gate_matrix
            for q in range(3, min(n, 15)):
                sim.apply_1q(gate_matrix("h"), q)
            self.fence(sim)
            cold_ms = []
            for _ in range(2):
                t0 = time.perf_counter()
                run_step()
                self.fence(sim)
                cold_ms.append(1e3 * (time.perf_counter() - t0))

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
            if not args.no_full_sweeps:
                # the comparison run with every pass a full sweep schedules differently (no sparse phase): the same planning for it
                dense = sim.tune(circuit, args.tune_candidates, args.tune_ms / 2, dense_start=True)
                tuning["full_sweeps_planning_seconds"] = dense["seconds"]
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
               "norm2": norm2, "probe": probe, "fuse": fuse, "tuning": tuning, "full_sweeps": full_sweeps, "cold_ms": cold_ms,
               "precision": precision, "vocabulary": vocabulary}
        if dist is not None:
            xs, xb = sim.exchange_seconds / steps, sim.exchange_bytes / steps
            res["exchange"] = {"per_step": sim.plan.exchanges,
                               "qubits_swapped": [len(s[1]) for s in sim.plan.steps if s[0] == "exchange"],
                               "bytes_sent_per_rank_per_step": xb, "seconds_per_step": xs,
                               "xgmi_gbps_per_rank": (xb / xs / 1e9) if xs > 0 else None,
                               "backend": getattr(sim, "exchange_backend", None),
                               "relayouts_fused_separate": (list(sim.shard.comm.pack_counts()) if getattr(sim.shard, "comm", None) is not None else None),
                               "predicted": getattr(sim, "exchange_prediction", None),
                               "note": "seconds: rank 0's transfers (rccl-native: HIP events on the engine's stream around the ncclGroup; the "
                                       "re-layout rides on the last tile pass before the exchange and is not in it, a separate pack kernel is; "
                                       "torch.distributed fallback: host clock incl. pack and the stream hand-offs); bytes: blocks that really "
                                       "travelled (ranks that hold nothing yet send nothing)"}
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
    selfcheck = run_selfcheck(b.dist, b.local_rank) if b.dist is not None else None
    if args.wisdom and os.path.exists(args.wisdom):
        from gpu_quantum_simulator_amd import _lib as _qlib
        _qlib.load().qsim_tune_table_load(args.wisdom.encode())

    n = args.qubits + (int(round(math.log2(b.world))) if args.scaling == "weak" else 0)
    seed = args.seed if args.seed is not None else 20240117 + n
    opts = {k: v for k, v in (("tile_bits", args.tile_bits), ("tile_low_bits", args.tile_low_bits),
                              ("tile_max_ops", args.tile_max_ops), ("grid_cap", args.grid_cap),
                              ("tile_threads", args.tile_threads), ("pingpong", args.pingpong)) if v is not None}
    default_workload = (args.precision == 64 and args.probe is None and n == 30 and args.depth == 1000 and args.fuse == 3 and args.gpus == 1
                        and not opts and args.vocabulary == "all" and b.dist is None)
    head = b.measure(n, args.depth, args.vocabulary, seed, args.steps, args.warmup, args.fuse, opts,
                     probe_q=args.probe, with_1q_probe=True, cold=default_workload and not args.no_one_shot and not (args.wisdom and os.path.exists(args.wisdom)))

    # other register sizes, same generator and defaults (north_star: n = 24/28/30/32)
    sizes = []
    if args.sizes and args.probe is None and args.precision == 64:
        want = [int(x) for x in args.sizes.split(",") if x.strip()]
        weak = args.qubits + int(round(math.log2(b.world)))  # the same shard size as the 1-GPU headline (n=33 on 8 GPUs)
        if b.world > 1 and weak not in want:
            want.append(weak)
        rows = [(m, args.vocabulary) for m in want]
        if b.world == 1 and 28 in want and args.vocabulary == "all":
            rows.insert(rows.index((28, "all")) + 1, (28, "clifford_t"))  # BASELINE configs[2] exactly: random Clifford+T, n=28, depth 1000
        for m, vocab in rows:
            if (m == n and vocab == args.vocabulary) or m - int(round(math.log2(b.world))) < 14:
                continue
            r = b.measure(m, args.depth, vocab, 20240117 + m, args.size_steps, 1, args.fuse, opts)
            sizes.append(r)

    # the fp32 state of the reference's CUDA variants (quantum_simulator_naive.cu:72-95,145-149) on the same circuit: an extra
    # record beside the fp64 headline, never the headline itself
    f32 = None
    if default_workload and not args.no_precision32:
        saved = args.precision
        f32 = b.measure(n, args.depth, args.vocabulary, seed, max(2, args.size_steps), 1, args.fuse, opts, precision=32)
        args.precision = saved

    if args.wisdom and b.rank == 0:
        from gpu_quantum_simulator_amd import _lib as _qlib
        _qlib.load().qsim_tune_table_save(args.wisdom.encode())
    if b.rank == 0:
        stats = head["stats"]
        ms_per_step = 1e3 * head["elapsed"] / args.steps
        value = args.depth * args.steps / head["elapsed"]
        roof = Bench.roofline_of(stats)
        if roof:
            roof["traffic"], roof["traffic_source"] = pmc_traffic(roof["kernel"], default_workload)
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
        if selfcheck is not None:
            out["selfcheck"] = selfcheck
        if default_workload and not args.no_one_shot:
            cold = head.get("cold_ms")
            out["one_shot"] = {"note": "what ONE run of the drop-in costs, the reference's protocol (quantum_simulator.c:143,244-248, "
                                       "tester.bash:12): no planning step, no wisdom file, plan cache empty; `value` above is the steady "
                                       "state of a circuit that runs again and again (cached plan, chosen schedule, measured tile-bit orders)",
                               "cli": one_shot_cli(n, head["gates"]),
                               "in_process_first_step_ms": cold[0] if cold else None,
                               "in_process_first_step_value": (args.depth / (cold[0] * 1e-3)) if cold else None,
                               "in_process_second_step_ms": cold[1] if cold else None}
        if f32 is not None:
            rf32 = Bench.roofline_of(f32["stats"])
            out["precision32"] = {"dtype": "f32", "workload": f32["workload"], "state_bytes": 8 << n,
                                  "value": args.depth * f32["steps"] / f32["elapsed"], "unit": "gate-applies/s",
                                  "ms_per_step": 1e3 * f32["elapsed"] / f32["steps"], "steps": f32["steps"],
                                  "launches_per_step": f32["stats"]["launches"] / f32["steps"], "norm2": f32["norm2"],
                                  "with_full_sweeps": f32.get("full_sweeps"), "geometry_planning": f32["tuning"],
                                  "roofline": None if rf32 is None else {k: rf32[k] for k in ("kernel", "achieved", "frac", "avg_launch_ms", "peak", "unit", "bound")},
                                  "note": "fp32 complex AoS state (float2), the precision of quantum_simulator_naive.cu:72-95,145-149; agrees with "
                                          "the fp64 oracle to fp32 rounding (tests/test_gpu_fp32.py, 2e-5), not to 1e-10: never the headline"}
        cpu = not args.no_cpu_baseline and args.gpus == 1 and b.dist is None
        if cpu:
            out["cpu_baseline"] = cpu_baseline(n, head["gates"], args.cpu_seconds)
        if cpu and args.probe is None and n == 30 and roof and roof["kernel"] == "tile":
            out["exchange_model"] = exchange_model(args.depth, args.vocabulary, roof["achieved"], ms_per_step)
        if sizes:
            table = []
            for r in sizes:
                rf = Bench.roofline_of(r["stats"])
                row = {"qubits": r["n"], "state_bytes": 16 << r["n"], "workload": r["workload"],
                       "value": args.depth * r["steps"] / r["elapsed"], "unit": "gate-applies/s",
                       "ms_per_step": 1e3 * r["elapsed"] / r["steps"], "steps": r["steps"],
                       "launches_per_step": r["stats"]["launches"] / r["steps"], "norm2": r["norm2"],
                       "geometry_planning": r["tuning"], "with_full_sweeps": r.get("full_sweeps"),
                       "roofline": None if rf is None else {k: rf[k] for k in ("kernel", "achieved", "frac", "avg_launch_ms", "peak", "unit", "bound")}}
                if (16 << r["n"]) <= (256 << 20) and row["roofline"]:
                    row["roofline"]["note"] = "the state fits the 256 MiB Infinity Cache: passes run from cache, the HBM roofline does not bound them"
                if "exchange" in r:
                    row["exchange"] = r["exchange"]
                if cpu:
                    # ~10 s of CPU work per size: about 170 gates at n=24, 10 at n=28; at n=32 one gate takes ~6 s: three of them
                    if r["vocabulary"] == args.vocabulary:
                        row["cpu_baseline"] = cpu_baseline(r["n"], r["gates"], 10.0 if r["n"] < 31 else 1.0, min_gates=3)
                table.append(row)
            out["sizes"] = table
        print(json.dumps(out), flush=True)
    if b.dist is not None:
        b.dist.barrier()
        b.dist.destroy_process_group()


if __name__ == "__main__":
    main()
