"""The sharded path without GPUs: planner + logical->physical map + pack layout + exchange pattern.

(1) VirtualCluster with CPU shards (oracle loops) for P = 2, 4, 8 must equal the oracle on the whole
    register.  (2) The same through torch.distributed with the gloo backend, world_size 2 and 4, one
    process per rank — the code path bench.py takes with nccl on real GPUs."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from gpu_quantum_simulator_amd import circuits, gate_matrix
from gpu_quantum_simulator_amd.distributed import (ShardPlan, ShardedSimulator, VirtualCluster, normalize_gates,
                                                   logical_from_physical, peers_of, physical_index)
from cpu_shard import CpuShard
from py_shard_plan import PyShardPlan

TOL = 1e-12


def _oracle_state(oracle, tmp_path, n, gates):
    path = circuits.write_qasm(str(tmp_path / f"c{n}.qasm"), n, gates)
    return oracle.run_qasm(path)[1]


@pytest.mark.parametrize("world", [2, 4, 8])
@pytest.mark.parametrize("n,depth,seed,vocab", [(6, 150, 1, "all"), (9, 400, 2, "all"), (11, 500, 3, "clifford_t")])
def test_virtual_cluster_equals_oracle(oracle, tmp_path, world, n, depth, seed, vocab):
    gates = circuits.random_gates(n, depth, seed, vocab)
    want = _oracle_state(oracle, tmp_path, n, gates)
    vc = VirtualCluster(n, world, gates, shard_factory=CpuShard)
    vc.run()
    got = vc.gather_logical()
    assert np.max(np.abs(got - want)) < TOL
    assert vc.plans[0].exchanges >= 1  # these circuits cannot avoid communication


def test_plan_is_identical_across_ranks_and_counts_fraction():
    n, p = 12, 3
    gates = normalize_gates(circuits.random_gates(n, 600, 5, "all"), gate_matrix)
    plans = [ShardPlan(n, p, gates, r) for r in range(8)]
    shape = [(s[0],) + (s[1:] if s[0] == "exchange" else ()) for s in plans[0].steps]
    for pl in plans[1:]:
        assert [(s[0],) + (s[1:] if s[0] == "exchange" else ()) for s in pl.steps] == shape
        assert pl.final_pos == plans[0].final_pos
    assert sorted(plans[0].final_pos) == list(range(n))
    for s in plans[0].steps:
        if s[0] == "exchange":
            J, L = s[1], s[2]
            assert len(J) == len(L) and list(J) == sorted(J) and list(L) == sorted(L)
            assert all(0 <= j < p for j in J) and all(0 <= l < n - p for l in L)


def test_communication_free_gates_stay_local():
    """Diagonal gates and controls on global qubits never trigger an exchange."""
    n, p = 8, 2
    gates = [("h", q) for q in range(6)] + [("t", 7), ("rz", 0.3, 6), ("cx", 7, 2), ("cx", 6, 0), ("z", 7), ("s", 6)]
    norm = normalize_gates(gates, gate_matrix)
    pl = PyShardPlan(n, p, norm, rank=3, lookahead_free_start=False)
    assert pl.exchanges == 0 and [s[0] for s in pl.steps] == ["local"]
    kinds = [op[0] for op in pl.steps[0][1]]
    assert kinds.count("scale") == 4 and kinds.count("u1") == 6 + 2  # rank 3 has both control bits set
    pl0 = PyShardPlan(n, p, norm, rank=0, lookahead_free_start=False)
    assert [op[0] for op in pl0.steps[0][1]].count("u1") == 6  # controls clear, all scalars are 1


def test_free_initial_placement_avoids_first_exchange():
    n, p = 8, 2
    gates = [("h", 7), ("h", 6), ("cx", 7, 6)] + [("t", 0), ("z", 1)]
    norm = normalize_gates(gates, gate_matrix)
    assert ShardPlan(n, p, norm, 0).exchanges == 0 and PyShardPlan(n, p, norm, 0).exchanges == 0
    assert PyShardPlan(n, p, norm, 0, lookahead_free_start=False).exchanges == 1


def test_peers_and_index_helpers():
    mine, members = peers_of(0b101, [0, 2])
    assert mine == 0b11 and members == [0b000, 0b001, 0b100, 0b101]
    pos = [2, 0, 1]
    assert physical_index(0b001, pos) == 0b100
    phys = np.arange(8)
    assert list(logical_from_physical(phys, pos)) == [physical_index(x, pos) for x in range(8)]


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n, gates, want, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sim = ShardedSimulator(n, gates, shard_factory=CpuShard)
        for _ in range(2):  # a second step must start from a clean state and the identity map
            sim.run_step()
        norm = sim.norm2()
        amp5 = sim.amplitude(5)
        shards = [torch.zeros_like(sim.shard.state) for _ in range(world)]
        dist.all_gather(shards, sim.shard.state)
        phys = torch.cat(shards).numpy().reshape(-1).view(np.complex128)
        got = logical_from_physical(phys, sim.plan.final_pos)
        err = float(np.max(np.abs(got - want)))
        draws = np.concatenate([np.random.default_rng(7).uniform(0, 1, 40), [0.0, 1.0, 1.5]])
        picked = sim.sample(draws, block_bits=4, chunk_bits=5)  # small blocks / pieces: several of each per shard
        q.put((rank, err, norm, abs(amp5 - want[5]), sim.plan.exchanges, sim.exchange_bytes, draws, picked))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_gloo_ranks_equal_oracle(oracle, tmp_path, world):
    n = 10
    gates = circuits.random_gates(n, 300, 40 + world, "all")
    want = _oracle_state(oracle, tmp_path, n, gates)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, gates, want, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    cumul = oracle.cumulative(want, n)
    for rank, err, norm, aerr, exchanges, xbytes, draws, picked in results:
        assert err < TOL and abs(norm - 1.0) < 1e-12 and aerr < TOL
        assert exchanges >= 1 and xbytes > 0
        # the measurement post-path on the sharded state: same indices on every rank, equal to the oracle's search
        assert np.array_equal(picked, results[0][7])
        for r, g in zip(draws, picked):
            w = oracle.measure(cumul, n, float(r))
            assert int(g) == w or (abs(int(g) - w) == 1 and min(abs(cumul[w] - r), abs(cumul[int(g)] - r)) < 1e-13)


class _BrokenShard(CpuShard):
    """A shard whose re-layout is wrong (the blocks of an exchange come out in the wrong order)."""

    def pack(self, Lsel):
        super().pack(Lsel)
        k = len(Lsel)
        sc = self.scratch.view(1 << k, -1, 2)
        sc.copy_(sc.flip(0).clone())


def _selfcheck_worker(rank, world, port, broken, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    try:
        rec = bench.run_selfcheck(dist, 0, shard_factory=_BrokenShard if broken else CpuShard, require_native=False, qubits=13, depth=250)
        q.put((rank, "ok", rec))
    except SystemExit as e:
        q.put((rank, "exit", e.code))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("broken", [False, True])
def test_bench_selfcheck_on_two_gloo_ranks(broken):
    """bench.py --gpus N checks a small sharded circuit against the oracle on the same ranks before it times anything
    (run_selfcheck): here through two gloo ranks with CPU shards — once intact (passes, reports its error), once with a
    shard whose exchange layout is wrong: every rank stops with exit code 3 and no number would be printed."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_selfcheck_worker, args=(r, world, port, broken, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    if broken:
        assert [r[1:] for r in results] == [("exit", 3), ("exit", 3)]
    else:
        assert all(r[1] == "ok" and r[2]["passed"] for r in results)
        rec = results[0][2]
        assert rec["max_abs_err"] < 1e-12 and rec["exchanges"] >= 1 and rec["qubits"] == 13


@pytest.mark.parametrize("world", [2, 4, 8])
@pytest.mark.parametrize("n,depth,seed,vocab", [(8, 300, 11, "all"), (12, 600, 12, "all"), (16, 800, 13, "clifford_t"),
                                                (30, 1000, 20240147, "all")])
def test_cpp_planner_equals_python_restatement(world, n, depth, seed, vocab, monkeypatch):
    """libqsim's planner (csrc/dist.cpp: the product path of both the C host and distributed.py) against the
    independent pure-Python restatement in tests/py_shard_plan.py: same exchanges, same final qubit map, and the same
    per-rank local ops on every rank.  The restatement covers the planner proper (which gates can run, Belady eviction,
    the two placement policies); the hand-over of a segment's small last pass to the next segment (small_tail, which asks
    the engine's own scheduler) is switched off for the comparison and tested on its own below."""
    monkeypatch.setenv("QSIM_SHARD_TAIL", "0")
    gates = normalize_gates(circuits.random_gates(n, depth, seed, vocab), gate_matrix)
    p = world.bit_length() - 1
    for rank in sorted({0, 1, world - 1}):
        py = PyShardPlan(n, p, gates, rank=rank)
        cc = ShardPlan(n, p, gates, rank=rank)
        assert [s[0] for s in cc.steps] == [s[0] for s in py.steps]
        assert cc.final_pos == py.final_pos and cc.exchanges == py.exchanges
        for a, b in zip(cc.steps, py.steps):
            if a[0] == "exchange":
                assert a == b
            else:
                assert len(a[1]) == len(b[1])
                for x, y in zip(a[1], b[1]):
                    assert x[0] == y[0]
                    if x[0] == "cx":
                        assert x[1:] == y[1:]
                    elif x[0] == "u1":
                        assert x[1] == y[1] and np.array_equal(x[2], y[2])
                    else:
                        assert x[1] == y[1]


@pytest.mark.parametrize("world", [2, 8])
def test_small_last_pass_waits_for_the_exchange(oracle, tmp_path, world, monkeypatch):
    """csrc/dist.cpp small_tail: a segment's last pass, when the engine's scheduler would fill it with only a handful of
    gates, is handed on to the segment after the exchange.  The plan stays a valid execution order (CPU shards equal the
    oracle), no gate is lost or duplicated, and on the bench circuit the shards sweep their part of the register less often."""
    n = 14
    gates = circuits.random_gates(n, 900, 77, "all")
    want = _oracle_state(oracle, tmp_path, n, gates)
    norm = normalize_gates(gates, gate_matrix)
    p = world.bit_length() - 1
    counts = {}
    for tail in ("0", "24"):
        monkeypatch.setenv("QSIM_SHARD_TAIL", tail)
        vc = VirtualCluster(n, world, gates, shard_factory=CpuShard)
        vc.run()
        assert np.max(np.abs(vc.gather_logical() - want)) < TOL
        pl = vc.plans[0]
        # every gate statement appears exactly once over the local steps of a rank that sees them all (scalars equal to 1 and
        # conditional X gates whose control bit is 0 are dropped per rank, so count on the plan of the all-ones rank)
        full = ShardPlan(n, p, norm, world - 1)
        counts[tail] = sum(len(st[1]) for st in full.steps if st[0] == "local")
    assert counts["0"] == counts["24"]
    # the n = 30 bench circuit, host only: swept bytes of shard P-1's schedule, with and without the hand-over
    from gpu_quantum_simulator_amd import Circuit
    n = 30
    norm = normalize_gates(circuits.random_gates(n, 1000, 20240117 + n, "all"), gate_matrix)
    sweeps = {}
    for tail in ("0", "24"):
        monkeypatch.setenv("QSIM_SHARD_TAIL", tail)
        pl = ShardPlan(n, p, norm, world - 1)
        m, tot, sup, empty = n - p, 0.0, 0, True
        for i, st in enumerate(pl.steps):
            if st[0] == "exchange":
                ro = pl.handle.exchange_roles(i, world - 1)
                empty, sup = bool(ro["empty_after"]), ro["new_support"]
                continue
            if empty:
                continue
            c = Circuit.empty(m)
            for op in st[1]:
                if op[0] == "cx":
                    c.append_cx(op[1], op[2])
                else:
                    c.append_1q(op[2] if op[0] == "u1" else [[op[1], 0], [0, op[1]]], op[1] if op[0] == "u1" else 0)
            tot += c.plan(initial_support=sup)["algorithmic_bytes"] / (32.0 * (1 << m))
            sup = (1 << m) - 1
        sweeps[tail] = tot
    assert sweeps["24"] <= sweeps["0"]
    if world == 8:
        assert sweeps["24"] <= sweeps["0"] - 1.0  # 14 -> 12 sweeps of the shard


def test_hand_over_is_checked_gate_by_gate(oracle, tmp_path, monkeypatch):
    """The last pass the engine's scheduler proposes for a segment is one shard's view: a CX controlled by a shard-id bit is an X
    there and nothing on the shards where the bit is 0, and products that cancel in one view (h x h x = 1: the cluster vanishes
    and no longer orders anything) do not in another.  A stress case of round 3 (n = 16, 88 gates, 4 shards) handed over a CX and
    a diagonal gate across three later non-diagonal gates on the same qubit: amplitudes off by 0.04.  The planner now moves a gate
    only if it commutes, by what it is, with every later gate of the segment that stays.  That case, and 36 seeded others with
    shards of >= 2^12 amplitudes (where the hand-over is active), against the oracle."""
    monkeypatch.setenv("QSIM_SHARD_TAIL", "24")
    cases = [(16, 88, 9424, "all", 4)]
    rng = np.random.default_rng(5)
    for k in range(36):
        world = int(rng.choice([2, 4, 8]))
        cases.append((int(rng.integers(12 + world.bit_length() - 1, 16)), int(rng.integers(40, 260)), 31000 + k,
                      "all" if rng.random() < 0.7 else "clifford_t", world))
    moved_any = False
    for n, depth, seed, vocab, world in cases:
        path = circuits.random_circuit_file(str(tmp_path / "c.qasm"), n, depth, seed, vocab)
        _, want, _, _ = oracle.run_qasm(path)
        from gpu_quantum_simulator_amd import Circuit
        c = Circuit.from_file(path)
        gates = [c.gate(i) for i in range(len(c))]
        vc = VirtualCluster(n, world, gates, shard_factory=CpuShard)
        vc.run()
        assert np.max(np.abs(vc.gather_logical() - want)) < TOL, (n, depth, seed, vocab, world)
        monkeypatch.setenv("QSIM_SHARD_TAIL", "0")
        plain = [len(st[1]) for st in ShardPlan(n, world.bit_length() - 1, normalize_gates(gates, gate_matrix), world - 1).steps if st[0] == "local"]
        monkeypatch.setenv("QSIM_SHARD_TAIL", "24")
        moved_any = moved_any or [len(st[1]) for st in vc.plans[world - 1].steps if st[0] == "local"] != plain
    assert moved_any  # the hand-over did happen in some of them


def test_launcher_spawns_ranks_as_children():
    """bench.py --gpus N (no external launcher) starts its ranks through launch.spawn_ranks: here 2 gloo ranks of a
    probe script go through the same function; the parent relays rank 0's line and the exit code."""
    import json
    import sys
    from gpu_quantum_simulator_amd import launch
    probe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "rank_probe.py")
    cmd = launch.rank_command(2, probe, ["--x", "1"], port=12345)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"] and "--nproc-per-node=2" in cmd and cmd[-3:] == [probe, "--x", "1"]
    rc, out = launch.spawn_ranks(2, probe, ["--x", "1"], timeout=300)
    assert rc == 0, out
    line = json.loads([ln for ln in out.splitlines() if ln.startswith('{"metric"')][-1])
    assert line["world"] == 2 and line["sum"] == 3.0 and line["argv"] == ["--x", "1"]


def test_bench_plain_command_with_more_gpus_than_present_fails_cleanly():
    """`python bench.py --gpus 2` where fewer GPUs exist: non-zero exit, a clear message, no hang, no JSON line."""
    import subprocess
    import sys
    from gpu_quantum_simulator_amd import launch
    present = launch.count_gpus()
    if torch.cuda.device_count() >= 2 or (present or 0) >= 2:
        pytest.skip("two GPUs are present")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode != 0
    assert "2 GPUs requested" in p.stderr and "present" in p.stderr
    assert '{"metric"' not in p.stdout


def _pack_np(state, m, Lsel):
    """numpy restatement of k_pack: block index = the Lsel bits, rest = the other bits in order."""
    k = len(Lsel)
    d = np.arange(1 << m, dtype=np.int64)
    rest, blk = d & ((1 << (m - k)) - 1), d >> (m - k)
    keep = [b for b in range(m) if b not in Lsel]
    src = np.zeros_like(d)
    for i, b in enumerate(keep):
        src |= ((rest >> i) & 1) << b
    for i, b in enumerate(Lsel):
        src |= ((blk >> i) & 1) << b
    return state[src]


@pytest.mark.parametrize("world", [2, 4, 8, 16, 32])
@pytest.mark.parametrize("n,depth,seed,vocab", [(7, 120, 21, "all"), (10, 400, 22, "all"), (12, 500, 23, "clifford_t"), (11, 60, 24, "all"),
                                                (15, 260, 25, "all"), (15, 120, 26, "clifford_t"), (16, 88, 9424, "all")])  # shards of >= 2^12: the hand-over is active
def test_sparse_exchange_protocol_on_poisoned_memory(oracle, tmp_path, world, n, depth, seed, vocab):
    if world >= 16 and n < 10:
        pytest.skip("shards of fewer than 2^5 amplitudes")
    """What an exchange may leave out (csrc/dist.cpp roles_of; qsim_shard_plan_exchange_roles): shards that hold nothing
    neither pack nor send, blocks that are zero throughout are not received, and a receiver only ever looks inside its
    new support.  Modelled here with numpy shards whose memory is NaN wherever the protocol did not write: if a block that
    matters stayed home, or a support were too small, the NaNs (or wrong amplitudes) reach the gathered state."""
    gates = circuits.random_gates(n, depth, seed, vocab)
    want = _oracle_state(oracle, tmp_path, n, gates)
    p = world.bit_length() - 1
    m = n - p
    norm = normalize_gates(gates, gate_matrix)
    plans = [ShardPlan(n, p, norm, r) for r in range(world)]
    h = plans[0].handle
    idx = np.arange(1 << m, dtype=np.int64)
    full = (1 << m) - 1
    mem = [np.full(1 << m, np.nan + 0j) for _ in range(world)]
    support = [full] * world
    empty = [r != 0 for r in range(world)]
    mem[0][:] = 0
    mem[0][0] = 1.0
    saved_blocks = 0
    for i, st in enumerate(plans[0].steps):
        if st[0] == "local":
            for r in range(world):
                if empty[r]:
                    continue  # every gate maps the zero vector to itself: the engine drops the queue
                s = np.where((idx & ~support[r]) == 0, mem[r], 0)  # outside the support: zero by definition, whatever memory holds
                assert not np.isnan(s).any(), "the support covers memory the exchange never wrote"
                for op in plans[r].steps[i][1]:
                    if op[0] == "cx":
                        oracle.apply_cx(s, m, op[1], op[2])
                    elif op[0] == "u1":
                        oracle.apply_1q(s, m, np.asarray(op[2]).T, op[1])
                    else:
                        s *= op[1]
                mem[r], support[r] = s, full
            continue
        _, J, Lsel = st
        k = len(J)
        # widths the executing paths accept: a tile pass re-lays out <= 3 bits, qsim_flush_pack falls back to the pack
        # kernel up to 8, and the role masks (uint32, one bit per block) bound an exchange at 5 qubits = groups of 32 shards
        assert 1 <= k <= 5 and k <= p and len(set(J)) == k and len(set(Lsel)) == k and all(0 <= b < m for b in Lsel)
        blk = 1 << (m - k)
        roles = [h.exchange_roles(i, r) for r in range(world)]
        mixed_local, mixed_rank = h.step_support(i)
        packed = []
        for r in range(world):
            ro = roles[r]
            assert bool(ro["empty_before"]) == empty[r]  # a shard-id bit cannot be mixed while it is one: empty stays empty between exchanges
            if ro["empty_before"]:
                packed.append(None)
                continue
            pk = _pack_np(mem[r], m, list(Lsel)).copy()
            for b in range(1 << k):
                if ro["unread"] >> b & 1:
                    pk[b * blk:(b + 1) * blk] = np.nan  # never written
            packed.append(pk)
        new_mem = [np.full(1 << m, np.nan + 0j) for _ in range(world)]
        for r in range(world):
            ro = roles[r]
            mine, members = peers_of(r, J)
            assert mine == ro["mine"]
            for b in range(1 << k):
                if ro["recv"] >> b & 1:
                    assert roles[members[b]]["send"] >> mine & 1, "a receive without the matching send would hang RCCL"
                    new_mem[r][b * blk:(b + 1) * blk] = packed[members[b]][mine * blk:(mine + 1) * blk]
                elif b != mine:
                    saved_blocks += 1
                if ro["send"] >> b & 1:
                    assert roles[members[b]]["recv"] >> mine & 1, "a send without the matching receive would hang RCCL"
            if ro["keep_own"]:
                new_mem[r][mine * blk:(mine + 1) * blk] = packed[r][mine * blk:(mine + 1) * blk]
            empty[r] = bool(ro["empty_after"])
            support[r] = ro["new_support"]
        mem = new_mem
    phys = np.concatenate([np.zeros(1 << m, dtype=np.complex128) if empty[r] else np.where((idx & ~support[r]) == 0, mem[r], 0) for r in range(world)])
    assert not np.isnan(phys).any()
    got = logical_from_physical(phys, plans[0].final_pos)
    assert np.max(np.abs(got - want)) < TOL
    if plans[0].exchanges:
        assert saved_blocks > 0  # the first exchange of a run always has empty shards (the initial global qubits are still |0>)


def test_exchange_model_prices_chunked_overlap():
    """bench.py exchange_model (host only): the plain prediction is exchanges + local passes; the overlapped one runs up to three
    passes on either side of an exchange chunk by chunk beside the transfer (priced from the shards' own schedules through
    qsim_plan_passes — a pass can only be chunked over index bits outside its tile).  It can never beat the longer of the two
    legs, never lose against the serial sum, and every pass it counts must exist."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_for_test", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    model = bench.exchange_model(300, "all", 4400.0, 66.0)
    assert [(c["qubits"], c["ranks"]) for c in model["configs"]] == [(30, 2), (30, 4), (30, 8), (33, 8)]
    for c in model["configs"]:
        plain, piped = c["predicted_step_ms"], c["predicted_step_ms_overlapped"]
        assert piped <= plain + 1e-9
        assert piped >= max(c["predicted_exchange_ms"], c["predicted_local_ms"]) - 1e-9
        assert len(c["overlap_per_exchange"]) == c["exchanges"]
        for e in c["overlap_per_exchange"]:
            assert 0 <= e["tail_passes"] <= 3 and 0 <= e["head_passes"] <= 3 and e["saved_ms"] >= 0
            if e["saved_ms"] > 0:
                assert 1 <= e["chunk_bits"] <= 3
