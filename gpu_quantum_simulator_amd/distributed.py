"""Sharded state vector: one process per GPU, 2^n amplitudes split contiguously over P = 2^p ranks.

New design — the reference is single-device.  Physical index bits 0..m-1 (m = n - p) are
local to a shard, bits m..n-1 are the rank id.  A logical->physical qubit map is kept on the host, so a
gate never forces data to move by itself:

  * gates whose qubits are all local run through the single-GPU engine (libqsim.so) on the shard;
  * a diagonal gate on a global qubit is a per-rank scalar; a CX with global control and local target is
    an X on the ranks whose control bit is 1 — no communication;
  * anything else on a global qubit waits.  When nothing more can run, ONE exchange swaps k global
    qubits with k local ones: `qsim_pack_bits` lays the shard out as 2^k contiguous blocks (one local
    HBM pass) and every rank sends block b to group member b and receives that member's block — for
    k = 1 exactly the pairwise half-shard send/recv, for k = p an all-to-all that drives all P-1 xGMI
    links at once (torch.distributed batch_isend_irecv = ncclSend/ncclRecv inside one group on RCCL).
    Which qubits become global is chosen by furthest next non-diagonal use (Belady) — the reference's
    relabelling idea (quantum_simulator_4x4_permute.cu:377-434) with the objective inverted: hot -> local.
  * the initial |0...0> is symmetric under qubit permutations, so the first placement is free.

The planner is the C++ one inside libqsim (csrc/dist.cpp, shared with the C host's qsim_cluster) and is deterministic:
every rank builds the same plan (only its own per-rank scalars / X gates differ), so ranks never need to agree on
anything at run time.  `VirtualCluster` drives P shards inside
one process (exchange = plain copies) so the whole path is testable on one GPU or, with a CPU shard
backend supplied by the tests, on no GPU at all.
"""
from __future__ import annotations

import math
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np

INF = 1 << 60


def _is_diag(U: np.ndarray) -> bool:
    return U[0, 1] == 0 and U[1, 0] == 0


def normalize_gates(gates: Sequence[Sequence], gate_matrix: Callable[[str], np.ndarray]) -> List[Tuple]:
    """('h', q) / ('rz', theta, q) / ('cx', c, t) / ('u1', q, U)  ->  ('u1', q, U) | ('cx', c, t)."""
    cache: Dict[str, np.ndarray] = {}
    out = []
    for g in gates:
        if g[0] == "cx":
            out.append(("cx", int(g[1]), int(g[2])))
        elif g[0] == "u1":
            out.append(("u1", int(g[1]), np.asarray(g[2], dtype=np.complex128).reshape(2, 2)))
        else:
            tok = f"rz({g[1]!r})" if g[0] == "rz" else g[0]
            if tok not in cache:
                m = gate_matrix(tok)
                if m is None:
                    raise ValueError(f"unknown gate {g[0]!r}")
                cache[tok] = m
            out.append(("u1", int(g[-1]), cache[tok]))
    return out


class ShardPlan:
    """Steps for one rank, produced by the C++ planner in libqsim (csrc/dist.cpp, the same one the C host's
    qsim_cluster uses): ('local', [ops]) and ('exchange', rank_bits, local_positions).

    ops: ('u1', local_pos, U) | ('cx', cpos, tpos) | ('scale', z).  Every rank sees the same sequence of step kinds and
    the same exchanges; only the per-rank scalars / conditional X gates differ."""

    def __init__(self, n: int, p: int, gates: List[Tuple], rank: int):
        from .simulator import Circuit, ShardPlanHandle
        self.n, self.p, self.m, self.rank = n, p, n - p, rank
        circ = Circuit.empty(n)
        for g in gates:
            if g[0] == "cx":
                circ.append_cx(g[1], g[2])
            else:
                circ.append_1q(g[2], g[1])
        self.handle = ShardPlanHandle(circ, 1 << p)
        self.steps: List[Tuple] = []
        self.exchanges = 0
        self.exchanged_fraction = 0.0
        for i in range(self.handle.num_steps):
            st = self.handle.step(i)
            if st[0] == "local":
                self.steps.append(("local", self.handle.local_ops(i, rank)))
            else:
                self.steps.append(st)
                self.exchanges += 1
                self.exchanged_fraction += 1.0 - 2.0 ** (-len(st[1]))
        self.final_pos = self.handle.final_pos()

    def predict(self, link_gbps: float = 50.0, pack_gbps: float = 5000.0):
        """(bytes each rank sends, seconds in exchanges) under the planner's cost model: per exchange one pack pass plus
        2^-k of the shard over each of 2^k - 1 links at once."""
        return self.handle.predict(link_gbps, pack_gbps)

    def tune(self, sim, max_candidates: int = 32, budget_ms: float = 6000.0) -> dict:
        """Geometry planning for this rank's local steps (qsim_shard_plan_tune); leaves `sim` reset."""
        return self.handle.tune(self.rank, sim, max_candidates, budget_ms)

    def apply_local(self, step: int, sim) -> None:
        """Queues this rank's ops of a local step on a Simulator, natively."""
        self.handle.apply_local(step, self.rank, sim)

    def local_sweeps(self, rank: Optional[int] = None) -> dict:
        """Host only: what rank `rank`'s engine will schedule for its local steps — passes, and the bytes they move in units
        of one full sweep (read + write) of the shard, the first passes of a sparse phase counted by the fraction of the
        shard they visit — and what it sends: per exchange the blocks that really travel (the first exchange of a run has
        shards that hold nothing).  The ingredients of bench.py's exchange_model."""
        from .simulator import Circuit
        rank = self.rank if rank is None else rank
        m, h = self.m, self.handle
        support, empty = 0, rank != 0
        out = {"passes": 0, "sweeps": 0.0, "exchanges": [], "steps": []}
        for i in range(h.num_steps):
            st = h.step(i)
            if st[0] == "exchange":
                ro = h.exchange_roles(i, rank)
                k = len(st[1])
                out["exchanges"].append({"qubits": k, "blocks_sent": bin(ro["send"]).count("1"), "blocks_received": bin(ro["recv"]).count("1"),
                                         "block_bytes": (16 << m) >> k, "local_positions_out": list(st[2]), "step": i})
                empty, support = bool(ro["empty_after"]), ro["new_support"]
                out["steps"].append("exchange")
                continue
            if empty:
                out["steps"].append("holds nothing")
                continue
            c = Circuit.empty(m)
            for op in h.local_ops(i, rank):
                if op[0] == "cx":
                    c.append_cx(op[1], op[2])
                elif op[0] == "u1":
                    c.append_1q(op[2], op[1])
                else:
                    c.append_1q([[op[1], 0], [0, op[1]]], 0)
            pl = c.plan(initial_support=support)
            sweeps = pl["algorithmic_bytes"] / (32.0 * (1 << m))
            out["passes"] += pl["launches"]
            out["sweeps"] += sweeps
            # pass by pass: the tile bits and the sweeps each pass moves — what decides which passes could run chunk by chunk
            # beside the exchange in front of or behind them (bench.py exchange_model)
            per_pass = [{"tile_mask": x["tile_mask"], "sweeps": x["bytes"] / (32.0 * (1 << m)), "blocks": x["blocks"]}
                        for x in c.passes(initial_support=support)]
            out["steps"].append({"passes": pl["launches"], "sweeps": sweeps, "per_pass": per_pass})
            support = (1 << m) - 1
        return out


_X = np.array([[0, 1], [1, 0]], dtype=np.complex128)


def _bind(shard, plan: "ShardPlan") -> None:
    """HipShard takes the plan itself; test backends (CPU shards) take the per-step op lists."""
    if hasattr(shard, "bind_plan"):
        shard.bind_plan(plan)
    else:
        for i, st in enumerate(plan.steps):
            if st[0] == "local":
                shard.compile(i, st[1])


def peers_of(rank: int, J: Sequence[int]) -> Tuple[int, List[int]]:
    """(my member index, member -> rank) for the exchange group over rank bits J."""
    k = len(J)
    mine = sum(((rank >> J[i]) & 1) << i for i in range(k))
    base = rank
    for j in J:
        base &= ~(1 << j)
    members = [base | sum(((b >> i) & 1) << J[i] for i in range(k)) for b in range(1 << k)]
    return mine, members


# ------------------------------------------------------------------------------------------------------
class HipShard:
    """One shard on one GPU: torch owns the two device buffers (state, exchange scratch), libqsim.so does
    all the arithmetic on the state buffer through qsim_create_external."""

    def __init__(self, m: int, device: int, fuse: int = 3, profile: bool = False, **opts):
        import torch
        from .simulator import Circuit, Simulator
        self.torch, self.m = torch, m
        self.dev = torch.device("cuda", device)
        self.state = torch.empty((1 << m, 2), dtype=torch.float64, device=self.dev)
        self.scratch = torch.empty((1 << m, 2), dtype=torch.float64, device=self.dev)
        self.sim = Simulator(m, device, fuse=fuse, profile=profile, external_ptr=self.state.data_ptr(), **opts)
        # the exchange scratch is idle during local steps: lent to the engine as the second buffer of its out-of-place
        # tile passes (QSIM_OPT_PINGPONG); after every flush the state is back in `state` and the scratch is ours again
        self.sim.set_spare_buffer(self.scratch.data_ptr())
        self._plan = None
        self.device_index = device
        self.comm = None  # RankComm: the exchanges on RCCL, issued by libqsim on the engine's own stream

    def attach_comm(self, world: int, rank: int, uid: bytes):
        from .simulator import RankComm
        self.comm = RankComm(self.sim, self.device_index, world, rank, uid, scratch_ptr=self.scratch.data_ptr())

    def bind_plan(self, plan: "ShardPlan"):
        """Local steps are queued straight from the C++ plan (qsim_shard_plan_apply_local): no Python per gate."""
        self._plan = plan

    # torch and libqsim share ONE HIP runtime in this process (both need libamdhip64.so.7 and the loader maps it once:
    # torch's bundled copy when torch is imported first, which _lib.load() makes sure of), so libqsim's kernels, copies
    # and RCCL calls work on torch-allocated buffers like on its own; only the STREAMS differ (the engine's stream vs
    # torch's current stream), which is why a hand-off between the two is a sync.
    def reset(self, holds_index0: bool):
        self.sim.reset(holds_index0)

    def apply_local(self, key: int, flush: bool = True):
        """flush=False: leave the step's gates queued — the exchange that follows launches them itself, so that the last
        tile pass can write the state straight into the packed layout (qsim_flush_pack)."""
        self._plan.apply_local(key, self.sim)
        if flush:
            self.sim.flush()

    def pack(self, Lsel):
        self.sim.pack_bits(list(Lsel), self.scratch.data_ptr())

    def sync(self):
        self.sim.sync()

    def blocks(self, k):
        return self.state.view(1 << k, -1, 2), self.scratch.view(1 << k, -1, 2)

    def norm2(self) -> float:
        return self.sim.norm2()

    def read_all(self) -> np.ndarray:
        return self.read(0, 1 << self.m)

    def read(self, first, count) -> np.ndarray:
        self.sim.sync()
        return self.state[first:first + count].cpu().numpy().reshape(-1).view(np.complex128)

    def stats(self):
        return self.sim.stats()

    def reset_stats(self):
        self.sim.reset_stats()

    def close(self):
        if self.comm is not None:
            self.comm.close()
            self.comm = None
        self.sim.close()


class ShardedSimulator:
    """One rank of a torch.distributed job (backend nccl = RCCL over xGMI; gloo for CPU tests)."""

    def __init__(self, n: int, gates: Sequence[Sequence], device: int = 0, fuse: int = 3, profile: bool = False,
                 shard_factory: Optional[Callable] = None, **opts):
        import torch.distributed as dist
        from .simulator import gate_matrix
        self.dist = dist
        self.world, self.rank = dist.get_world_size(), dist.get_rank()
        p = int(round(math.log2(self.world)))
        if 1 << p != self.world:
            raise ValueError("world size must be a power of two")
        self.n, self.p, self.m = n, p, n - p
        self.plan = ShardPlan(n, p, normalize_gates(gates, gate_matrix), self.rank)
        self.shard = (shard_factory or HipShard)(self.m, device, fuse=fuse, profile=profile, **opts)
        _bind(self.shard, self.plan)
        self._xsec = 0.0
        self._xbytes = 0
        sent = [x["blocks_sent"] * x["block_bytes"] for x in self.plan.local_sweeps()["exchanges"]]
        per_link = [x["block_bytes"] for x in self.plan.local_sweeps()["exchanges"] if x["blocks_sent"] or x["blocks_received"]]
        self.exchange_prediction = {"model": "per exchange: 2^-k of the shard per link, 2^k - 1 links at once, 50 GB/s per link and direction; blocks "
                                             "of ranks that hold nothing yet stay home; the re-layout rides on the last tile pass before the exchange",
                                    "bytes_sent_by_this_rank_per_step": float(sum(sent)), "seconds_per_step": sum(b / 50e9 for b in per_link)}
        self.exchange_backend = "torch.distributed"
        import os
        # QSIM_EXCHANGE=torch keeps the round-1 form (pack, then torch.distributed send/recv on torch's stream): an
        # operator's switch for a node where the library's own communicator misbehaves; set it on every rank
        if (self.world > 1 and hasattr(self.shard, "attach_comm") and dist.get_backend() == "nccl"
                and os.environ.get("QSIM_EXCHANGE", "") != "torch"):
            self._attach_native_comm()
        if self.exchange_backend == "torch.distributed":
            self._warm_up_links()

    def _attach_native_comm(self):
        """Every byte of state then travels through ncclSend / ncclRecv issued by libqsim itself (csrc/dist.cpp
        qsim_rank_comm_exchange); torch.distributed only carries the 128-byte RCCL id.  All ranks must agree on the
        outcome, so a failure anywhere (reduced with MIN) sends every rank to the torch.distributed path."""
        import sys
        import torch
        from .simulator import RankComm
        dist, dev = self.dist, self.shard.state.device
        t = torch.zeros(128, dtype=torch.uint8, device=dev)
        ok = 1
        try:
            if self.rank == 0:
                t.copy_(torch.frombuffer(bytearray(RankComm.unique_id()), dtype=torch.uint8))
        except Exception as e:  # noqa: BLE001 - reported, then agreed on collectively
            sys.stderr.write(f"[rank {self.rank}] RCCL id: {e}\n")
            ok = 0
        dist.broadcast(t, 0)
        if ok:
            try:
                self.shard.attach_comm(self.world, self.rank, bytes(t.cpu().numpy().tobytes()))
            except Exception as e:  # noqa: BLE001
                sys.stderr.write(f"[rank {self.rank}] native RCCL communicator failed, using torch.distributed: {e}\n")
                ok = 0
        flag = torch.tensor([ok], dtype=torch.int32, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 1:
            self.exchange_backend = "rccl-native"
        elif self.shard.comm is not None:
            self.shard.comm.close()
            self.shard.comm = None

    @property
    def exchange_seconds(self) -> float:
        if self.exchange_backend == "rccl-native":
            return self.shard.comm.stats()[2]  # HIP-event time on the engine's stream
        return self._xsec

    @property
    def exchange_bytes(self) -> float:
        if self.exchange_backend == "rccl-native":
            return self.shard.comm.stats()[1]
        return self._xbytes

    def _warm_up_links(self):
        """One tiny send/recv with every peer this rank will ever exchange with, so communicator set-up (RCCL builds
        its point-to-point channels lazily) never lands inside a timed step."""
        if self.world == 1:
            return
        import torch
        dist, shard = self.dist, self.shard
        dev = shard.state.device
        peers = sorted({peer for st in self.plan.steps if st[0] == "exchange"
                        for peer in peers_of(self.rank, st[1])[1] if peer != self.rank})
        if peers:
            tx = torch.zeros(8, dtype=torch.float64, device=dev)
            rx = [torch.zeros(8, dtype=torch.float64, device=dev) for _ in peers]
            ops = []
            for i, peer in enumerate(peers):
                ops.append(dist.P2POp(dist.isend, tx, peer))
                ops.append(dist.P2POp(dist.irecv, rx[i], peer))
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        if any(st[0] == "exchange" and len(st[1]) == self.p for st in self.plan.steps):
            a = torch.zeros(self.world * 2, dtype=torch.float64, device=dev)
            dist.all_to_all_single(torch.empty_like(a), a)
        if dev.type == "cuda":
            torch.cuda.synchronize()
        dist.barrier()

    def tune(self, max_candidates: int = 32, budget_ms: float = 6000.0):
        """Plans the pass geometries of this rank's local steps (HIP shards only; a no-op for test backends)."""
        if hasattr(self.shard, "sim"):
            return self.plan.tune(self.shard.sim, max_candidates, budget_ms)
        return None

    def run_step(self):
        import time
        self.shard.reset(self.rank == 0)
        for i, st in enumerate(self.plan.steps):
            if st[0] == "local":
                before_exchange = i + 1 < len(self.plan.steps) and self.plan.steps[i + 1][0] == "exchange"
                if before_exchange and self.exchange_backend == "rccl-native":
                    self.shard.apply_local(i, flush=False)
                else:
                    self.shard.apply_local(i)
            elif self.exchange_backend == "rccl-native":
                self.shard.comm.exchange_step(self.plan.handle, i)  # pack + one ncclGroup on the engine's stream; nothing waits here
            else:
                t0 = time.perf_counter()
                self._exchange(st[1], st[2])
                self._xsec += time.perf_counter() - t0

    def _exchange(self, J, Lsel):
        dist, shard = self.dist, self.shard
        k = len(J)
        shard.pack(Lsel)
        shard.sync()  # the pack ran on the engine's stream; the collective uses torch's
        mine, members = peers_of(self.rank, J)
        st, sc = shard.blocks(k)
        if k == self.p and tuple(J) == tuple(range(self.p)):
            # every rank takes part and member b IS rank b: one all-to-all collective (block b of the scratch goes to
            # rank b, block b of the state comes from rank b) — RCCL drives all P-1 xGMI links at once
            dist.all_to_all_single(st.view(-1), sc.view(-1))
            self._xbytes += (len(members) - 1) * sc[0].numel() * 8
        else:
            ops = []
            for b, peer in enumerate(members):
                if b == mine:
                    st[b].copy_(sc[b])
                else:
                    ops.append(dist.P2POp(dist.isend, sc[b], peer))
                    ops.append(dist.P2POp(dist.irecv, st[b], peer))
                    self._xbytes += sc[b].numel() * 8
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        if st.is_cuda:
            shard.torch.cuda.current_stream().synchronize()

    # -- bench.py interface
    def sync(self):
        self.shard.sync()

    def stats(self):
        return self.shard.stats()

    def reset_stats(self):
        self.shard.reset_stats()
        self._xsec = 0.0
        self._xbytes = 0
        if self.exchange_backend == "rccl-native":
            self.shard.comm.stats(reset=True)

    def norm2(self) -> float:
        import torch
        t = torch.tensor([self.shard.norm2()], dtype=torch.float64)
        if self.dist.get_backend() == "nccl":
            t = t.cuda()
        self.dist.all_reduce(t)
        return float(t.item())

    def gather_logical(self) -> np.ndarray:
        """Collective: every rank contributes its shard and gets the whole state in LOGICAL index order (small registers:
        the self-check of bench.py --gpus N, tests)."""
        import torch
        mine = torch.from_numpy(np.ascontiguousarray(self.shard.read_all()).view(np.float64).copy())
        if self.dist.get_backend() == "nccl":
            mine = mine.cuda()
        parts = [torch.empty_like(mine) for _ in range(self.world)]
        self.dist.all_gather(parts, mine)
        phys = torch.cat(parts).cpu().numpy().view(np.complex128)
        return logical_from_physical(phys, self.plan.final_pos)

    def amplitude(self, logical_index: int) -> complex:
        """One amplitude by LOGICAL basis index (collective)."""
        import torch
        phys = physical_index(logical_index, self.plan.final_pos)
        owner, local = phys >> self.m, phys & ((1 << self.m) - 1)
        t = torch.zeros(2, dtype=torch.float64)
        if owner == self.rank:
            a = self.shard.read(local, 1)[0]
            t[0], t[1] = a.real, a.imag
        if self.dist.get_backend() == "nccl":
            t = t.cuda()
        self.dist.all_reduce(t)
        return complex(float(t[0]), float(t[1]))

    def sample(self, randoms, block_bits: int = 12, chunk_bits: int = 22) -> np.ndarray:
        """measurement() (quantum_simulator.c:270-283) on the sharded state, indices in LOGICAL order (collective: every
        rank calls it with the same numbers and gets the same indices).  Each rank sums |a|^2 of its shard into the
        logical 2^block_bits-amplitude blocks it holds a part of; ONE all-reduce adds the P partial sums per block
        (SURVEY 8f row 1); a draw then needs the amplitudes of its own block only, contributed by their owners."""
        import torch
        dist, n, m = self.dist, self.n, self.m
        pos = list(self.plan.final_pos)
        inv = [0] * n  # physical bit -> logical qubit
        for q, pq in enumerate(pos):
            inv[pq] = q
        bb = min(block_bits, n)
        nblocks = 1 << (n - bb)
        part = np.zeros(nblocks, dtype=np.float64)
        piece = 1 << min(chunk_bits, m)
        j = np.arange(piece, dtype=np.int64)
        low = np.zeros(piece, dtype=np.int64)  # logical image of the physical bits inside a piece
        for b in range(min(chunk_bits, m)):
            low |= ((j >> b) & 1) << inv[b]
        for at in range(0, 1 << m, piece):
            hi = 0
            phys_hi = (self.rank << m) | at
            for b in range(min(chunk_bits, m), n):
                hi |= ((phys_hi >> b) & 1) << inv[b]
            a = self.shard.read(at, piece)
            w = a.real * a.real + a.imag * a.imag
            part += np.bincount((low | hi) >> bb, weights=w, minlength=nblocks)
        cuda = dist.get_backend() == "nccl"
        t = torch.from_numpy(part)
        if cuda:
            t = t.cuda()
        dist.all_reduce(t)
        prefix = np.cumsum(t.cpu().numpy())  # cumulative at the END of each block
        r_all = np.ascontiguousarray(randoms, dtype=np.float64).reshape(-1)
        out = np.zeros(r_all.size, dtype=np.uint64)
        bsize = 1 << bb
        cached, blk = -1, None
        for k, rnd in enumerate(r_all):
            lo, hi_ = 0, nblocks
            while lo < hi_:  # first block whose end value is non-zero and >= r
                mid = (lo + hi_) >> 1
                if prefix[mid] == 0.0 or prefix[mid] < rnd:
                    lo = mid + 1
                else:
                    hi_ = mid
            idx, found, b = (1 << n) - 1, False, lo
            while b < nblocks and not found:
                if b != cached:
                    logical = (b << bb) + np.arange(bsize, dtype=np.int64)
                    phys = np.zeros(bsize, dtype=np.int64)
                    for q, pq in enumerate(pos):
                        phys |= ((logical >> q) & 1) << pq
                    mine = np.nonzero((phys >> m) == self.rank)[0]
                    buf = np.zeros(2 * bsize, dtype=np.float64)
                    for i in mine:  # owners fill their slots; the sum over ranks is the block
                        amp = self.shard.read(int(phys[i]) & ((1 << m) - 1), 1)[0]
                        buf[2 * i], buf[2 * i + 1] = amp.real, amp.imag
                    tb = torch.from_numpy(buf)
                    if cuda:
                        tb = tb.cuda()
                    dist.all_reduce(tb)
                    blk = tb.cpu().numpy()
                    cached = b
                cum = prefix[b - 1] if b else 0.0
                probs = blk[0::2] * blk[0::2] + blk[1::2] * blk[1::2]
                for i in range(bsize):
                    cum += probs[i]
                    if not (cum == 0.0 or cum < rnd):
                        idx, found = (b << bb) + i, True
                        break
                b += 1
            out[k] = idx
        return out

    def close(self):
        self.shard.close()


def physical_index(logical: int, pos: Sequence[int]) -> int:
    out = 0
    for q, pq in enumerate(pos):
        out |= ((logical >> q) & 1) << pq
    return out


def logical_from_physical(phys: np.ndarray, pos: Sequence[int]) -> np.ndarray:
    """Reorders a full physical-order vector into logical order (tests, small n)."""
    n = len(pos)
    idx = np.arange(1 << n, dtype=np.int64)
    src = np.zeros_like(idx)
    for q, pq in enumerate(pos):
        src |= ((idx >> q) & 1) << pq
    return phys[src]


class VirtualCluster:
    """P shards in ONE process (all on one device, or CPU shards from the tests): same plans, same pack
    kernel, the exchange done with plain copies.  Used to validate the sharded path bit for bit against
    the unsharded one where P devices are not available."""

    def __init__(self, n: int, world: int, gates: Sequence[Sequence], shard_factory: Optional[Callable] = None,
                 device: int = 0, **opts):
        from .simulator import gate_matrix
        p = int(round(math.log2(world)))
        assert 1 << p == world
        self.n, self.p, self.m, self.world = n, p, n - p, world
        norm = normalize_gates(gates, gate_matrix)
        self.plans = [ShardPlan(n, p, norm, r) for r in range(world)]
        factory = shard_factory or HipShard
        self.shards = [factory(self.m, device, **opts) for _ in range(world)]
        for r in range(world):
            _bind(self.shards[r], self.plans[r])

    def run(self):
        for r, sh in enumerate(self.shards):
            sh.reset(r == 0)
        nsteps = len(self.plans[0].steps)
        assert all(len(pl.steps) == nsteps for pl in self.plans)
        for i in range(nsteps):
            kind = self.plans[0].steps[i][0]
            assert all(pl.steps[i][0] == kind for pl in self.plans)
            if kind == "local":
                for sh in self.shards:
                    sh.apply_local(i)
            else:
                _, J, Lsel = self.plans[0].steps[i]
                assert all(pl.steps[i] == self.plans[0].steps[i] for pl in self.plans)
                for sh in self.shards:
                    sh.pack(Lsel)
                for sh in self.shards:
                    sh.sync()
                k = len(J)
                for r, sh in enumerate(self.shards):
                    mine, members = peers_of(r, J)
                    st, _ = sh.blocks(k)
                    for b, peer in enumerate(members):
                        _, peer_sc = self.shards[peer].blocks(k)
                        st[b].copy_(peer_sc[mine])
                if self.shards[0].state.is_cuda:
                    self.shards[0].torch.cuda.synchronize()

    def gather_logical(self) -> np.ndarray:
        phys = np.concatenate([sh.read_all() for sh in self.shards])
        return logical_from_physical(phys, self.plans[0].final_pos)

    def close(self):
        for sh in self.shards:
            sh.close()
