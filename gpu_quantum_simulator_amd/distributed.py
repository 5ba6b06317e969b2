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

The planner is plain Python and identical on every rank (only the emitted per-rank scalars / X gates
differ), so ranks never need to agree on anything at run time.  `VirtualCluster` drives P shards inside
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
    """Steps for one rank: ('local', [ops]) and ('exchange', rank_bits, local_positions).

    ops: ('u1', local_pos, U) | ('cx', cpos, tpos) | ('scale', z).  Every rank sees the same sequence of
    step kinds and the same exchanges."""

    def __init__(self, n: int, p: int, gates: List[Tuple], rank: int, lookahead_free_start: bool = True):
        assert 0 <= p <= n - 2 or p == 0, "need at least two local qubits"
        self.n, self.p, self.m, self.rank = n, p, n - p, rank
        self.steps: List[Tuple] = []
        self.exchanges = 0
        self.exchanged_fraction = 0.0  # sum over exchanges of the shard fraction sent
        pos = list(range(n))  # logical -> physical
        remaining = list(gates)
        first = True
        while remaining:
            if self.p and first and lookahead_free_start:
                # |0...0> is permutation-symmetric: choose the first global set for free
                new_glob = self._choose_globals(remaining, pos)
                self._relabel_free(pos, new_glob)
            first = False
            run, deferred = self._split(remaining, pos)
            if run:
                self.steps.append(("local", self._emit(run, pos)))
            if deferred:
                new_glob = self._choose_globals(deferred, pos)
                J, Lsel = self._exchange(pos, new_glob)
                if not J:
                    raise RuntimeError("planner made no progress")
                self.steps.append(("exchange", tuple(J), tuple(Lsel)))
                self.exchanges += 1
                self.exchanged_fraction += 1.0 - 2.0 ** (-len(J))
            remaining = deferred
        self.final_pos = pos

    # -- which gates can run under the current placement
    def _needs_local(self, g) -> Tuple[int, ...]:
        """Logical qubits this gate needs in local positions."""
        if g[0] == "cx":
            return (g[2],) if g[1] != g[2] else ()
        return () if _is_diag(g[2]) else (g[1],)

    def _split(self, gates, pos):
        m = self.m
        run, deferred, blocked = [], [], set()
        for g in gates:
            qs = {g[1], g[2]} if g[0] == "cx" else {g[1]}
            if qs & blocked:
                blocked |= qs
                deferred.append(g)
                continue
            if all(pos[q] < m for q in self._needs_local(g)):
                run.append(g)
            else:
                blocked |= qs
                deferred.append(g)
        return run, deferred

    def _emit(self, run, pos):
        m, ops = self.m, []
        for g in run:
            if g[0] == "cx":
                c, t = g[1], g[2]
                if c == t:
                    continue
                if pos[c] < m:
                    ops.append(("cx", pos[c], pos[t]))
                elif (self.rank >> (pos[c] - m)) & 1:
                    ops.append(("u1", pos[t], _X))
            else:
                q, U = g[1], g[2]
                if pos[q] < m:
                    ops.append(("u1", pos[q], U))
                else:
                    b = (self.rank >> (pos[q] - m)) & 1
                    z = complex(U[b, b])
                    if z != 1.0:
                        ops.append(("scale", z))
        return ops

    # -- placement
    def _choose_globals(self, gates, pos) -> List[int]:
        """The p logical qubits whose next use that needs locality is furthest away."""
        nxt = [INF] * self.n
        found = 0
        for i, g in enumerate(gates):
            for q in self._needs_local(g):
                if nxt[q] == INF:
                    nxt[q] = i
                    found += 1
            if found == self.n:
                break
        # prefer: far next use; then already-global (nothing to move); then a high position (long pack runs)
        order = sorted(range(self.n), key=lambda q: (nxt[q], pos[q] >= self.m, pos[q]), reverse=True)
        return order[: self.p]

    def _relabel_free(self, pos, new_glob):
        """Initial placement: permute the map without moving data."""
        m = self.m
        cur_glob = [q for q in range(self.n) if pos[q] >= m]
        outgoing = [q for q in new_glob if pos[q] < m]
        incoming = [q for q in cur_glob if q not in new_glob]
        for a, b in zip(outgoing, incoming):
            pos[a], pos[b] = pos[b], pos[a]

    def _exchange(self, pos, new_glob):
        """Updates pos for swapping the outgoing locals with the incoming globals; returns (rank bit ids,
        local positions), both ascending and paired in that order."""
        m = self.m
        cur_glob = [q for q in range(self.n) if pos[q] >= m]
        outgoing = sorted((q for q in new_glob if pos[q] < m), key=lambda q: pos[q])
        incoming = sorted((q for q in cur_glob if q not in new_glob), key=lambda q: pos[q])
        k = len(outgoing)
        assert k == len(incoming)
        if k == 0:
            return [], []
        Lsel = [pos[q] for q in outgoing]
        J = [pos[q] - m for q in incoming]
        sel = set(Lsel)
        # remaining locals compact downwards in order; incoming globals land in the top k local positions
        newpos = list(pos)
        for q in range(self.n):
            if pos[q] < m and pos[q] not in sel:
                newpos[q] = pos[q] - sum(1 for s in Lsel if s < pos[q])
        for i, q in enumerate(incoming):
            newpos[q] = m - k + i
        for i, q in enumerate(outgoing):
            newpos[q] = m + J[i]
        pos[:] = newpos
        return J, Lsel


_X = np.array([[0, 1], [1, 0]], dtype=np.complex128)


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
        self._Circuit = Circuit
        self._compiled: Dict[int, object] = {}

    def compile(self, key: int, ops):
        c = self._Circuit.empty(self.m)
        for op in ops:
            if op[0] == "cx":
                c.append_cx(op[1], op[2])
            elif op[0] == "u1":
                c.append_1q(op[2], op[1])
            else:
                c.append_1q(np.diag([op[1], op[1]]), 0)  # per-rank scalar, folds into the next fused block
        self._compiled[key] = c

    # The buffers belong to torch's HIP runtime, libqsim runs on the system one (see _lib.load): device
    # addresses are shared process-wide, so kernels work on them directly, but host<->device copies and
    # stream ordering stay with the runtime that owns the allocation.  Hand-offs are host-side syncs.
    def reset(self, holds_index0: bool):
        self.sim.reset(holds_index0)

    def apply_local(self, key: int):
        self.sim.run(self._compiled[key])
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
        for i, st in enumerate(self.plan.steps):
            if st[0] == "local":
                self.shard.compile(i, st[1])
        self.exchange_seconds = 0.0
        self.exchange_bytes = 0
        self._warm_up_links()

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

    def run_step(self):
        import time
        self.shard.reset(self.rank == 0)
        for i, st in enumerate(self.plan.steps):
            if st[0] == "local":
                self.shard.apply_local(i)
            else:
                t0 = time.perf_counter()
                self._exchange(st[1], st[2])
                self.exchange_seconds += time.perf_counter() - t0

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
            self.exchange_bytes += (len(members) - 1) * sc[0].numel() * 8
        else:
            ops = []
            for b, peer in enumerate(members):
                if b == mine:
                    st[b].copy_(sc[b])
                else:
                    ops.append(dist.P2POp(dist.isend, sc[b], peer))
                    ops.append(dist.P2POp(dist.irecv, st[b], peer))
                    self.exchange_bytes += sc[b].numel() * 8
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
        self.exchange_seconds = 0.0
        self.exchange_bytes = 0

    def norm2(self) -> float:
        import torch
        t = torch.tensor([self.shard.norm2()], dtype=torch.float64)
        if self.dist.get_backend() == "nccl":
            t = t.cuda()
        self.dist.all_reduce(t)
        return float(t.item())

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
            for i, st in enumerate(self.plans[r].steps):
                if st[0] == "local":
                    self.shards[r].compile(i, st[1])

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
