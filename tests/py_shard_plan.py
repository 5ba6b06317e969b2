"""Pure-Python restatement of the shard planner (csrc/dist.cpp build_plan), kept in tests/ as an independent cross-check:
the product path uses the C++ planner through distributed.ShardPlan; tests require the two to agree step by step."""
from typing import List, Tuple

import numpy as np

INF = 1 << 60
_X = np.array([[0, 1], [1, 0]], dtype=np.complex128)


def _is_diag(U: np.ndarray) -> bool:
    return U[0, 1] == 0 and U[1, 0] == 0


class PyShardPlan:
    """Steps for one rank: ('local', [ops]) and ('exchange', rank_bits, local_positions).

    ops: ('u1', local_pos, U) | ('cx', cpos, tpos) | ('scale', z).  Every rank sees the same sequence of
    step kinds and the same exchanges."""

    LINK_UNITS, PACK_UNITS = 25600, 256  # csrc/dist.cpp plan_cost: shard/link and pack pass in common integer units

    def __init__(self, n: int, p: int, gates: List[Tuple], rank: int, lookahead_free_start: bool = True):
        assert 0 <= p <= n - 2 or p == 0, "need at least two local qubits"
        self.n, self.p, self.m, self.rank = n, p, n - p, rank
        # two placement policies are planned in full; the cheaper plan under the exchange cost model is kept
        keep = self._build(gates, lookahead_free_start, full_swap=False)
        best = keep
        if p > 1:
            full = self._build(gates, lookahead_free_start, full_swap=True)
            if self._cost(full[0]) < self._cost(keep[0]):
                best = full
        self.steps, self.exchanges, self.exchanged_fraction, self.final_pos = best

    def _cost(self, steps) -> int:
        return sum(self.PACK_UNITS + (self.LINK_UNITS >> len(s[1])) for s in steps if s[0] == "exchange")

    def _build(self, gates, lookahead_free_start, full_swap):
        steps: List[Tuple] = []
        exchanges = 0
        fraction = 0.0  # sum over exchanges of the shard fraction sent
        pos = list(range(self.n))  # logical -> physical
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
                steps.append(("local", self._emit(run, pos)))
            if deferred:
                new_glob = self._choose_globals(deferred, pos, local_only=full_swap)
                J, Lsel = self._exchange(pos, new_glob)
                if not J:
                    raise RuntimeError("planner made no progress")
                steps.append(("exchange", tuple(J), tuple(Lsel)))
                exchanges += 1
                fraction += 1.0 - 2.0 ** (-len(J))
            remaining = deferred
        return steps, exchanges, fraction, pos

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
    def _choose_globals(self, gates, pos, local_only: bool = False) -> List[int]:
        """The p logical qubits whose next use that needs locality is furthest away (local_only: among the qubits that
        are local now, so that the exchange swaps all p global qubits)."""
        if self.m < self.p:
            local_only = False
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
        cand = [q for q in range(self.n) if not local_only or pos[q] < self.m]
        order = sorted(cand, key=lambda q: (nxt[q], pos[q] >= self.m, pos[q]), reverse=True)
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


