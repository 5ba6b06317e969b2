"""CPU stand-in for distributed.HipShard, for tests only: same interface, state in host memory, gates
applied with the ORACLE's loops, pack done with numpy.  Lets the planner, the logical->physical map and the
exchange pattern run under gloo (or in one process) without a GPU."""
import numpy as np
import torch

from oracle import oracle


class CpuShard:
    torch = torch

    def __init__(self, m, device=0, **_opts):
        self.m = m
        self.state = torch.zeros((1 << m, 2), dtype=torch.float64)
        self.scratch = torch.zeros((1 << m, 2), dtype=torch.float64)
        self._ops = {}
        self.gates_applied = 0

    def _np(self):
        return self.state.numpy().reshape(-1).view(np.complex128)

    def compile(self, key, ops):
        self._ops[key] = ops

    def reset(self, holds_index0):
        self.state.zero_()
        if holds_index0:
            self.state[0, 0] = 1.0

    def apply_local(self, key):
        s = self._np()
        for op in self._ops[key]:
            if op[0] == "cx":
                oracle.apply_cx(s, self.m, op[1], op[2])
            elif op[0] == "u1":
                oracle.apply_1q(s, self.m, np.asarray(op[2]).T, op[1])  # the oracle applies the transpose
            else:
                s *= op[1]
            self.gates_applied += 1

    def pack(self, Lsel):
        m, k = self.m, len(Lsel)
        d = np.arange(1 << m, dtype=np.int64)
        rest, blk = d & ((1 << (m - k)) - 1), d >> (m - k)
        keep = [b for b in range(m) if b not in Lsel]
        src = np.zeros_like(d)
        for i, b in enumerate(keep):
            src |= ((rest >> i) & 1) << b
        for i, b in enumerate(Lsel):
            src |= ((blk >> i) & 1) << b
        self.scratch.numpy()[:] = self.state.numpy()[src]

    def sync(self):
        pass

    def blocks(self, k):
        return self.state.view(1 << k, -1, 2), self.scratch.view(1 << k, -1, 2)

    def norm2(self):
        return float(np.vdot(self._np(), self._np()).real)

    def read_all(self):
        return self._np().copy()

    def read(self, first, count):
        return self._np()[first:first + count].copy()

    def stats(self):
        return {"gates": self.gates_applied}

    def reset_stats(self):
        self.gates_applied = 0

    def close(self):
        pass
