"""Seeded synthetic-circuit generators that emit the OPENQASM-3 subset the reference parser accepts.

The reference ships only two circuits (entanglement.qasm, grover_3_18.qasm) and benchmarks on
`random_circs_ad/random_<n>.qasm`, which it does not ship (tester.bash:12).  These generators produce
files of that shape: the header of grover_3_18.qasm:1-3, then one gate statement per line drawn from
the vocabulary of quantum_simulator.c:13-23.

Pure Python, no numpy, so the same bytes come out on every host for a given seed.
"""
from __future__ import annotations

import math
from typing import Iterable, List, Sequence, Tuple

CLIFFORD_T_1Q = ("h", "s", "sdg", "t", "tdg", "x", "z", "sx")
ALL_1Q = CLIFFORD_T_1Q + ("rz",)

Gate = Tuple  # ("h", q) | ("rz", theta, q) | ("cx", control, target)


class XorShift64Star:
    """xorshift64* — small, deterministic, identical everywhere."""

    def __init__(self, seed: int):
        self.s = (seed * 0x9E3779B97F4A7C15 + 0xD1B54A32D192ED03) & 0xFFFFFFFFFFFFFFFF or 1

    def next_u64(self) -> int:
        x = self.s
        x ^= x >> 12
        x ^= (x << 25) & 0xFFFFFFFFFFFFFFFF
        x ^= x >> 27
        self.s = x
        return (x * 0x2545F4914F6CDD1D) & 0xFFFFFFFFFFFFFFFF

    def below(self, n: int) -> int:
        return self.next_u64() % n

    def uniform(self) -> float:
        return (self.next_u64() >> 11) / float(1 << 53)


def random_gates(n: int, depth: int, seed: int, vocabulary: str = "all", cx_prob: float = 0.25) -> List[Gate]:
    """`depth` gate statements on n qubits.

    vocabulary "clifford_t": uniform over {h,s,sdg,t,tdg,x,z,sx} plus cx with probability cx_prob
    (BASELINE config 3).  "all": the same plus rz(theta), theta uniform in (-pi, pi) (config 4).
    Qubits are uniform; cx operands are distinct.
    """
    rng = XorShift64Star(seed)
    names = CLIFFORD_T_1Q if vocabulary == "clifford_t" else ALL_1Q
    gates: List[Gate] = []
    for _ in range(depth):
        if n >= 2 and rng.uniform() < cx_prob:
            c = rng.below(n)
            t = rng.below(n - 1)
            if t >= c:
                t += 1
            gates.append(("cx", c, t))
        else:
            g = names[rng.below(len(names))]
            q = rng.below(n)
            if g == "rz":
                theta = (2.0 * rng.uniform() - 1.0) * math.pi
                gates.append(("rz", theta, q))
            else:
                gates.append((g, q))
    return gates


def qasm_text(n: int, gates: Iterable[Gate], *, crlf: bool = False, physical: bool = False,
              qubit_style: str = "prefix") -> str:
    """Render gates as the text quantum_simulator.c:115-254 parses.

    qubit_style "prefix" -> `qubit[n] q;` (grover_3_18.qasm:3), "suffix" -> `qubit q[n];`
    (entanglement.qasm:3).  physical=True writes operands as `$k` (accepted at :225,:231).
    """
    eol = "\r\n" if crlf else "\n"

    def op(q: int) -> str:
        return f"${q}" if physical else f"q[{q}]"

    lines = ["OPENQASM 3.0;", 'include "stdgates.inc";',
             f"qubit[{n}] q;" if qubit_style == "prefix" else f"qubit q[{n}];"]
    for g in gates:
        if g[0] == "cx":
            lines.append(f"cx {op(g[1])}, {op(g[2])};")
        elif g[0] == "rz":
            lines.append(f"rz({g[1]!r}) {op(g[2])};")
        else:
            lines.append(f"{g[0]} {op(g[1])};")
    return eol.join(lines) + eol


def write_qasm(path: str, n: int, gates: Sequence[Gate], **kw) -> str:
    with open(path, "w", newline="") as f:
        f.write(qasm_text(n, gates, **kw))
    return path


def random_circuit_file(path: str, n: int, depth: int, seed: int, vocabulary: str = "all", **kw) -> str:
    return write_qasm(path, n, random_gates(n, depth, seed, vocabulary), **kw)


def probe_gates(n: int, target: int, repeats: int = 30, name: str = "h") -> List[Gate]:
    """Single-qubit roofline probe: `repeats` consecutive gates on one target."""
    return [(name, target)] * repeats
