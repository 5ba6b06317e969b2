"""Property test of the host scheduler (no GPU): for arbitrary small circuits over the full gate vocabulary plus
generic 2-qubit unitaries, every fusion level and arbitrary tile geometries must replay to the same state as applying
the gates one by one."""
import numpy as np
from hypothesis import HealthCheck, given, settings, strategies as st

from gpu_quantum_simulator_amd import Circuit, gate_matrix
from helpers import np_apply_1q, np_apply_cx, np_apply_kq, random_unitary, replay_schedule

NAMES = ["h", "s", "sdg", "t", "tdg", "x", "z", "sx", "rz(0.25)", "rz(-2.5)"]


@st.composite
def circuits_st(draw):
    n = draw(st.integers(2, 9))
    m = draw(st.integers(1, 60))
    gates = []
    for _ in range(m):
        kind = draw(st.integers(0, 9))
        if kind <= 5:
            gates.append(("u1", draw(st.sampled_from(NAMES)), draw(st.integers(0, n - 1))))
        elif kind <= 8:
            c = draw(st.integers(0, n - 1))
            t = draw(st.integers(0, n - 1))  # c == t allowed: the reference's silent no-op
            gates.append(("cx", c, t))
        else:
            hi = draw(st.integers(1, n - 1))
            lo = draw(st.integers(0, hi - 1))
            gates.append(("u2", draw(st.integers(0, 2 ** 31)), hi, lo))
    return n, gates


@settings(max_examples=150, deadline=None, suppress_health_check=[HealthCheck.too_slow])
@given(circuits_st(), st.integers(0, 3), st.integers(8, 12), st.integers(2, 6), st.integers(1, 8))
def test_schedule_replay_matches_gate_by_gate(case, fuse, tile_bits, tile_low_bits, tile_max_ops):
    n, gates = case
    if tile_bits - tile_low_bits < 2:
        tile_low_bits = tile_bits - 2
    c = Circuit.empty(n)
    ref = np.zeros(1 << n, dtype=np.complex128)
    ref[0] = 1
    for g in gates:
        if g[0] == "u1":
            U = gate_matrix(g[1])
            c.append_1q(U, g[2])
            ref = np_apply_1q(ref, n, U, g[2])
        elif g[0] == "cx":
            c.append_cx(g[1], g[2])
            ref = np_apply_cx(ref, n, g[1], g[2])
        else:
            U = random_unitary(4, np.random.default_rng(g[1]))
            c.append_2q(U, g[2], g[3])
            ref = np_apply_kq(ref, n, U, (g[2], g[3]))
    sched = c.schedule(fuse=fuse, tile_bits=tile_bits, tile_low_bits=tile_low_bits, tile_max_ops=tile_max_ops)
    got = replay_schedule(n, sched)
    assert np.max(np.abs(got - ref)) < 1e-12
    assert sum(s[5] for s in sched) <= len(gates)
    for s in sched:  # blocks of a tile pass stay expressible: at most 3 qubits, at most 4 entries per row beyond 2 qubits
        if s[1] == "tile" and len(s[3]) == 3:
            assert int((s[4] != 0).sum(axis=1).max()) <= 4
