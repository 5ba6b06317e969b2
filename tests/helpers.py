"""Small numpy helpers for the tests (standard U·v orientation, q[0] = least-significant index bit)."""
import numpy as np


def np_apply_1q(state, n, U, q):
    s = state.reshape(1 << (n - q - 1), 2, 1 << q)
    out = np.einsum("ab,xby->xay", np.asarray(U, dtype=np.complex128).reshape(2, 2), s)
    return out.reshape(-1)


def np_apply_2q(state, n, U, q_hi, q_lo):
    assert q_hi > q_lo
    s = state.reshape(1 << (n - q_hi - 1), 2, 1 << (q_hi - q_lo - 1), 2, 1 << q_lo)
    M = np.asarray(U, dtype=np.complex128).reshape(2, 2, 2, 2)  # [hi_out, lo_out, hi_in, lo_in]
    out = np.einsum("acbd,xbydz->xaycz", M, s)
    return out.reshape(-1)


def np_apply_cx(state, n, control, target):
    if control == target:
        return state
    idx = np.arange(1 << n)
    src = np.where((idx >> control) & 1, idx ^ (1 << target), idx)
    return state[src]


def np_apply_kq(state, n, U, qubits):
    """k-qubit unitary, `qubits` most significant first (matrix index bits in that order)."""
    k = len(qubits)
    t = state.reshape([2] * n)  # axis a <-> qubit n-1-a
    axes = [n - 1 - q for q in qubits]
    M = np.asarray(U, dtype=np.complex128).reshape([2] * (2 * k))
    out = np.tensordot(M, t, axes=(list(range(k, 2 * k)), axes))  # new axes 0..k-1 = the gate's output indices
    out = np.moveaxis(out, list(range(k)), axes)
    return np.ascontiguousarray(out).reshape(-1)


def replay_schedule(n, sched):
    """Applies Circuit.schedule() output to |0..0> with numpy."""
    s = np.zeros(1 << n, dtype=np.complex128)
    s[0] = 1
    for _pass, _k, kind, qs, m, _folded in sched:
        if kind == "cx":
            s = np_apply_cx(s, n, qs[0], qs[1])
        else:
            s = np_apply_kq(s, n, m, qs)
    return s


def random_unitary(dim, rng):
    a = rng.standard_normal((dim, dim)) + 1j * rng.standard_normal((dim, dim))
    q, r = np.linalg.qr(a)
    return q * (np.diag(r) / np.abs(np.diag(r)))
