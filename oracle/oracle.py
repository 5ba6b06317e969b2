"""ctypes loader for oracle/liboracle.so (the CPU restatement) and, when present, oracle/_ref.

TEST INFRASTRUCTURE.  Importers allowed: tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg.
The product package gpu_quantum_simulator_amd never imports this module.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_REF = None


def build(with_reference: bool = True) -> None:
    subprocess.check_call(["make", "-s", "-C", HERE, "all"])
    if with_reference and os.path.isdir(os.environ.get("REFERENCE", "/root/reference")):
        subprocess.check_call(["make", "-s", "-C", HERE, "ref"])


def lib() -> ctypes.CDLL:
    global _LIB
    if _LIB is None:
        path = os.path.join(HERE, "liboracle.so")
        if not os.path.exists(path):
            build(with_reference=False)
        L = ctypes.CDLL(path)
        dp = ctypes.POINTER(ctypes.c_double)
        L.oracle_apply_1q.argtypes = [dp, ctypes.c_int, dp, ctypes.c_int]
        L.oracle_apply_cx.argtypes = [dp, ctypes.c_int, ctypes.c_int, ctypes.c_int]
        L.oracle_gate_matrix.argtypes = [ctypes.c_char_p, dp]
        L.oracle_gate_matrix.restype = ctypes.c_int
        L.oracle_run_qasm.argtypes = [ctypes.c_char_p, ctypes.POINTER(ctypes.c_int), dp, ctypes.c_long,
                                      ctypes.POINTER(ctypes.c_long), ctypes.POINTER(ctypes.c_int)]
        L.oracle_run_qasm.restype = ctypes.c_void_p
        L.oracle_free.argtypes = [ctypes.c_void_p]
        L.oracle_cumulative.argtypes = [dp, ctypes.c_int]
        L.oracle_cumulative.restype = ctypes.c_void_p
        L.oracle_measure.argtypes = [dp, ctypes.c_int, ctypes.c_double]
        L.oracle_measure.restype = ctypes.c_longlong
        L.oracle_draw_randn.restype = ctypes.c_double
        L.oracle_putb.argtypes = [ctypes.c_longlong, ctypes.c_int, ctypes.c_char_p]
        _LIB = L
    return _LIB


def _dp(a: np.ndarray):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


def apply_1q(state: np.ndarray, n: int, U, target: int) -> None:
    """In place.  `U` is the reference's argument: the loop applies its TRANSPOSE (quantum_simulator.c:88-89)."""
    u = np.ascontiguousarray(np.asarray(U, dtype=np.complex128).reshape(4)).view(np.float64)
    assert state.dtype == np.complex128 and state.flags.c_contiguous and state.size == 1 << n
    lib().oracle_apply_1q(_dp(state.view(np.float64)), n, _dp(u), target)


def apply_cx(state: np.ndarray, n: int, control: int, target: int) -> None:
    assert state.dtype == np.complex128 and state.flags.c_contiguous and state.size == 1 << n
    lib().oracle_apply_cx(_dp(state.view(np.float64)), n, control, target)


def gate_matrix(token: str):
    """-> (kind, 2x2 complex ndarray or None); kind: 0 unknown, 1 qubit, 2 cx, 3 single-qubit gate."""
    u = np.zeros(8)
    kind = lib().oracle_gate_matrix(token.encode(), _dp(u))
    return kind, (u.view(np.complex128).reshape(2, 2).copy() if kind == 3 else None)


def run_qasm(path: str, max_gates: int = -1):
    """-> (n, amplitudes complex128[2^n], seconds, gates_done).  Raises on parse errors."""
    n = ctypes.c_int(0)
    secs = ctypes.c_double(0)
    done = ctypes.c_long(0)
    err = ctypes.c_int(0)
    p = lib().oracle_run_qasm(path.encode(), ctypes.byref(n), ctypes.byref(secs), max_gates,
                              ctypes.byref(done), ctypes.byref(err))
    if not p:
        raise RuntimeError({1: "cannot open circuit file", 2: "allocation failure", 3: "unknown token",
                            4: "operand out of range"}.get(err.value, f"oracle error {err.value}"))
    a = np.ctypeslib.as_array(ctypes.cast(p, ctypes.POINTER(ctypes.c_double)), shape=(2 << n.value,)).copy()
    lib().oracle_free(p)
    return n.value, a.view(np.complex128), secs.value, done.value


def cumulative(state: np.ndarray, n: int) -> np.ndarray:
    """compute_state_cumulative_distribution (quantum_simulator.c:256-268)."""
    p = lib().oracle_cumulative(_dp(np.ascontiguousarray(state).view(np.float64)), n)
    out = np.ctypeslib.as_array(ctypes.cast(p, ctypes.POINTER(ctypes.c_double)), shape=(1 << n,)).copy()
    lib().oracle_free(p)
    return out


def measure(cumul: np.ndarray, n: int, randn: float) -> int:
    return int(lib().oracle_measure(_dp(cumul), n, randn))


def putb(value: int, length: int) -> str:
    buf = ctypes.create_string_buffer(length + 1)
    lib().oracle_putb(value, length, buf)
    return buf.value.decode()


def zero_state(n: int) -> np.ndarray:
    s = np.zeros(1 << n, dtype=np.complex128)
    s[0] = 1.0
    return s


# ---- the real reference, when oracle/_ref was built in the container --------------------------------

def have_reference() -> bool:
    return os.path.exists(os.path.join(HERE, "_ref", "libqsref.so"))


def reference_lib() -> ctypes.CDLL:
    global _REF
    if _REF is None:
        R = ctypes.CDLL(os.path.join(HERE, "_ref", "libqsref.so"))
        R.compute_state_vector.restype = ctypes.c_void_p
        R.compute_state_vector.argtypes = [ctypes.c_char_p, ctypes.POINTER(ctypes.c_int)]
        dp = ctypes.POINTER(ctypes.c_double)
        R.execute_single_qubit_gate.argtypes = [dp, ctypes.c_int, dp, ctypes.c_int]
        R.execute_cnot.argtypes = [dp, ctypes.c_int, ctypes.c_int, ctypes.c_int]
        _REF = R
    return _REF


def reference_run_qasm(path: str):
    """Runs the compiled reference's compute_state_vector (it prints its time line to stdout)."""
    n = ctypes.c_int(0)
    p = reference_lib().compute_state_vector(path.encode(), ctypes.byref(n))
    if not p:
        raise RuntimeError("reference returned NULL")
    a = np.ctypeslib.as_array(ctypes.cast(p, ctypes.POINTER(ctypes.c_double)), shape=(2 << n.value,)).copy()
    ctypes.CDLL(None).free(ctypes.c_void_p(p))
    return n.value, a.view(np.complex128)
