"""Pins the oracle: oracle/cpu_ref.c must reproduce, BIT FOR BIT, the amplitudes the real reference
(quantum_simulator.c compiled by oracle/Makefile `ref`) produced for every committed fixture, and —
when oracle/_ref is present (container) — must equal the reference run live on fresh circuits."""
import json
import os

import numpy as np
import pytest

from gpu_quantum_simulator_amd import circuits


def _cases(golden_dir):
    with open(os.path.join(golden_dir, "MANIFEST.json")) as f:
        return json.load(f)


def test_manifest_lists_reference_samples(golden_dir):
    m = _cases(golden_dir)
    assert m["entanglement"]["n"] == 2 and m["grover_3_18"]["n"] == 6


@pytest.mark.parametrize("name", [
    "entanglement", "grover_3_18", "rand_n3_all_lf", "rand_n5_all_crlf_physical", "rand_n8_all_suffix",
    "rand_n9_clifford_t", "rand_n10_all", "rand_n12_all", "rand_n12_clifford_t_physical"])
def test_oracle_bit_exact_vs_golden(oracle, golden_dir, name):
    want = np.load(os.path.join(golden_dir, name + ".npy"), allow_pickle=False)
    n, got, _, gates = oracle.run_qasm(os.path.join(golden_dir, name + ".qasm"))
    assert n == _cases(golden_dir)[name]["n"]
    assert got.view(np.float64).tobytes() == want.tobytes()
    assert gates > 0


def test_known_answers_from_survey(oracle, golden_dir):
    # values recorded from the reference run
    _, bell, _, _ = oracle.run_qasm(os.path.join(golden_dir, "entanglement.qasm"))
    assert bell[0] == bell[3] == 0.70710678118654746 and bell[1] == bell[2] == 0
    _, g, _, _ = oracle.run_qasm(os.path.join(golden_dir, "grover_3_18.qasm"))
    assert g[3] == complex(0.57407326014844051, -0.41234821420033285)
    assert g[18] == complex(0.57407326014843929, -0.41234821420033274)
    assert int(np.argmax(np.abs(g) ** 2)) == 3


def test_grover_widened_to_18_qubits(oracle, golden_dir):
    want = np.load(os.path.join(golden_dir, "grover_3_18_n18.first64.npy"), allow_pickle=False)
    n, got, _, _ = oracle.run_qasm(os.path.join(golden_dir, "grover_3_18_n18.qasm"))
    assert n == 18
    assert got[:64].view(np.float64).tobytes() == want.tobytes()
    assert not got[64:].any()


def test_truncated_run_counts_gates(oracle, golden_dir):
    n, amps, _, done = oracle.run_qasm(os.path.join(golden_dir, "rand_n8_all_suffix.qasm"), max_gates=7)
    assert done == 7 and n == 8 and abs(np.vdot(amps, amps).real - 1) < 1e-12


def test_gate_table_matches_reference_constants(oracle):
    kind, s = oracle.gate_matrix("s")
    assert kind == 3 and s[1, 1] == complex(6.123233995736766e-17, 1.0)  # cexp(I*PI/2), SURVEY a4
    _, h = oracle.gate_matrix("h")
    assert h[0, 0] == 1.0 / np.sqrt(2.0) and h[1, 1] == -h[0, 0]
    _, rz = oracle.gate_matrix("rz(0.5)")
    assert rz[0, 0] == 1 and rz[1, 1] == complex(np.cos(0.5), np.sin(0.5))
    assert oracle.gate_matrix("cx")[0] == 2 and oracle.gate_matrix("qubit")[0] == 1
    assert oracle.gate_matrix("ccx")[0] == 0 and oracle.gate_matrix("y")[0] == 0


def test_errors(oracle, tmp_path):
    bad = tmp_path / "bad.qasm"
    bad.write_text('OPENQASM 3.0;\ninclude "stdgates.inc";\nqubit[2] q;\nfoo q[0];\n')
    with pytest.raises(RuntimeError, match="unknown token"):
        oracle.run_qasm(str(bad))
    with pytest.raises(RuntimeError, match="cannot open"):
        oracle.run_qasm(str(tmp_path / "missing.qasm"))
    oob = tmp_path / "oob.qasm"
    oob.write_text('OPENQASM 3.0;\ninclude "stdgates.inc";\nqubit[2] q;\nh q[5];\n')
    with pytest.raises(RuntimeError, match="out of range"):
        oracle.run_qasm(str(oob))


def test_cx_same_qubit_is_noop(oracle):
    s = oracle.zero_state(3)
    _, h = oracle.gate_matrix("h")
    for q in range(3):
        oracle.apply_1q(s, 3, h, q)
    before = s.copy()
    oracle.apply_cx(s, 3, 1, 1)  # quantum_simulator.c:99 — no index has bit 1 both clear and set
    assert (s == before).all()


def test_no_trailing_newline_and_blank_lines(oracle, tmp_path):
    a = tmp_path / "a.qasm"
    a.write_text('OPENQASM 3.0;\ninclude "stdgates.inc";\nqubit[3] q;\n\n  h q[0];\n\ncx q[0],q[2];\n   \n')
    b = tmp_path / "b.qasm"
    b.write_text('OPENQASM 3.0;\ninclude "stdgates.inc";\nqubit[3] q;\nh q[0];\ncx q[0], q[2];')
    assert (oracle.run_qasm(str(a))[1] == oracle.run_qasm(str(b))[1]).all()


@pytest.mark.parametrize("seed,n,kw", [(101, 4, {}), (102, 7, {"crlf": True}), (103, 11, {"physical": True}),
                                       (104, 13, {"qubit_style": "suffix"})])
def test_oracle_equals_live_reference(oracle, tmp_path, seed, n, kw):
    if not oracle.have_reference():
        pytest.skip("oracle/_ref not built (no /root/reference on this host)")
    path = circuits.random_circuit_file(str(tmp_path / "c.qasm"), n, 150, seed, "all", **kw)
    rn, ref = oracle.reference_run_qasm(path)
    on, got, _, _ = oracle.run_qasm(path)
    assert rn == on == n
    assert got.view(np.float64).tobytes() == ref.view(np.float64).tobytes()


def test_oracle_kernels_equal_live_reference(oracle):
    if not oracle.have_reference():
        pytest.skip("oracle/_ref not built")
    import ctypes
    rng = np.random.default_rng(5)
    n = 9
    a = (rng.standard_normal(1 << n) + 1j * rng.standard_normal(1 << n)).astype(np.complex128)
    b = a.copy()
    U = (rng.standard_normal(4) + 1j * rng.standard_normal(4)).astype(np.complex128)  # NOT symmetric
    dp = ctypes.POINTER(ctypes.c_double)
    R = oracle.reference_lib()
    for q in range(n):
        oracle.apply_1q(a, n, U, q)
        R.execute_single_qubit_gate(b.view(np.float64).ctypes.data_as(dp), n, U.view(np.float64).ctypes.data_as(dp), q)
        oracle.apply_cx(a, n, q, (q + 3) % n)
        R.execute_cnot(b.view(np.float64).ctypes.data_as(dp), n, q, (q + 3) % n)
    assert a.tobytes() == b.tobytes()


def test_measurement_post_path_equals_live_reference(oracle, golden_dir):
    """quantum_simulator.c:256-293 (dead code in the reference's main, still compiled into oracle/_ref)."""
    if not oracle.have_reference():
        pytest.skip("oracle/_ref not built")
    import ctypes
    R = oracle.reference_lib()
    dp = ctypes.POINTER(ctypes.c_double)
    R.compute_state_cumulative_distribution.argtypes = [dp, ctypes.c_int]
    R.compute_state_cumulative_distribution.restype = ctypes.c_void_p
    R.measurement.argtypes = [dp, ctypes.c_int]
    R.measurement.restype = ctypes.c_longlong
    libc = ctypes.CDLL(None)
    n, amps, _, _ = oracle.run_qasm(os.path.join(golden_dir, "rand_n10_all.qasm"))
    p = R.compute_state_cumulative_distribution(amps.view(np.float64).ctypes.data_as(dp), n)
    ref_cumul = np.ctypeslib.as_array(ctypes.cast(p, dp), shape=(1 << n,)).copy()
    got = oracle.cumulative(amps, n)
    assert got.tobytes() == ref_cumul.tobytes()
    for seed in range(1, 40):
        libc.srand(seed)
        want = R.measurement(ref_cumul.ctypes.data_as(dp), n)
        libc.srand(seed)
        randn = oracle.lib().oracle_draw_randn()
        assert oracle.measure(got, n, randn) == want
    libc.free(ctypes.c_void_p(p))
    assert oracle.putb(5, 6) == "000101" and oracle.putb(63, 6) == "111111"
    # all-zero prefix is skipped (cumul == 0), a draw above the total lands on the last index
    c = np.array([0.0, 0.0, 0.25, 0.25, 1.0, 1.0, 1.0, 1.0])
    assert oracle.measure(c, 3, 0.0) == 2 and oracle.measure(c, 3, 0.25) == 2 and oracle.measure(c, 3, 0.26) == 4
    assert oracle.measure(c, 3, 1.5) == 7
