"""Host logic on the CPU: tokenizer, gate table and the fusion scheduler of libqsim.so (no device calls).

The scheduler's output (fused blocks in launch order) is replayed with numpy and compared with the
oracle running the unfused circuit — this pins the fusion algebra and the reordering rules
 without a GPU."""
import os

import numpy as np
import pytest

from gpu_quantum_simulator_amd import Circuit, circuits, gate_matrix
from gpu_quantum_simulator_amd import _lib
from helpers import np_apply_1q, np_apply_2q, np_apply_cx, random_unitary, replay_schedule

TOL = 1e-12


def test_library_loads_and_exports_every_declared_symbol():
    lib = _lib.load()
    for name in list(_lib.SIGNATURES) + list(_lib.LEGACY_SYMBOLS):
        assert hasattr(lib, name), name


def test_header_and_binding_agree():
    """Every function prototype in include/qsim.h has a ctypes signature and vice versa."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "include", "qsim.h")).read()
    declared = set(re.findall(r"\b(qsim_[a-z0-9_]+)\s*\(", text)) - {"qsim_sched_cb"}
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    legacy = open(os.path.join(root, "include", "qsim_legacy.h")).read()
    for name in _lib.LEGACY_SYMBOLS:
        assert re.search(rf"\b{name}\s*\(", legacy)


def test_geometry_table_file_round_trip(tmp_path):
    """qsim_tune_table_save / _load (the table qsim_tune_circuit fills on a GPU): text lines, malformed ones skipped."""
    lib = _lib.load()
    lib.qsim_tune_table_clear()
    src = tmp_path / "wisdom.txt"
    src.write_text("30 0 12 3 3b00000 6.5600 7.6700 25 20 24 21 23\n"          # five bits: 20, 21, 23, 24, 25
                   "28 1 13 4 30300 1.9000 2.1000 8 17 9 16\n"
                   "30 0 12 3 3b00000 6.5 7.6 25 20 24 21 22\n"                  # 22 is not in the set: skipped
                   "garbage\n")
    assert lib.qsim_tune_table_load(str(src).encode()) == 2 and lib.qsim_tune_table_size() == 2
    out = tmp_path / "out.txt"
    assert lib.qsim_tune_table_save(str(out).encode()) == 0
    lines = sorted(out.read_text().splitlines())
    assert lines == ["28 1 13 4 30300 1.9000 2.1000 8 17 9 16", "30 0 12 3 3b00000 6.5600 7.6700 25 20 24 21 23"]
    assert lib.qsim_tune_table_load(str(tmp_path / "missing.txt").encode()) == -1
    lib.qsim_tune_table_clear()
    assert lib.qsim_tune_table_size() == 0
    # measured schedule choices travel in the same file ("sched" lines: circuit key, scheduler setting), malformed ones are skipped
    src.write_text("sched 1234abcd 0 2 2 32 0 7\nsched 77 1 1 1 0 1\nsched zz\nsched 5 1 -1 1 0 0\n30 0 12 3 3b00000 6.5600 7.6700 25 20 24 21 23\n")
    assert lib.qsim_tune_table_load(str(src).encode()) == 3 and lib.qsim_tune_table_size() == 1
    assert lib.qsim_tune_table_save(str(out).encode()) == 0
    lines = sorted(out.read_text().splitlines())
    assert lines == ["30 0 12 3 3b00000 6.5600 7.6700 25 20 24 21 23", "sched 1234abcd 0 2 2 32 0 7", "sched 77 1 1 1 0 1 0"]  # ... cap, default?, tie-break seed
    lib.qsim_tune_table_clear()
    assert lib.qsim_tune_table_save(str(out).encode()) == 0 and out.read_text() == ""


def test_no_device_means_loud_failure_not_fallback():
    lib = _lib.load()
    if lib.qsim_device_count() > 0:
        pytest.skip("a GPU is present")
    from gpu_quantum_simulator_amd import Simulator
    with pytest.raises(_lib.QsimError, match="no HIP device"):
        Simulator(3)


def test_gate_table_equals_oracle(oracle):
    for tok in ["x", "sx", "z", "s", "sdg", "t", "tdg", "h", "rz(0.25)", "rz(-3.0000000001)", "rz(1e-3)"]:
        kind, want = oracle.gate_matrix(tok)
        got = gate_matrix(tok)
        assert kind == 3 and got is not None
        # the product stores U, the oracle stores the reference's table; all are symmetric
        assert got.tobytes() == want.tobytes(), tok
    assert gate_matrix("cx") is None and gate_matrix("ccx") is None and gate_matrix("rz(") is None


@pytest.mark.parametrize("name", ["entanglement", "grover_3_18", "rand_n5_all_crlf_physical", "rand_n8_all_suffix",
                                  "rand_n9_clifford_t", "rand_n12_clifford_t_physical"])
def test_tokenizer_matches_generator_and_oracle(oracle, golden_dir, name):
    c = Circuit.from_file(os.path.join(golden_dir, name + ".qasm"))
    n, amps, _, gates = oracle.run_qasm(os.path.join(golden_dir, name + ".qasm"))
    assert c.num_qubits == n and len(c) == gates
    # replaying the parsed gate list with the oracle's kernels reproduces the oracle's own run bit for bit
    s = oracle.zero_state(n)
    for i in range(len(c)):
        g = c.gate(i)
        if g[0] == "cx":
            oracle.apply_cx(s, n, g[1], g[2])
        else:
            oracle.apply_1q(s, n, g[2].T, g[1])  # the oracle applies the transpose of its argument
    assert s.tobytes() == amps.tobytes()


def test_counted_header_form():
    text = "3 4\nh q[0];\ncx q[0], q[2];\nrz(0.5) $1;\nt q[2];\nx q[1];\n"  # 5th statement beyond the count
    c = Circuit.from_text(text)
    assert c.num_qubits == 3 and len(c) == 4
    assert c.gate(1) == ("cx", 0, 2) and c.gate(2)[1] == 1


def test_parse_errors():
    hdr = 'OPENQASM 3.0;\ninclude "stdgates.inc";\n'
    with pytest.raises(_lib.QsimError, match="Unknown token: foo") as e:
        Circuit.from_text(hdr + "qubit[2] q;\nfoo q[0];\n")
    assert e.value.code == _lib.ERR_PARSE
    with pytest.raises(_lib.QsimError, match="out of range"):
        Circuit.from_text(hdr + "qubit[2] q;\nh q[2];\n")
    with pytest.raises(_lib.QsimError, match="before the qubit statement"):
        Circuit.from_text(hdr + "h q[0];\n")
    with pytest.raises(_lib.QsimError) as e:
        Circuit.from_file("/nonexistent/file.qasm")
    assert e.value.code == _lib.ERR_OPEN
    assert len(Circuit.from_text(hdr + "qubit[2] q;")) == 0  # empty circuit (nothing after the qubit line) is fine
    with pytest.raises(_lib.QsimError, match="no qubit statement"):
        Circuit.from_text(hdr)


def test_parser_corners_pinned_to_the_oracle(oracle, tmp_path):
    """Two corners of quantum_simulator.c's statement loop (VERDICT r01 'missing' 7), checked against the oracle — and
    against the compiled reference itself where oracle/_ref exists (this container):
      * a second `qubit` statement re-initialises the state (:162-181): earlier gates are void;
      * `qubit` as the LAST statement followed by a newline is an 'Unknown token' error (:147-151,:179-180);
        with no character after it the file is accepted and yields |0...0>."""
    hdr = 'OPENQASM 3.0;\ninclude "stdgates.inc";\n'
    twice = tmp_path / "twice.qasm"
    twice.write_text(hdr + "qubit[3] q;\nh q[0];\ncx q[0], q[2];\nqubit[2] q;\nx q[1];\nh q[0];\n")
    n, want, _, gates = oracle.run_qasm(str(twice))
    c = Circuit.from_file(str(twice))
    assert (c.num_qubits, len(c)) == (n, 2) == (2, 2)
    assert c.gate(0)[:2] == ("u1", 1) and c.gate(1)[:2] == ("u1", 0)
    got = replay_schedule(n, c.schedule(fuse=0))
    assert np.max(np.abs(got - want)) < TOL
    # an operand that was legal under the FIRST register but not under the second is rejected
    with pytest.raises(_lib.QsimError, match="out of range"):
        Circuit.from_text(hdr + "qubit[3] q;\nh q[0];\nqubit[2] q;\nx q[2];\n")

    last_nl = tmp_path / "qubit_last_nl.qasm"
    last_nl.write_text(hdr + "qubit[3] q;\n")
    with pytest.raises(RuntimeError, match="unknown token"):
        oracle.run_qasm(str(last_nl))
    with pytest.raises(_lib.QsimError, match="Unknown token: \n") as e:
        Circuit.from_file(str(last_nl))
    assert e.value.code == _lib.ERR_PARSE
    with pytest.raises(_lib.QsimError, match="Unknown token:  "):
        Circuit.from_text(hdr + "qubit[3] q;\n  ")  # the last blank read is the "token"

    last = tmp_path / "qubit_last.qasm"
    last.write_text(hdr + "qubit[3] q;")
    n, want, _, gates = oracle.run_qasm(str(last))
    c = Circuit.from_file(str(last))
    assert (c.num_qubits, len(c)) == (n, gates) == (3, 0)
    assert want[0] == 1 and not want[1:].any()

    if oracle.have_reference():
        for path, ok in ((twice, True), (last, True), (last_nl, False)):
            if ok:
                rn, ramps = oracle.reference_run_qasm(str(path))
                on, oamps, _, _ = oracle.run_qasm(str(path))
                assert rn == on and ramps.tobytes() == oamps.tobytes()
            else:
                with pytest.raises(RuntimeError, match="NULL"):
                    oracle.reference_run_qasm(str(path))


@pytest.mark.parametrize("fuse", [0, 1, 2, 3])
@pytest.mark.parametrize("n,depth,seed,vocab", [(3, 80, 1, "all"), (6, 300, 2, "all"), (9, 400, 3, "clifford_t"),
                                                (11, 600, 4, "all")])
def test_schedule_replay_equals_oracle(oracle, tmp_path, fuse, n, depth, seed, vocab):
    gates = circuits.random_gates(n, depth, seed, vocab)
    path = circuits.write_qasm(str(tmp_path / "c.qasm"), n, gates)
    _, want, _, _ = oracle.run_qasm(path)
    c = Circuit.from_file(path)
    # small tiles so that grouping, high-bit selection and blocking all get exercised at n <= 11
    sched = c.schedule(fuse=fuse, tile_bits=8, tile_low_bits=4, tile_max_ops=6)
    got = replay_schedule(n, sched)
    assert np.max(np.abs(got - want)) < TOL
    assert sum(s[5] for s in sched) <= depth  # folded gate counts never exceed the input


def test_schedule_counts_every_gate_once_at_level0(tmp_path):
    gates = circuits.random_gates(7, 200, 9, "all")
    c = Circuit.from_gates(7, gates)
    sched = c.schedule(fuse=0)
    assert len(sched) == 200 and all(s[5] == 1 for s in sched)
    assert [s[2] for s in sched] == ["cx" if g[0] == "cx" else "u1" for g in gates]


def test_level2_pairs_follow_reference_example():
    """h q0; cx q0,q1 folds into ONE 4x4 = CX·(I⊗H) (the case A of quantum_simulator_4x4.cu:336-349)."""
    c = Circuit.from_text('OPENQASM 3.0;\ninclude "stdgates.inc";\nqubit q[2];\nh q[0];\ncx q[0], q[1];\n')
    sched = c.schedule(fuse=2)
    assert len(sched) == 1 and sched[0][2] == "u2" and sched[0][3] == (1, 0) and sched[0][5] == 2
    H = gate_matrix("h")
    CX_ctrl_lo = np.array([[1, 0, 0, 0], [0, 0, 0, 1], [0, 0, 1, 0], [0, 1, 0, 0]], dtype=complex)
    assert np.allclose(sched[0][4], CX_ctrl_lo @ np.kron(np.eye(2), H), atol=0, rtol=0)


def test_exact_identity_is_dropped_but_near_identity_is_kept():
    c = Circuit.empty(3)
    X = gate_matrix("x")
    c.append_1q(X, 1)
    c.append_1q(X, 1)  # X·X = I exactly
    eps = np.array([[1, 0], [0, np.exp(1e-4j)]])  # would pass the reference's 1e-3 isIdentity test (B9)
    c.append_1q(eps, 2)
    sched = c.schedule(fuse=2)
    assert len(sched) == 1 and sched[0][3] == (2,) and np.array_equal(sched[0][4], eps)


def test_generic_two_qubit_gates_and_blocking(oracle):
    rng = np.random.default_rng(7)
    n = 10
    c = Circuit.empty(n)
    ref = np.zeros(1 << n, dtype=np.complex128)
    ref[0] = 1
    for _ in range(120):
        r = rng.integers(3)
        if r == 0:
            q = int(rng.integers(n)); U = random_unitary(2, rng)
            c.append_1q(U, q); ref = np_apply_1q(ref, n, U, q)
        elif r == 1:
            a, b = (int(x) for x in rng.choice(n, 2, replace=False))
            c.append_cx(a, b); ref = np_apply_cx(ref, n, a, b)
        else:
            lo, hi = sorted(int(x) for x in rng.choice(n, 2, replace=False))
            U = random_unitary(4, rng)
            c.append_2q(U, hi, lo); ref = np_apply_2q(ref, n, U, hi, lo)
    for fuse in (0, 1, 2, 3):
        got = replay_schedule(n, c.schedule(fuse=fuse, tile_bits=8, tile_low_bits=5, tile_max_ops=4))
        assert np.max(np.abs(got - ref)) < TOL, fuse


def test_plan_reports_fewer_passes_with_more_fusion():
    gates = circuits.random_gates(30, 1000, 20240117 + 30, "all")
    c = Circuit.from_gates(30, gates)
    launches = [c.plan(fuse=f)["launches"] for f in (0, 1, 2, 3)]
    assert launches[0] == 1000 and launches[0] > launches[1] > launches[2] > launches[3]
    p3 = c.plan(fuse=3)
    # A run from |0...0>: the first tile passes only visit the part of the register their gates have reached (2^-18 of it
    # for the first one), later ones read and write the whole state (32 bytes per amplitude); a cluster that is still a
    # bare CX goes through the swap kernel, which moves the control = 1 half only.
    k = p3["kernels"]
    assert p3["gates"] == 1000 and sum(v["launches"] for v in k.values()) == p3["launches"]
    full = ((p3["launches"] - k["cx"]["launches"]) * 32.0 + k["cx"]["launches"] * 16.0) * 2 ** 30
    assert 0.5 * full < p3["algorithmic_bytes"] < 0.8 * full
    assert k["tile"]["launches"] >= p3["launches"] - 1 and k["cx"]["launches"] <= 1
    # a circuit on 6 qubits of a 30-qubit register never leaves the sparse phase: its passes move next to nothing
    small = Circuit.from_gates(30, circuits.random_gates(6, 300, 3, "all")).plan(fuse=3)
    assert small["algorithmic_bytes"] < small["launches"] * 32.0 * 2 ** 12 * 1.01


def test_tile_passes_respect_geometry_limits():
    gates = circuits.random_gates(20, 600, 5, "all")
    c = Circuit.from_gates(20, gates)
    sched = c.schedule(fuse=3, tile_bits=10, tile_low_bits=6, tile_max_ops=5)
    by_pass = {}
    for s in sched:
        by_pass.setdefault(s[0], []).append(s)
    for ops in by_pass.values():
        assert len(ops) <= 5
        if ops[0][1] == "tile":
            # a block may leave qubits it is block-diagonal in outside the tile (they select a sub-block per tile);
            # every other high qubit needs one of the 4 slots
            def mixing(qs, U):
                k = len(qs)
                out = set()
                for a, q in enumerate(qs):  # qs: most significant first
                    bit = 1 << (k - 1 - a)
                    r, c = np.nonzero(U)
                    if np.any((r ^ c) & bit):
                        out.add(q)
                return out
            high = {q for o in ops for q in mixing(o[3], o[4]) if q >= 6}
            assert len(high) <= 4
            # a block: at most 5 qubits it mixes (they must be inside the tile) plus at most 2 selecting ones, and at
            # most 4 entries per row (one LDS trip of the sparse form; dense 4x4 for two qubits)
            for o in ops:
                assert len(mixing(o[3], o[4])) <= 5 and len(o[3]) <= 7
                assert (o[4] != 0).sum(axis=1).max() <= 4


def test_tokenizer_formatting_fuzz_against_the_oracle(oracle, tmp_path):
    """200 seeded files in the shapes quantum_simulator.c:115-254 accepts — blanks, tabs, CR/LF, commas and semicolons as
    separators, `q[k]` / `$k` / other register names, both `qubit` spellings, rz angles in several notations, no final
    newline — parsed by libqsim's tokenizer and by the oracle (the restatement of the reference's fscanf loop): same
    register size, same gate count, and the parsed gate list replayed with the oracle's kernels reproduces the oracle's
    own amplitudes bit for bit."""
    rng = np.random.default_rng(77)
    names1 = ["x", "sx", "z", "s", "sdg", "t", "tdg", "h"]

    def operand(q):
        style = rng.integers(0, 4)
        return [f"q[{q}]", f"${q}", f"reg[{q}]", f"q[ {q}"][style] if style < 3 else f"q[{q}]"

    def angle():
        v = float(rng.uniform(-3.1, 3.1))
        return [repr(v), f"{v:.6f}", f"{v:.3e}", f"{v:+.4f}"][rng.integers(0, 4)]

    for case in range(200):
        n = int(rng.integers(1, 7))
        eol = "\r\n" if rng.random() < 0.3 else "\n"
        sep = lambda: [" ", "  ", "\t", " \t "][rng.integers(0, 4)]
        lines = ["OPENQASM 3.0;", 'include "stdgates.inc";',
                 (f"qubit[{n}] q;" if rng.random() < 0.5 else f"qubit q[{n}];")]
        depth = int(rng.integers(1, 40))
        for _ in range(depth):
            if n >= 2 and rng.random() < 0.25:
                a, b = (int(x) for x in rng.choice(n, 2, replace=False))
                comma = [", ", ",", " , ", " "][rng.integers(0, 4)]
                stmt = f"cx{sep()}{operand(a)}{comma}{operand(b)}"
            elif rng.random() < 0.2:
                stmt = f"rz({angle()}){sep()}{operand(int(rng.integers(0, n)))}"
            else:
                stmt = f"{names1[rng.integers(0, len(names1))]}{sep()}{operand(int(rng.integers(0, n)))}"
            end = [";", "; ", ";;", " ;"][rng.integers(0, 4)]
            lines.append(stmt + end)
        text = eol.join(lines) + (eol if rng.random() < 0.8 else "")
        if rng.random() < 0.2:
            text = text.replace(eol + "h", eol + eol + "  h", 1)  # blank line and indentation
        path = tmp_path / "f.qasm"
        path.write_bytes(text.encode())
        on, amps, _, gates = oracle.run_qasm(str(path))
        c = Circuit.from_file(str(path))
        assert (c.num_qubits, len(c)) == (on, gates), (case, text)
        s = oracle.zero_state(on)
        for i in range(len(c)):
            g = c.gate(i)
            if g[0] == "cx":
                oracle.apply_cx(s, on, g[1], g[2])
            else:
                oracle.apply_1q(s, on, g[2].T, g[1])
        assert s.tobytes() == amps.tobytes(), (case, text)
