"""Parity tests proper: the HIP path (through the C ABI of libqsim.so) against the oracle on the same
inputs.  fp64 tolerance from BASELINE.json's north_star: 1e-10 absolute per amplitude (the kernels use
FMA contraction and fused matrices, so results are not bit-identical to the reference's C arithmetic)."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from gpu_quantum_simulator_amd import Circuit, Simulator, circuits, gate_matrix, run_qasm
from gpu_quantum_simulator_amd import _lib
from helpers import np_apply_1q, np_apply_2q, np_apply_cx, random_unitary

pytestmark = pytest.mark.gpu
TOL = 1e-10  # north_star: "amplitudes within 1e-10 abs for fp64"

GOLDEN = ["entanglement", "grover_3_18", "rand_n3_all_lf", "rand_n5_all_crlf_physical", "rand_n8_all_suffix",
          "rand_n9_clifford_t", "rand_n10_all", "rand_n12_all", "rand_n12_clifford_t_physical"]


def _rand_state(n, seed):
    rng = np.random.default_rng(seed)
    s = rng.standard_normal(1 << n) + 1j * rng.standard_normal(1 << n)
    return (s / np.linalg.norm(s)).astype(np.complex128)


@pytest.mark.parametrize("fuse", [0, 1, 2, 3])
@pytest.mark.parametrize("name", GOLDEN)
def test_golden_fixtures(golden_dir, name, fuse):
    want = np.load(os.path.join(golden_dir, name + ".npy"), allow_pickle=False).view(np.complex128).reshape(-1)
    got = run_qasm(os.path.join(golden_dir, name + ".qasm"), fuse=fuse)
    assert got.shape == want.shape
    assert np.max(np.abs(got - want)) < TOL


@pytest.mark.parametrize("fuse", [0, 3])
def test_grover_widened_to_18_qubits(golden_dir, fuse):
    want = np.load(os.path.join(golden_dir, "grover_3_18_n18.first64.npy"), allow_pickle=False).view(np.complex128).reshape(-1)
    got = run_qasm(os.path.join(golden_dir, "grover_3_18_n18.qasm"), fuse=fuse)
    assert got.size == 1 << 18
    assert np.max(np.abs(got[:64] - want)) < TOL and not got[64:].any()
    assert int(np.argmax(np.abs(got))) in (3, 18)


def test_dense_1q_every_target_bit(oracle):
    """k_gate1_lo (q < 6, shuffle butterfly) and k_gate1_hi (q >= 6) with a NON-symmetric unitary, so the
    orientation (standard U·v; the oracle applies the transpose of its argument) is pinned too."""
    n = 13
    rng = np.random.default_rng(1)
    with Simulator(n, fuse=0) as sim:
        for q in range(n):
            U = random_unitary(2, rng)
            s = _rand_state(n, 100 + q)
            sim.write(s)
            sim.apply_1q(U, q)
            got = sim.read()
            want = s.copy()
            oracle.apply_1q(want, n, U.T, q)
            assert np.max(np.abs(got - want)) < TOL, q
        kinds = sim.stats()["kernels"]
        assert kinds["gate1_lo"]["launches"] == 6 and kinds["gate1"]["launches"] == n - 6


def test_diagonal_1q_every_target_bit(oracle):
    n = 12
    with Simulator(n, fuse=0) as sim:
        for q in range(n):
            for tok in ("t", "z", "rz(0.3)"):
                U = gate_matrix(tok)
                s = _rand_state(n, 200 + q)
                sim.write(s)
                sim.apply_1q(U, q)
                got = sim.read()
                want = s.copy()
                oracle.apply_1q(want, n, U.T, q)
                assert np.max(np.abs(got - want)) < TOL, (q, tok)
            D = np.diag([np.exp(0.7j), np.exp(-0.2j)])  # d0 != 1: the full-sweep diagonal kernel
            s = _rand_state(n, 300 + q)
            sim.write(s)
            sim.apply_1q(D, q)
            assert np.max(np.abs(sim.read() - np_apply_1q(s, n, D, q))) < TOL
        assert sim.stats()["kernels"]["phase"]["launches"] == 4 * n


def test_cx_every_ordered_pair(oracle):
    n = 11
    with Simulator(n, fuse=0) as sim:
        s = _rand_state(n, 7)
        for c in range(n):
            for t in range(n):
                sim.write(s)
                sim.apply_cx(c, t)
                got = sim.read()
                want = s.copy()
                oracle.apply_cx(want, n, c, t)
                assert np.array_equal(got, want), (c, t)  # pure data movement: bit-exact


def test_dense_2q_every_pair():
    """k_gate2_hh for lo >= 6, single-op tile passes otherwise."""
    n = 12
    rng = np.random.default_rng(3)
    with Simulator(n, fuse=0) as sim:
        s = _rand_state(n, 8)
        for hi in range(1, n):
            for lo in range(hi):
                U = random_unitary(4, rng)
                sim.write(s)
                sim.apply_2q(U, hi, lo)
                got = sim.read()
                assert np.max(np.abs(got - np_apply_2q(s, n, U, hi, lo))) < TOL, (hi, lo)
        k = sim.stats()["kernels"]
        assert k["gate2"]["launches"] == 15 and k["tile"]["launches"] == 66 - 15


@pytest.mark.parametrize("n,depth,seed,vocab,opts", [
    (14, 400, 21, "all", {}),
    (16, 600, 22, "clifford_t", {}),
    (17, 500, 23, "all", {"tile_bits": 10, "tile_low_bits": 6}),
    (18, 500, 24, "all", {"tile_bits": 13, "tile_low_bits": 6}),
    (20, 300, 25, "all", {"tile_bits": 11, "tile_low_bits": 5, "tile_max_ops": 3}),
    (19, 400, 27, "all", {"tile_bits": 9, "tile_low_bits": 2}),
    (15, 400, 28, "clifford_t", {"tile_bits": 8, "tile_low_bits": 6}),
    (20, 300, 26, "all", {"grid_cap": 64}),
])
def test_random_circuits_cache_blocked(oracle, tmp_path, n, depth, seed, vocab, opts):
    path = circuits.random_circuit_file(str(tmp_path / "c.qasm"), n, depth, seed, vocab)
    _, want, _, _ = oracle.run_qasm(path)
    got = run_qasm(path, fuse=3, **opts)
    assert np.max(np.abs(got - want)) < TOL


@pytest.mark.parametrize("fuse", [0, 1, 2])
def test_random_circuit_other_fusion_levels(oracle, tmp_path, fuse):
    n = 17
    path = circuits.random_circuit_file(str(tmp_path / "c.qasm"), n, 400, 31 + fuse, "all")
    _, want, _, _ = oracle.run_qasm(path)
    assert np.max(np.abs(run_qasm(path, fuse=fuse) - want)) < TOL


def test_tiny_registers(oracle, tmp_path):
    for n in (1, 2, 3, 5, 6, 7):
        path = circuits.random_circuit_file(str(tmp_path / f"c{n}.qasm"), n, 60, 40 + n, "all")
        _, want, _, _ = oracle.run_qasm(path)
        for fuse in (0, 2, 3):
            assert np.max(np.abs(run_qasm(path, fuse=fuse) - want)) < TOL, (n, fuse)


def test_cx_same_qubit_and_flush_semantics():
    with Simulator(4) as sim:
        H = gate_matrix("h")
        for q in range(4):
            sim.apply_1q(H, q)
        sim.apply_cx(2, 2)  # silent no-op in the reference (quantum_simulator.c:99)
        a = sim.read()
        assert np.allclose(a, 0.25, atol=1e-15)
        assert sim.stats()["gates"] == 5
        with pytest.raises(_lib.QsimError):
            sim.apply_1q(H, 4)
        with pytest.raises(_lib.QsimError):
            sim.apply_cx(0, 9)


def test_n26_against_truncated_oracle(oracle, tmp_path):
    """1 GiB state (beyond the 256 MiB Infinity Cache): 24 gates on the CPU take a few seconds."""
    n = 26
    gates = circuits.random_gates(n, 24, 77, "all")
    path = circuits.write_qasm(str(tmp_path / "c.qasm"), n, gates)
    _, want, _, _ = oracle.run_qasm(path)
    for fuse in (0, 3):
        c = Circuit.from_file(path)
        with Simulator(n, fuse=fuse) as sim:
            sim.run(c)
            got = sim.read()
        assert np.max(np.abs(got - want)) < TOL, fuse


def test_n26_clifford_t_against_truncated_oracle(oracle, tmp_path):
    """The Clifford+T vocabulary of BASELINE configs[2] beyond the Infinity Cache, amplitude by amplitude against the
    oracle (the CPU needs a few seconds for 24 gates at n = 26)."""
    n = 26
    gates = circuits.random_gates(n, 24, 20240117 + n, "clifford_t")
    path = circuits.write_qasm(str(tmp_path / "c.qasm"), n, gates)
    _, want, _, _ = oracle.run_qasm(path)
    c = Circuit.from_file(path)
    for fuse in (0, 3):
        with Simulator(n, fuse=fuse) as sim:
            sim.run(c)
            got = sim.read()
        assert np.max(np.abs(got - want)) < TOL, fuse


def _sample_windows(sim, n, seed, windows=48, width=256):
    """Head, tail and `windows` seeded interior windows of the state, concatenated."""
    rng = np.random.default_rng(seed)
    N = 1 << n
    firsts = [0, N - 4096] + [int(x) for x in rng.integers(4096, N - 4096 - width, windows)]
    widths = [4096, 4096] + [width] * windows
    return np.concatenate([sim.read(f, w) for f, w in zip(firsts, widths)])


@pytest.mark.parametrize("n,vocab", [(28, "clifford_t"), (30, "all")])
def test_full_size_two_paths_agree(n, vocab):
    """The EXACT BASELINE workloads — configs[2] `random Clifford+T, n=28, depth 1000` and configs[3] `random circuit
    n=30, depth 1000` (bench.py's seed) — run twice through independent device paths: fuse 3 (scheduler, merged sparse
    blocks, k_tile) and fuse 0 (one launch per gate statement through k_gate1_hi/lo, k_phase, k_cx, each of which the
    tests above check against the oracle on every target bit).  ~20 000 sampled amplitudes (head, tail, 48 seeded
    interior windows) must agree within 1e-10, and both norms must be 1."""
    depth = 1000
    gates = circuits.random_gates(n, depth, 20240117 + n, vocab)
    c = Circuit.from_gates(n, gates)
    with Simulator(n, fuse=3) as sim:
        sim.run(c)
        assert abs(sim.norm2() - 1.0) < 1e-10
        fused = _sample_windows(sim, n, 7)
        assert sim.stats()["launches"] < 40  # really the cache-blocked path
        sim.reset()
        sim.set_option(_lib.OPT_FUSE, 0)
        sim.reset_stats()
        sim.run(c)
        assert abs(sim.norm2() - 1.0) < 1e-10
        plain = _sample_windows(sim, n, 7)
        st = sim.stats()
        assert st["launches"] >= depth - 5 and st["kernels"]["tile"]["launches"] == 0  # one launch per statement, no tile pass
    assert np.max(np.abs(fused)) > 1e-6  # the windows are not all zeros
    assert np.max(np.abs(fused - plain)) < TOL


@pytest.mark.parametrize("shards", [2, 4, 8])
def test_full_size_sharded_run_agrees_with_the_single_state(shards):
    """BASELINE configs[3] as it is asked for on 2 / 4 / 8 GPUs — `random circuit n=30, depth 1000`, bench.py's seed — through the
    sharded path at full size: P shards on this one device (the links taken out, everything else as on P devices: the planner's
    segmentation, support carried through the exchanges, every re-layout done by the last tile pass in front of it, schedules
    chosen per shard) against the single state's fuse-3 run, on ~20 000 sampled amplitudes in LOGICAL order within 1e-10."""
    from gpu_quantum_simulator_amd import Cluster
    n = 30
    c = Circuit.from_gates(n, circuits.random_gates(n, 1000, 20240117 + n, "all"))
    with Simulator(n, fuse=3, pingpong=0) as sim:
        sim.run(c)
        assert abs(sim.norm2() - 1.0) < 1e-10
        single = _sample_windows(sim, n, 7)
    with Cluster(n, shards, devices=[0] * shards) as cl:
        cl.plan(c)
        cl.run(c)
        assert abs(cl.norm2() - 1.0) < 1e-10
        sharded = _sample_windows(cl, n, 7)
        ex, _ = cl.exchange_stats()
        fused, separate = cl.pack_counts()
        assert ex >= 2 and fused > 0 and separate == 0  # really sharded, and no re-layout needed a sweep of its own
    assert np.max(np.abs(single)) > 1e-6
    assert np.max(np.abs(sharded - single)) < TOL


def test_largest_sharded_config_on_virtual_shards():
    """BASELINE configs[4] — `n=33 fp64 (128 GiB state) across 8 GPUs`, cross-GPU 2-qubit gates straddling the shard boundary —
    through the sharded path at FULL size on this one device: 8 virtual shards of 16 GiB (one 128 GiB state pool + one 128 GiB
    scratch pool; the links taken out, everything else as on 8 devices: planner, support through the exchanges, every
    re-layout on the last tile pass, per-shard schedules) against the single state's fuse-3 run of the same circuit
    (`random circuit (all), n=33, depth 1000`, seed 20240150 = bench.py's), ~20 000 sampled amplitudes in LOGICAL order within
    1e-10, both norms 1.  Two times 128 GiB plus a third 128 GiB do not fit one card, so the single state runs first (in place),
    only its samples are kept, then the cluster.  If the card cannot hold the two pools the same check runs at n=32 / P=8 and says
    so (pytest -rs shows the reason in the skip-less form: the assertion message names the size that ran).
    The relabelling idea the planner inverts: quantum_simulator_4x4_permute.cu:377-434."""
    import torch
    from gpu_quantum_simulator_amd import Cluster
    free_b, total_b = torch.cuda.mem_get_info()
    n = 33 if free_b > 2 * (16 << 33) + total_b // 32 else 32
    shards = 8
    c = Circuit.from_gates(n, circuits.random_gates(n, 1000, 20240117 + n, "all"))
    with Simulator(n, fuse=3, pingpong=0) as sim:
        sim.run(c)
        assert abs(sim.norm2() - 1.0) < 1e-10
        single = _sample_windows(sim, n, 7)
    with Cluster(n, shards, devices=[0] * shards) as cl:
        cl.plan(c)
        cl.run(c)
        assert abs(cl.norm2() - 1.0) < 1e-10
        sharded = _sample_windows(cl, n, 7)
        ex, _ = cl.exchange_stats()
        fused, separate = cl.pack_counts()
        assert ex >= 2 and fused > 0 and separate == 0, (n, ex, fused, separate)
    assert np.max(np.abs(single)) > 1e-7, n
    assert np.max(np.abs(sharded - single)) < TOL, n
    print(f"configs[4] on virtual shards ran at n={n}, P={shards}: {ex} exchanges, re-layouts fused/separate = {fused}/{separate}")


@pytest.mark.parametrize("n", [32, 33])
def test_largest_registers_two_paths_agree(n):
    """The two largest north_star sizes, amplitude by amplitude on sampled windows (until now only norms):
    n = 32 runs its tile passes out of place between two 64 GiB buffers; n = 33 (128 GiB) does the same where the card has
    room for 2 x 128 GiB next to everything else (it does on the 288 GiB boxes of this pool) and in place otherwise — either
    way the same amplitudes.  250 gate statements of the `all` vocabulary through fuse 3 (scheduler, merged
    blocks, k_tile) and through fuse 0 (one launch per statement, kernels that are oracle-checked on every target bit);
    ~20 000 sampled amplitudes within 1e-10, both norms 1."""
    gates = [("h", q) for q in range(n)] + circuits.random_gates(n, 250, 20240117 + n, "all")  # the layer of h: every sampled window is populated
    depth = len(gates)
    c = Circuit.from_gates(n, gates)
    with Simulator(n, fuse=3) as sim:
        sim.run(c)
        assert abs(sim.norm2() - 1.0) < 1e-10
        fused = _sample_windows(sim, n, 11)
        st = sim.stats()
        assert st["launches"] < 20 and st["kernels"]["tile"]["launches"] >= 3
        import torch
        free_b, total_b = torch.cuda.mem_get_info()
        two_buffers = (total_b - free_b) > 1.5 * (16 << n)
        assert two_buffers or n == 33
        sim.reset()
        sim.set_option(_lib.OPT_FUSE, 0)
        sim.reset_stats()
        sim.run(c)
        assert abs(sim.norm2() - 1.0) < 1e-10
        plain = _sample_windows(sim, n, 11)
        st = sim.stats()
        assert st["launches"] >= depth - 5 and st["kernels"]["tile"]["launches"] == 0
    assert np.count_nonzero(np.abs(fused) > 1e-7) > fused.size // 16  # (a second h on a qubit empties half of the windows again)
    assert np.max(np.abs(fused - plain)) < TOL


def _inverse(gates):
    inv = []
    for g in reversed(gates):
        if g[0] == "cx":
            inv.append(g)
        elif g[0] == "rz":
            inv.append(("rz", -g[1], g[2]))
        elif g[0] == "sx":
            inv.extend([g, g, g])  # sx^4 = I
        else:
            inv.append(({"s": "sdg", "sdg": "s", "t": "tdg", "tdg": "t"}.get(g[0], g[0]), g[1]))
    return inv


@pytest.mark.parametrize("n,fuse,depth", [(28, 3, 300), (30, 3, 200), (30, 2, 60)])
def test_full_size_round_trip(n, fuse, depth):
    """BASELINE sizes, size-independent property: circuit followed by its inverse returns |0...0>,
    and the norm is 1 in between."""
    gates = circuits.random_gates(n, depth, 900 + n, "all")
    fwd = Circuit.from_gates(n, gates)
    bwd = Circuit.from_gates(n, _inverse(gates))
    with Simulator(n, fuse=fuse) as sim:
        sim.run(fwd)
        assert abs(sim.norm2() - 1.0) < 1e-10
        mid = sim.read(0, 4)
        assert abs(mid[0]) < 0.999  # the state really moved
        sim.run(bwd)
        assert abs(sim.norm2() - 1.0) < 1e-10
        head = sim.read(0, 1 << 12)
        assert abs(head[0] - 1.0) < TOL and np.max(np.abs(head[1:])) < TOL
        tail = sim.read((1 << n) - 4096, 4096)
        assert np.max(np.abs(tail)) < TOL


def test_linearity_at_n28():
    """U(a|x> + b|y>) = aU|x> + bU|y> on sampled amplitudes."""
    n = 27
    gates = circuits.random_gates(n, 120, 555, "all")
    c = Circuit.from_gates(n, gates)
    outs = []
    with Simulator(n) as sim:
        for prep in ((), (("x", 3),), (("x", 20), ("x", 3))):
            sim.reset()
            for g in prep:
                sim.apply_1q(gate_matrix(g[0]), g[1])
            sim.run(c)
            outs.append(sim.read(12345, 2048))
        # superposition of basis states |8> and |2^20+8>: H on qubit 20 after X on 3
        sim.reset()
        sim.apply_1q(gate_matrix("x"), 3)
        sim.apply_1q(gate_matrix("h"), 20)
        sim.run(c)
        mix = sim.read(12345, 2048)
    want = (outs[1] + outs[2]) / np.sqrt(2.0)
    assert np.max(np.abs(mix - want)) < TOL


def test_cli_matches_reference_surface(oracle, golden_dir, tmp_path):
    exe = _lib.CLI_PATH
    dump = str(tmp_path / "amps.bin")
    env = dict(os.environ, QSIM_DUMP=dump)
    p = subprocess.run([exe, os.path.join(golden_dir, "grover_3_18.qasm"), "10"], capture_output=True, text=True, env=env)
    assert p.returncode == 0
    lines = p.stdout.splitlines()
    assert len(lines) == 1 and float(lines[0]) >= 0 and "." in lines[0]  # exactly one "%lf" line
    want = np.load(os.path.join(golden_dir, "grover_3_18.npy")).view(np.complex128).reshape(-1)
    got = np.fromfile(dump, dtype=np.complex128)
    assert np.max(np.abs(got - want)) < TOL
    # error surface
    p = subprocess.run([exe], capture_output=True, text=True)
    assert p.returncode == 1 and p.stdout.startswith("QUANTUM CIRCUIT SIMULATOR\nUsage: ")
    p = subprocess.run([exe, "/no/such/file.qasm", "1"], capture_output=True, text=True)
    assert p.returncode == 1 and p.stdout == "ERROR: cannot open circuit file\n"
    bad = tmp_path / "bad.qasm"
    bad.write_text('OPENQASM 3.0;\ninclude "stdgates.inc";\nqubit[2] q;\nfoo q[0];\n')
    p = subprocess.run([exe, str(bad), "1"], capture_output=True, text=True)
    assert p.returncode == 1 and p.stdout.startswith("Unknown token: foo\nInput format: ")
    assert p.stdout.endswith("ERROR while parsing quantum circuit\n")


def test_counted_header_file_through_the_cli(oracle, tmp_path):
    """the CUDA variants' `<num_qubit> <num_gates>` file form (quantum_simulator_naive.cu:239-240, gate
    loop :258-397) through bin/qsim and the HIP path, against the oracle on the equivalent OPENQASM text.  A statement
    beyond the announced count is ignored, as the reference's counted loop would."""
    n, depth = 14, 400
    gates = circuits.random_gates(n, depth, 99, "all")
    ref_path = circuits.write_qasm(str(tmp_path / "ref.qasm"), n, gates)
    _, want, _, _ = oracle.run_qasm(ref_path)
    body = circuits.qasm_text(n, gates + [("x", 0)], physical=False).split("\n", 3)[3]  # gate statements only (+1 extra)
    counted = tmp_path / "counted.qasm"
    counted.write_text(f"{n} {depth}\n" + body)
    for fuse in ("0", "3"):
        dump = str(tmp_path / f"amps{fuse}.bin")
        env = dict(os.environ, QSIM_DUMP=dump, QSIM_FUSE=fuse)
        p = subprocess.run([_lib.CLI_PATH, str(counted)], capture_output=True, text=True, env=env)  # argv[1] only, like naive.cu:135-139
        assert p.returncode == 0 and len(p.stdout.splitlines()) == 1, p.stdout
        got = np.fromfile(dump, dtype=np.complex128)
        assert got.size == 1 << n and np.max(np.abs(got - want)) < TOL
    # and through the library entry point
    c = Circuit.from_file(str(counted))
    assert (c.num_qubits, len(c)) == (n, depth)
    with Simulator(n) as sim:
        sim.run(c)
        assert np.max(np.abs(sim.read() - want)) < TOL


def test_legacy_entry_points(oracle, golden_dir, capfd):
    lib = _lib.load()
    lib.compute_state_vector.restype = ctypes.c_void_p
    lib.compute_state_vector.argtypes = [ctypes.c_char_p, ctypes.POINTER(ctypes.c_int)]
    n = ctypes.c_int(0)
    p = lib.compute_state_vector(os.path.join(golden_dir, "rand_n10_all.qasm").encode(), ctypes.byref(n))
    assert p and n.value == 10
    got = np.ctypeslib.as_array(ctypes.cast(p, ctypes.POINTER(ctypes.c_double)), shape=(2 << 10,)).copy().view(np.complex128)
    ctypes.CDLL(None).free(ctypes.c_void_p(p))
    want = np.load(os.path.join(golden_dir, "rand_n10_all.npy")).view(np.complex128).reshape(-1)
    assert np.max(np.abs(got - want)) < TOL

    dp = ctypes.POINTER(ctypes.c_double)
    lib.execute_single_qubit_gate.argtypes = [dp, ctypes.c_int, dp, ctypes.c_int]
    lib.execute_cnot.argtypes = [dp, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    rng = np.random.default_rng(11)
    U = (rng.standard_normal(4) + 1j * rng.standard_normal(4)).astype(np.complex128)  # arbitrary, not symmetric
    a = _rand_state(9, 12)
    b = a.copy()
    for q in (0, 5, 8):
        lib.execute_single_qubit_gate(a.view(np.float64).ctypes.data_as(dp), 9, U.view(np.float64).ctypes.data_as(dp), q)
        oracle.apply_1q(b, 9, U, q)  # same argument, same (transposed) meaning as quantum_simulator.c:88-89
        lib.execute_cnot(a.view(np.float64).ctypes.data_as(dp), 9, q, (q + 4) % 9)
        oracle.apply_cx(b, 9, q, (q + 4) % 9)
    assert np.max(np.abs(a - b)) < 1e-9 * max(1.0, np.max(np.abs(b)))


def test_profile_stats_and_bytes():
    n = 22
    with Simulator(n, fuse=0, profile=True) as sim:
        H = gate_matrix("h")
        sim.reset_stats()
        for _ in range(5):
            sim.apply_1q(H, 10)
        sim.apply_cx(3, 15)
        sim.sync()
        st = sim.stats()
    S = 16.0 * (1 << n)
    assert st["kernels"]["gate1"]["launches"] == 5 and st["kernels"]["gate1"]["bytes"] == 5 * 2 * S
    assert st["kernels"]["cx"]["bytes"] == S
    assert st["kernels"]["gate1"]["ms"] > 0 and st["kernels"]["cx"]["ms"] > 0


@pytest.mark.parametrize("bits", [(0,), (1, 2), (0, 5, 9), (3, 11, 12), (11, 12, 13), (6,), (0, 1, 2, 3)])
def test_pack_bits_layout(bits):
    """qsim_pack_bits: dst[(block << (n-k)) | rest] = src, block = selected bits, rest = the others in order."""
    import torch
    n = 14
    k = len(bits)
    s = _rand_state(n, 50)
    with Simulator(n) as sim:
        sim.write(s)
        dst = torch.zeros((1 << n, 2), dtype=torch.float64, device="cuda")
        torch.cuda.synchronize()  # the fill runs on torch's stream, the pack on the engine's: order them
        sim.pack_bits(bits, dst.data_ptr())
        sim.sync()
        got = dst.cpu().numpy().reshape(-1).view(np.complex128)
    d = np.arange(1 << n, dtype=np.int64)
    rest, blk = d & ((1 << (n - k)) - 1), d >> (n - k)
    keep = [b for b in range(n) if b not in bits]
    src = np.zeros_like(d)
    for i, b in enumerate(keep):
        src |= ((rest >> i) & 1) << b
    for i, b in enumerate(bits):
        src |= ((blk >> i) & 1) << b
    assert np.array_equal(got, s[src])


@pytest.mark.parametrize("bits", [(0,), (3, 11), (0, 5, 9), (11, 12, 13)])
def test_pack_bits_to_separate_blocks_and_buffer_swap(bits):
    """qsim_pack_bits_to writes block b of the same layout to its own destination (here: scattered over a larger torch
    buffer, in reverse order) and qsim_swap_buffer makes a packed buffer the state — the two halves of the one-kernel
    exchange qsim_cluster uses between shards that share a device."""
    import torch
    n = 14
    k = len(bits)
    blk = 1 << (n - k)
    s = _rand_state(n, 51)
    with Simulator(n) as sim:
        sim.write(s)
        ref = torch.zeros((1 << n, 2), dtype=torch.float64, device="cuda")
        big = torch.zeros((2 << n, 2), dtype=torch.float64, device="cuda")
        torch.cuda.synchronize()  # the fills run on torch's stream, the packs on the engine's: order them
        sim.pack_bits(bits, ref.data_ptr())
        starts = [(2 * ((1 << k) - 1 - b)) * blk for b in range(1 << k)]  # reverse order, a gap after every block
        sim.pack_bits_to(bits, [big.data_ptr() + 16 * st for st in starts])
        sim.sync()
        want = ref.cpu().numpy().reshape(-1).view(np.complex128)
        got = big.cpu().numpy().reshape(-1).view(np.complex128)
        for b, st in enumerate(starts):
            assert np.array_equal(got[st:st + blk], want[b * blk:(b + 1) * blk]), b
            assert not got[st + blk:st + 2 * blk].any()  # the gaps stay untouched
        with pytest.raises(_lib.QsimError, match="overlaps the state"):
            sim.pack_bits_to(bits, [sim.device_ptr] * (1 << k))
    # swap: a state created by libqsim (it owns its buffer) trades it for a spare one of the same size
    with Simulator(n) as sim:
        sim.write(s)
        spare_t = torch.zeros((1 << n, 2), dtype=torch.float64, device="cuda")
        torch.cuda.synchronize()
        spare = spare_t.data_ptr()
        old_ptr = sim.device_ptr
        sim.pack_bits_to(bits[:1], [spare, spare + 16 * (1 << (n - 1))])  # spare = the state packed on bits[0]
        was = sim.swap_buffer(spare)
        assert was == old_ptr and sim.device_ptr == spare
        got = sim.read()
        d = np.arange(1 << n, dtype=np.int64)
        rest, top = d & ((1 << (n - 1)) - 1), d >> (n - 1)
        src = ((rest >> bits[0]) << (bits[0] + 1)) | (rest & ((1 << bits[0]) - 1)) | (top << bits[0])
        assert np.array_equal(got, s[src])
        assert sim.swap_buffer(was) == spare  # hand the original back before the state is destroyed (torch owns `spare`)
        assert np.array_equal(sim.read(), s)


def test_masked_block_sums_and_gather():
    """qsim_block_prob_masked / qsim_gather_masked: blocks that are bit-deposits instead of ranges, against numpy."""
    n = 13
    s = _rand_state(n, 52)
    rng = np.random.default_rng(3)

    def deposit(x, mask):
        out, j = 0, 0
        for b in range(64):
            if mask >> b & 1:
                out |= ((x >> j) & 1) << b
                j += 1
        return out

    with Simulator(n) as sim:
        sim.write(s)
        for _ in range(6):
            bits = rng.permutation(n)
            nlo = int(rng.integers(1, 9))
            lo_mask = sum(1 << int(b) for b in bits[:nlo])
            hi_mask = sum(1 << int(b) for b in bits[nlo:])
            sums = sim.block_prob_masked(hi_mask, lo_mask)
            idx = np.array([[deposit(w, hi_mask) | deposit(i, lo_mask) for i in range(1 << nlo)] for w in range(1 << (n - nlo))])
            want = (np.abs(s[idx]) ** 2).sum(axis=1)
            assert np.max(np.abs(sums - want)) < 1e-15
            assert abs(sums.sum() - 1.0) < 1e-13
            w = int(rng.integers(0, 1 << (n - nlo)))
            assert np.array_equal(sim.gather_masked(deposit(w, hi_mask), lo_mask), s[idx[w]])
        with pytest.raises(_lib.QsimError, match="disjoint"):
            sim.block_prob_masked(0b11, 0b110)


def test_rccl_rank_comm_on_one_gpu():
    """The native RCCL call sites (csrc/dist.cpp) with the one GPU present: a 1-rank communicator from
    ncclGetUniqueId / ncclCommInitRank, shard data through ncclSend -> ncclRecv on the engine's stream (loopback), and
    the argument checks of the exchange entry point.  Exchanges between ranks need more GPUs than this box has; their
    data path is the same pack + send/recv and is covered by the gloo tests (plan) and the virtual-shard tests (layout)."""
    from gpu_quantum_simulator_amd import RankComm
    n = 16
    s = _rand_state(n, 53)
    uid = RankComm.unique_id()
    assert len(uid) == 128 and any(uid)
    with Simulator(n) as sim:
        sim.write(s)
        comm = RankComm(sim, 0, 1, 0, uid)
        comm.loopback(2 << n)       # the whole shard
        comm.loopback(1024)
        assert np.array_equal(sim.read(), s)
        with pytest.raises(_lib.QsimError, match="unsupported"):
            comm.exchange([0], [3])  # a single rank has nobody to exchange with
        assert comm.stats() == (0, 0.0, 0.0)
        comm.close()
    assert b"ncclSend" in subprocess.run(["nm", "-D", _lib.LIB_PATH], capture_output=True).stdout


@pytest.mark.parametrize("world,opts", [(2, {}), (4, {}), (8, {}), (4, {"pingpong": 2, "tile_bits": 9})])
def test_virtual_shards_on_one_gpu_equal_oracle(oracle, tmp_path, world, opts):
    """The sharded path (planner, per-rank gates, pack kernel, block exchange) with P shards on ONE device,
    against the oracle on the whole register.  Last case: every shard's tile passes run out of place between its state
    and its exchange scratch, which the shard lends to the engine between exchanges (HipShard)."""
    from gpu_quantum_simulator_amd.distributed import VirtualCluster
    n = 16
    gates = circuits.random_gates(n, 500, 60 + world, "all")
    path = circuits.write_qasm(str(tmp_path / "c.qasm"), n, gates)
    _, want, _, _ = oracle.run_qasm(path)
    vc = VirtualCluster(n, world, gates, **opts)
    try:
        vc.run()
        got = vc.gather_logical()
    finally:
        vc.close()
    assert np.max(np.abs(got - want)) < TOL
    assert vc.plans[0].exchanges >= 1


def test_single_rank_process_group_bench_path(oracle, tmp_path):
    """ShardedSimulator under torch.distributed with world_size 1 (nccl): the bench's N>1 code path minus peers."""
    import torch
    import torch.distributed as dist
    from gpu_quantum_simulator_amd.distributed import ShardedSimulator, logical_from_physical
    n = 15
    gates = circuits.random_gates(n, 300, 71, "all")
    path = circuits.write_qasm(str(tmp_path / "c.qasm"), n, gates)
    _, want, _, _ = oracle.run_qasm(path)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        sim = ShardedSimulator(n, gates, device=0)
        rep = sim.tune(max_candidates=4, budget_ms=0)  # the planning step bench.py runs on every rank
        assert rep is not None and rep["tile_passes"] >= 1
        sim.run_step()
        sim.run_step()
        assert abs(sim.norm2() - 1.0) < 1e-10
        got = logical_from_physical(sim.shard.read_all(), sim.plan.final_pos)
        assert np.max(np.abs(got - want)) < TOL
        assert abs(sim.amplitude(7) - want[7]) < TOL
        cumul = oracle.cumulative(want, n)  # the collective measurement post-path, here with one rank over RCCL
        draws = np.random.default_rng(8).uniform(0, 1, 25)
        for r, g in zip(draws, sim.sample(draws)):
            w = oracle.measure(cumul, n, float(r))
            assert int(g) == w or (abs(int(g) - w) == 1 and min(abs(cumul[w] - r), abs(cumul[int(g)] - r)) < 1e-13)
        sim.close()
    finally:
        dist.destroy_process_group()


def _perm_phase(rng, dim):
    m = np.zeros((dim, dim), dtype=np.complex128)
    for r, c in enumerate(rng.permutation(dim)):
        m[r, c] = np.exp(1j * rng.uniform(-np.pi, np.pi))
    return m


@pytest.mark.parametrize("qubits", [(13, 9, 4), (12, 5, 0), (2, 1, 0), (13, 12, 11), (7, 3, 2), (10, 6, 1)])
def test_sparse_block_forms_in_tile_passes(qubits):
    """Every TOP_SP form (2 and 3 qubits; 1, 2 and 4 entries per row; identity rows) on low, mixed and high
    tile-local bits: two-qubit gates with exact-zero structure on overlapping pairs are merged by the scheduler
    into sparse 3-qubit blocks, replayed here with numpy."""
    from helpers import np_apply_kq
    n = 14
    a, b, c = qubits
    rng = np.random.default_rng(sum(qubits))
    H = gate_matrix("h")
    cx_hi = np.eye(4)[[0, 1, 3, 2]].astype(np.complex128)             # control = high qubit: two identity rows
    pair = cx_hi @ np.kron(np.diag([1, np.exp(0.3j)]), H)             # two 2x2 blocks
    gates = [(_perm_phase(rng, 4), (a, b)), (_perm_phase(rng, 4), (b, c)),   # monomial x monomial -> 3q, 1 entry/row
             (pair, (a, b)), (_perm_phase(rng, 4), (b, c)),                  # pair x monomial     -> 3q, 2 entries/row
             (pair, (a, b)), (pair, (b, c)),                                 # pair x pair         -> 3q, 4 entries/row
             (cx_hi, (a, c)), (cx_hi, (b, c)),                               # bare CXs: identity rows
             (random_unitary(4, rng), (a, c))]                               # dense 4x4
    s0 = _rand_state(n, 77)
    want = s0.copy()
    with Simulator(n, fuse=3, tile_bits=10, tile_low_bits=3) as sim:
        sim.write(s0)
        for U, (hi, lo) in gates:
            sim.apply_2q(U, hi, lo)
            want = np_apply_kq(want, n, U, (hi, lo))
        got = sim.read()
        st = sim.stats()
    assert np.max(np.abs(got - want)) < TOL
    assert st["launches"] <= 3 and st["kernels"]["tile"]["launches"] >= 1  # the nine gates ran as one or two tile passes


def test_largest_single_gpu_register():
    """n = 33 (128 GiB): 64-bit indexing end to end; H on the top and bottom qubits, a CX across, norm and samples."""
    n = 33
    H = gate_matrix("h")
    with Simulator(n, fuse=3) as sim:
        sim.apply_1q(H, n - 1)
        sim.apply_1q(H, 0)
        sim.apply_cx(n - 1, 17)
        sim.apply_1q(gate_matrix("t"), 17)
        assert abs(sim.norm2() - 1.0) < 1e-12
        lo = sim.read(0, 2)
        hi = sim.read((1 << (n - 1)) | (1 << 17), 2)
        assert np.allclose(lo, 0.5, atol=1e-15)
        assert np.allclose(hi, 0.5 * np.exp(0.25j * np.pi), atol=1e-15)
        assert abs(sim.read((1 << (n - 1)), 1)[0]) < 1e-15


def test_measurement_post_path(oracle, golden_dir, tmp_path):
    """qsim_sample against the oracle's cumulative distribution + search (quantum_simulator.c:256-283)."""
    n = 14
    path = circuits.random_circuit_file(str(tmp_path / "c.qasm"), n, 300, 91, "all")
    _, want_state, _, _ = oracle.run_qasm(path)
    cumul = oracle.cumulative(want_state, n)
    rng = np.random.default_rng(3)
    draws = np.concatenate([rng.uniform(0, 1, 300), [0.0, 1e-300, 0.5, 1.0, 1.5]])
    c = Circuit.from_file(path)
    with Simulator(n) as sim:
        sim.run(c)
        got = sim.sample(draws)
    mism = 0
    for r, g in zip(draws, got):
        w = oracle.measure(cumul, n, float(r))
        if int(g) != w:  # only allowed when the draw sits on a boundary to within rounding of the summation order
            assert abs(int(g) - w) == 1 and min(abs(cumul[w] - r), abs(cumul[int(g)] - r)) < 1e-13
            mism += 1
    assert mism <= 2
    # a state with leading zero amplitudes: cumul == 0 entries are skipped
    with Simulator(6) as sim:
        sim.apply_1q(gate_matrix("x"), 5)
        assert list(sim.sample([0.0, 0.3, 1.0])) == [32, 32, 32]
    # CLI: the lines the reference has commented out, behind QSIM_MEASURE
    env = dict(os.environ, QSIM_MEASURE="1")
    p = subprocess.run([_lib.CLI_PATH, os.path.join(golden_dir, "entanglement.qasm"), "5"], capture_output=True, text=True, env=env)
    lines = p.stdout.splitlines()
    assert p.returncode == 0 and len(lines) == 6 and float(lines[0]) >= 0
    assert all(l in ("MEASUREMENT: 00 (0)", "MEASUREMENT: 11 (3)") for l in lines[1:])


@pytest.mark.parametrize("shards,opts", [(2, {}), (4, {}), (8, {}), (16, {}), (4, {"pingpong": 2, "tile_bits": 9})])
def test_c_host_cluster_virtual_shards(oracle, tmp_path, shards, opts):
    """qsim_cluster (csrc/dist.cpp): the C host's one-process sharded path, all shards on device 0.  Sixteen shards swap
    up to four qubits at once, more than the one-kernel exchange takes (eight block destinations): those exchanges go
    through the pack + copy form, in the same run as one-kernel ones."""
    from gpu_quantum_simulator_amd import Cluster
    n = 17
    path = circuits.random_circuit_file(str(tmp_path / "c.qasm"), n, 600, 80 + shards, "all")
    _, want, _, _ = oracle.run_qasm(path)
    c = Circuit.from_file(path)
    with Cluster(n, shards, devices=[0] * shards, **opts) as cl:  # last case: out-of-place tile passes + buffer swaps at exchanges
        for _ in range(2):  # a second run must start from a clean state and map
            cl.run(c)
        assert abs(cl.norm2() - 1.0) < 1e-10
        got = cl.read()
        assert np.max(np.abs(got - want)) < TOL
        assert np.max(np.abs(cl.read(12345, 7) - want[12345:12352])) < TOL  # the one-by-one path
        ex, nbytes = cl.exchange_stats()
        assert ex >= 2 and nbytes > 0
        assert cl.exchange_mode == "direct"  # every shard on this one device: pack straight into the members' buffers


def test_cli_sharded(oracle, golden_dir, tmp_path):
    dump = str(tmp_path / "amps.bin")
    env = dict(os.environ, QSIM_SHARDS="4", QSIM_DUMP=dump, QSIM_STATS="1")
    p = subprocess.run([_lib.CLI_PATH, os.path.join(golden_dir, "rand_n12_all.qasm"), "1"], capture_output=True, text=True, env=env)
    assert p.returncode == 0 and len(p.stdout.splitlines()) == 1 and '"shards": 4' in p.stderr
    want = np.load(os.path.join(golden_dir, "rand_n12_all.npy")).view(np.complex128).reshape(-1)
    assert np.max(np.abs(np.fromfile(dump, dtype=np.complex128) - want)) < TOL


def test_edge_cases_empty_tiny_and_auto_flush(oracle, tmp_path):
    hdr = 'OPENQASM 3.0;\ninclude "stdgates.inc";\n'
    # empty circuit: |0...0> untouched (and the lazily written state is materialised by the read)
    p = tmp_path / "empty.qasm"
    p.write_text(hdr + "qubit[5] q;")  # no newline: with one the reference reports an unknown token (test_parser_corners...)
    a = run_qasm(str(p))
    assert a[0] == 1 and not a[1:].any()
    # one-qubit register, every fuse level
    p = tmp_path / "one.qasm"
    p.write_text(hdr + "qubit[1] q;\nh q[0];\nt q[0];\nsx q[0];\nrz(0.37) q[0];\nx q[0];\n")
    _, want, _, _ = oracle.run_qasm(str(p))
    for fuse in (0, 1, 2, 3):
        assert np.max(np.abs(run_qasm(str(p), fuse=fuse) - want)) < TOL
    # zero-qubit register: a single amplitude
    with Simulator(0) as sim:
        assert sim.read()[0] == 1
    # long circuit with a tiny pending queue: many automatic flushes, options changed mid-stream
    n = 15
    gates = circuits.random_gates(n, 3000, 123, "all")
    path = circuits.write_qasm(str(tmp_path / "long.qasm"), n, gates)
    _, want, _, _ = oracle.run_qasm(path)
    c = Circuit.from_file(path)
    with Simulator(n, max_pending=37) as sim:
        sim.run(c, 0, 1000)
        sim.set_option(_lib.OPT_FUSE, 1)
        sim.run(c, 1000, 1000)
        sim.set_option(_lib.OPT_FUSE, 3)
        sim.set_option(_lib.OPT_TILE_BITS, 9)
        sim.run(c, 2000, -1)
        assert np.max(np.abs(sim.read() - want)) < TOL
        assert sim.stats()["gates"] == 3000
    # reset in the middle of queued work drops it
    with Simulator(6) as sim:
        sim.apply_1q(gate_matrix("h"), 3)
        sim.reset()
        a = sim.read()
        assert a[0] == 1 and not a[1:].any()


@pytest.mark.parametrize("n,depth,seed,vocab,opts", [(16, 500, 31, "all", {}), (20, 700, 32, "all", {}), (14, 400, 33, "clifford_t", {}),
                                                   (18, 600, 34, "all", {"tile_bits": 11, "tile_low_bits": 3}),
                                                   (17, 500, 35, "all", {"tile_bits": 13, "tile_low_bits": 4})])
def test_tile_bit_order_does_not_change_results(oracle, tmp_path, n, depth, seed, vocab, opts):
    """The engine may walk a pass's high tile bits in any order (TileGeom::high need not ascend): every block form must
    translate its qubits through the order it is given.  QSIM_OPT_DEBUG_TILE_ORDER shuffles every pass's order."""
    gates = circuits.random_gates(n, depth, seed, vocab)
    path = circuits.write_qasm(str(tmp_path / "c.qasm"), n, gates)
    _, want, _, _ = oracle.run_qasm(path)
    c = Circuit.from_file(path)
    for k in (1, 2, 3):
        with Simulator(n, fuse=3, debug_tile_order=k, profile=True, **opts) as sim:
            sim.run(c)
            got = sim.read()
            orders = [o for o in sim.launch_log_orders() if o]
        assert np.max(np.abs(got - want)) < TOL, k
        assert any(o != sorted(o) for o in orders)  # the orders really were shuffled


def test_geometry_planning_keeps_results_and_fills_the_table(oracle, tmp_path):
    """qsim_tune_circuit: measures candidate orders for every pass of the schedule, leaves the state reset, and a later
    run of the same circuit (which now uses the measured orders) still equals the oracle."""
    lib = _lib.load()
    n, depth = 20, 600
    gates = circuits.random_gates(n, depth, 36, "all")
    path = circuits.write_qasm(str(tmp_path / "c.qasm"), n, gates)
    _, want, _, _ = oracle.run_qasm(path)
    c = Circuit.from_file(path)
    lib.qsim_tune_table_clear()
    with Simulator(n, fuse=3, profile=True) as sim:
        sim.apply_1q(gate_matrix("h"), 3)  # pending work is flushed first, then the state comes back reset
        rep = sim.tune(c, max_candidates=6, budget_ms=0)
        assert rep["tile_passes"] >= 2 and rep["passes_tuned"] >= 2 and rep["candidates_timed"] >= rep["passes_tuned"]
        assert rep["ms_best"] <= rep["ms_ascending"] + 1e-9
        assert lib.qsim_tune_table_size() == rep["passes_tuned"]
        a = sim.read(0, 16)
        assert a[0] == 1 and not a[1:].any()
        sim.reset_stats()
        sim.run(c)
        assert np.max(np.abs(sim.read() - want)) < TOL
        again = sim.tune(c, max_candidates=6, budget_ms=0)
        assert again["passes_tuned"] == 0 and again["already_known"] == rep["passes_tuned"]
        # the planning travels: saved, forgotten, loaded — the measured schedule choice ("sched" line) and the geometries are
        # back, a third planning call measures nothing, and the run is still the oracle's
        wisdom = str(tmp_path / "wisdom.txt").encode()
        assert lib.qsim_tune_table_save(wisdom) == 0
        assert any(ln.startswith("sched ") for ln in open(wisdom.decode()).read().splitlines())
        lib.qsim_tune_table_clear()
        assert lib.qsim_tune_table_load(wisdom) == rep["passes_tuned"] + 1
        third = sim.tune(c, max_candidates=6, budget_ms=0)
        assert third["passes_tuned"] == 0 and third["candidates_timed"] == 0 and third["already_known"] == rep["passes_tuned"]
        sim.run(c)
        assert np.max(np.abs(sim.read() - want)) < TOL
    lib.qsim_tune_table_clear()
    assert lib.qsim_tune_table_size() == 0


def test_plan_cache_replays_identical_queues_only(oracle, tmp_path):
    """QSIM_OPT_PLAN_CACHE: a gate queue that was planned before is replayed from the cached plan (same launches, same
    amplitudes); a queue that differs in one matrix entry, one qubit or one option is planned anew; the planning step
    (qsim_tune_circuit) invalidates plans built with older bit orders."""
    n, depth = 18, 500
    gates = circuits.random_gates(n, depth, 61, "all")
    path = circuits.write_qasm(str(tmp_path / "c.qasm"), n, gates)
    _, want, _, _ = oracle.run_qasm(path)
    c = Circuit.from_file(path)
    gates2 = list(gates)
    gates2[137] = ("rz", 0.123456, 5)
    path2 = circuits.write_qasm(str(tmp_path / "c2.qasm"), n, gates2)
    _, want2, _, _ = oracle.run_qasm(path2)
    c2 = Circuit.from_file(path2)
    with Simulator(n, fuse=3, profile=True) as sim:
        logs = []
        for rep in range(3):
            sim.reset()
            sim.reset_stats()
            sim.run(c)
            assert np.max(np.abs(sim.read() - want)) < TOL, rep
            logs.append([(k, nops, hm) for k, nops, hm, _ in sim.launch_log()])
        assert logs[0] == logs[1] == logs[2] and len(logs[0]) >= 2
        # the block forms of the same launches (qsim_launch_log_blocks): one per block of a tile pass, replays included
        sim.set_option(_lib.OPT_PROFILE, 2)
        sim.reset(); sim.reset_stats(); sim.run(c); sim.sync()
        assert [(k, nops, hm) for k, nops, hm, _ in sim.launch_log()] == logs[2]
        for (k, nops, _), forms in zip(logs[2], sim.launch_log_blocks()):
            assert (len(forms) <= nops and len(forms) >= 1) if k == "tile" else forms == []
            assert all(t in (1, 2, 4) and 1 <= q <= 6 and 0 <= sel <= 2 for t, q, _, sel in forms)
        sim.reset(); sim.run(c2)
        assert np.max(np.abs(sim.read() - want2)) < TOL          # one gate changed: not the cached plan
        sim.reset(); sim.run(c)
        assert np.max(np.abs(sim.read() - want)) < TOL           # and back (both plans are cached now)
        sim.set_option(_lib.OPT_TILE_BITS, 10)                   # an option that shapes the plan
        sim.reset(); sim.run(c)
        assert np.max(np.abs(sim.read() - want)) < TOL
        sim.set_option(_lib.OPT_TILE_BITS, 12)
        rep = sim.tune(c, max_candidates=5, budget_ms=0)         # new bit orders: plans built before are stale
        assert rep["passes_tuned"] >= 1
        sim.run(c)
        assert np.max(np.abs(sim.read() - want)) < TOL
        sim.reset(); sim.run(c)                                  # replay of the plan built after tuning
        assert np.max(np.abs(sim.read() - want)) < TOL
        sim.set_option(_lib.OPT_PLAN_CACHE, 0)
        sim.reset(); sim.run(c)
        assert np.max(np.abs(sim.read() - want)) < TOL
        # more distinct queues than the cache holds: eviction, then the first one again
        sim.set_option(_lib.OPT_PLAN_CACHE, 1)
        for seed in range(10):
            g = circuits.random_gates(n, 60, 500 + seed, "all")
            pth = circuits.write_qasm(str(tmp_path / f"e{seed}.qasm"), n, g)
            _, w, _, _ = oracle.run_qasm(pth)
            cc = Circuit.from_file(pth)
            for _ in range(2):
                sim.reset(); sim.run(cc)
                assert np.max(np.abs(sim.read() - w)) < TOL, seed
        sim.reset(); sim.run(c)
        assert np.max(np.abs(sim.read() - want)) < TOL
    _lib.load().qsim_tune_table_clear()


def test_randomised_geometry_sweep(oracle, tmp_path):
    """Seeded sweep over register sizes, vocabularies and every engine option (tile size, low bits, ops per pass,
    threads, padding start, grid cap): each case against the oracle.  Catches geometry corner cases (n just above the
    tile size, tiles wider than the register, under-filled tiles) that the fixed cases above do not hit."""
    rng = np.random.default_rng(20240117)
    worst = 0.0
    for case in range(60):
        n = int(rng.integers(2, 21))
        depth = int(rng.integers(20, 400))
        vocab = "all" if rng.random() < 0.7 else "clifford_t"
        tile_bits = int(rng.integers(8, 14))
        tile_low = int(rng.integers(max(2, tile_bits - 10), min(6, tile_bits - 2) + 1))
        opts = {"tile_bits": tile_bits, "tile_low_bits": tile_low, "tile_max_ops": int(rng.integers(1, 40)),
                "tile_pad_from": int(rng.integers(-1, 20))}
        if tile_bits >= 12 and rng.random() < 0.5:
            opts["tile_threads"] = int(rng.choice([256, 512, 1024] if tile_bits == 12 else [512, 1024]))
        if rng.random() < 0.3:
            opts["grid_cap"] = int(rng.integers(1, 64))
        fuse = int(rng.choice([0, 1, 2, 3, 3, 3]))
        path = circuits.random_circuit_file(str(tmp_path / f"f{case}.qasm"), n, depth, 5000 + case, vocab)
        _, want, _, _ = oracle.run_qasm(path)
        got = run_qasm(path, fuse=fuse, **opts)
        err = float(np.max(np.abs(got - want)))
        assert err < TOL, (case, n, depth, vocab, fuse, opts, err)
        worst = max(worst, err)
    assert worst < 1e-12


@pytest.mark.parametrize("opts", [{}, {"tile_bits": 11, "tile_low_bits": 3}, {"tile_bits": 13, "tile_low_bits": 4},
                                  {"tile_bits": 10, "tile_low_bits": 2}, {"tile_bits": 12, "tile_low_bits": 3, "tile_max_ops": 64}])
def test_out_of_tile_selectors_and_wide_blocks(oracle, tmp_path, opts):
    """Blocks that are block-diagonal in a qubit leave it outside the tile (coefficient banks picked per tile), and
    neighbouring blocks are merged on up to 5 tile qubits.  A circuit built to stress both: controls and diagonal gates
    on the TOP qubits of a 22-qubit register (never tile qubits unless mixed), targets and Hadamards spread below, long
    CX ladders over 4-5 adjacent qubits so that wide merged blocks appear."""
    n = 22
    rng = np.random.default_rng(2024)
    lines = ["OPENQASM 3.0;", 'include "stdgates.inc";', f"qubit[{n}] q;"]
    for q in range(n):
        lines.append(f"h q[{q}];")
    for rep in range(40):
        top = int(rng.integers(14, n))
        for _ in range(6):
            t = int(rng.integers(0, 14))
            lines.append(f"cx q[{top}], q[{t}];")
            lines.append(f"{rng.choice(['t', 's', 'z', 'tdg'])} q[{top}];")
            lines.append(f"rz({rng.uniform(-3, 3)!r}) q[{int(rng.integers(14, n))}];")
            lines.append(f"{rng.choice(['h', 'sx', 'x', 't'])} q[{t}];")
        a = int(rng.integers(3, 10))
        for k in range(4):  # ladder on a..a+4
            lines.append(f"cx q[{a + k}], q[{a + k + 1}];")
            lines.append(f"t q[{a + k + 1}];")
        lines.append(f"h q[{top}];")  # now the top qubit is mixed: it has to enter a tile
    path = str(tmp_path / "sel.qasm")
    with open(path, "w") as f:
        f.write("\n".join(lines) + "\n")
    _, want, _, _ = oracle.run_qasm(path)
    got = run_qasm(path, fuse=3, **opts)
    assert np.max(np.abs(got - want)) < TOL
    # the schedule really uses both features (checked on the host: no GPU needed for this part)
    sched = Circuit.from_file(path).schedule(fuse=3, tile_bits=opts.get("tile_bits", 12), tile_low_bits=opts.get("tile_low_bits", 3))
    tile_ops = [s for s in sched if s[1] == "tile"]
    assert any(len(s[3]) >= 4 for s in tile_ops)
    if opts.get("tile_bits", 12) >= 11:
        def mixing(qs, U):
            r, c = np.nonzero(U)
            return [q for a, q in enumerate(qs) if np.any((r ^ c) & (1 << (len(qs) - 1 - a)))]
        assert any(len(mixing(s[3], s[4])) < len(s[3]) for s in tile_ops)


@pytest.mark.parametrize("shards", [2, 8])
def test_measurement_on_sharded_state(oracle, golden_dir, tmp_path, shards):
    """qsim_cluster_sample: measurement() in LOGICAL order on a state whose qubit map was permuted by exchanges."""
    from gpu_quantum_simulator_amd import Cluster
    n = 15
    path = circuits.random_circuit_file(str(tmp_path / "c.qasm"), n, 500, 93, "all")
    _, want_state, _, _ = oracle.run_qasm(path)
    cumul = oracle.cumulative(want_state, n)
    rng = np.random.default_rng(4)
    draws = np.concatenate([rng.uniform(0, 1, 200), [0.0, 1e-300, 0.5, 1.0, 1.5]])
    with Cluster(n, shards, devices=[0] * shards) as cl:
        cl.run(Circuit.from_file(path))
        ex, _ = cl.exchange_stats()
        assert ex >= 1  # the map is not the identity any more
        got = cl.sample(draws)
    mism = 0
    for r, g in zip(draws, got):
        w = oracle.measure(cumul, n, float(r))
        if int(g) != w:
            assert abs(int(g) - w) == 1 and min(abs(cumul[w] - r), abs(cumul[int(g)] - r)) < 1e-13
            mism += 1
    assert mism <= 2
    # CLI, sharded, behind QSIM_MEASURE
    env = dict(os.environ, QSIM_MEASURE="1", QSIM_SHARDS=str(shards))
    import re
    p = subprocess.run([_lib.CLI_PATH, os.path.join(golden_dir, "grover_3_18.qasm"), "40"], capture_output=True, text=True, env=env)
    lines = p.stdout.splitlines()
    assert p.returncode == 0 and len(lines) == 41 and float(lines[0]) >= 0
    hits = 0
    for l in lines[1:]:
        m = re.fullmatch(r"MEASUREMENT: ([01]{6}) \((\d+)\)", l)
        assert m and int(m.group(1), 2) == int(m.group(2))
        hits += int(m.group(2)) in (3, 18)
    assert hits >= 25  # the two marked states carry ~99.9 % of the probability


def test_out_of_place_passes_end_in_the_states_own_buffer(oracle, tmp_path):
    """QSIM_OPT_PINGPONG = 2: tile passes read one buffer and write the other.  Whatever the number of passes in a flush
    (1, 2, 3, many; fresh plans and replayed ones; with non-tile launches in between) the amplitudes match the oracle,
    qsim_device_ptr is the same pointer before and after, and in-place runs of the same circuits agree bit for bit."""
    n = 16
    cases = [circuits.random_gates(n, d, 700 + d, "all") for d in (3, 40, 90, 200, 400)]
    cases.append(circuits.random_gates(n, 300, 77, "clifford_t"))
    with Simulator(n, fuse=3, profile=True, pingpong=2, tile_bits=10) as pp, Simulator(n, fuse=3, pingpong=0, tile_bits=10) as ip:
        home = pp.device_ptr
        counts = set()
        for i, gates in enumerate(cases):
            path = circuits.write_qasm(str(tmp_path / f"p{i}.qasm"), n, gates)
            _, want, _, _ = oracle.run_qasm(path)
            c = Circuit.from_file(path)
            ip.reset(); ip.run(c)
            ref = ip.read()
            for rep in range(2):  # the second run replays the cached plan
                pp.reset(); pp.reset_stats(); pp.run(c)
                got = pp.read()
                assert pp.device_ptr == home, (i, rep)
                assert np.max(np.abs(got - want)) < TOL, (i, rep)
                assert np.array_equal(got, ref), (i, rep)
            counts.add(sum(1 for k, *_ in pp.launch_log() if k == "tile") % 2)
        assert counts == {0, 1}  # both an even and an odd number of tile passes were exercised
        # gates applied one flush at a time (a single pass per flush: always in place), then a state written by the caller
        s0 = _rand_state(n, 5)
        pp.write(s0); ip.write(s0)
        for g in cases[1][:25]:
            for sim in (pp, ip):
                sim.run(Circuit.from_gates(n, [g])); sim.flush()
        assert np.array_equal(pp.read(), ip.read()) and pp.device_ptr == home


def test_lent_spare_buffer_with_an_external_state(oracle, tmp_path):
    """qsim_set_spare_buffer on a state that lives in caller-owned memory (qsim_create_external — the sharded runs): the
    result lands in the caller's buffer, the lent one is scratch; taking it back returns the state to in-place passes."""
    import torch
    n = 15
    gates = circuits.random_gates(n, 350, 31, "all")
    path = circuits.write_qasm(str(tmp_path / "x.qasm"), n, gates)
    _, want, _, _ = oracle.run_qasm(path)
    c = Circuit.from_file(path)
    state = torch.zeros((1 << n, 2), dtype=torch.float64, device="cuda:0")
    spare = torch.full((1 << n, 2), 7.0, dtype=torch.float64, device="cuda:0")
    torch.cuda.synchronize()
    with Simulator(n, fuse=3, profile=True, pingpong=2, tile_bits=10, external_ptr=state.data_ptr()) as sim:
        sim.reset(); sim.run(c); sim.sync()
        assert float(spare.min()) == 7.0 and float(spare.max()) == 7.0  # nothing lent yet: in place
        sim.set_spare_buffer(spare.data_ptr())
        for rep in range(2):
            sim.reset(); sim.run(c); sim.sync()
            got = state.cpu().numpy().reshape(-1).view(np.complex128)
            assert np.max(np.abs(got - want)) < TOL, rep
            assert sim.device_ptr == state.data_ptr()
        assert float(spare.abs().sum()) != 7.0 * 2 * (1 << n)  # the lent buffer was written to
        sim.set_spare_buffer(None)
        spare.fill_(3.0); torch.cuda.synchronize()
        sim.reset(); sim.run(c); sim.sync()
        got = state.cpu().numpy().reshape(-1).view(np.complex128)
        assert np.max(np.abs(got - want)) < TOL and float(spare.min()) == 3.0 and float(spare.max()) == 3.0
        with pytest.raises(Exception):
            sim.set_spare_buffer(state.data_ptr())


def test_sparse_start_visits_only_the_support(oracle, tmp_path):
    """QSIM_OPT_SPARSE_START: after a reset the tile passes only visit tiles inside the state's support and treat memory
    outside it as zero without ever having written it.  Checked where that can go wrong: stale amplitudes of an earlier,
    dense run in both buffers; a circuit that leaves most qubits untouched (the zeros are only written when the state is
    read); a single-kernel gate and a caller's write in the middle of the sparse phase; agreement with plain full
    sweeps (bit-exact once the state is dense); and the launch log shows the first passes moving a fraction of the bytes."""
    n = 18
    dense = Circuit.from_gates(n, circuits.random_gates(n, 400, 5, "all"))
    few = [g for g in circuits.random_gates(7, 120, 9, "all")]           # touches qubits 0..6 only
    path = circuits.write_qasm(str(tmp_path / "few.qasm"), n, few)
    _, want_few, _, _ = oracle.run_qasm(path)
    c_few = Circuit.from_file(path)
    gates = circuits.random_gates(n, 500, 21, "all")
    path2 = circuits.write_qasm(str(tmp_path / "r.qasm"), n, gates)
    _, want, _, _ = oracle.run_qasm(path2)
    c = Circuit.from_file(path2)
    for pp in (0, 2):
        with Simulator(n, fuse=3, profile=True, pingpong=pp, tile_bits=10) as sim, \
                Simulator(n, fuse=3, pingpong=pp, tile_bits=10, sparse_start=0) as full:
            sim.run(dense); sim.sync()                                   # leaves dense garbage behind
            sim.reset(); sim.run(c_few)
            got = sim.read()
            assert np.max(np.abs(got - want_few)) < TOL and not got[128:].any()
            sim.reset(); sim.reset_stats(); sim.run(c)
            got = sim.read()
            full.run(c)
            # (not bit for bit: knowing the support, the scheduler groups the first passes differently)
            assert np.max(np.abs(got - want)) < TOL and np.max(np.abs(got - full.read())) < 1e-13
            st = sim.stats()
            tile = st["kernels"]["tile"]
            assert tile["bytes"] < 0.95 * tile["launches"] * 32 * (1 << n)  # the first passes did not sweep the register
            # a dense single-qubit kernel (its own launch at fuse 0) and a write in the middle of the sparse phase
            sim.set_option(_lib.OPT_FUSE, 3)
            sim.reset(); sim.run(Circuit.from_gates(n, gates[:40])); sim.flush()
            sim.set_option(_lib.OPT_FUSE, 0)
            sim.run(Circuit.from_gates(n, gates[40:45])); sim.flush()
            sim.set_option(_lib.OPT_FUSE, 3)
            sim.run(Circuit.from_gates(n, gates[45:])); sim.flush()
            assert np.max(np.abs(sim.read() - want)) < TOL
            s0 = _rand_state(n, 77)
            sim.reset(); sim.run(Circuit.from_gates(n, gates[:30])); sim.flush()
            sim.write(s0)
            sim.run(Circuit.from_gates(n, gates[:60]))
            full.write(s0); full.run(Circuit.from_gates(n, gates[:60]))
            assert np.array_equal(sim.read(), full.read())


def test_planning_for_a_dense_start(oracle):
    """qsim_tune_circuit_from(dense_start): the schedule of a circuit that runs on a caller-written (dense) state differs in
    its first passes from the one after a reset; planning it fills the table for THOSE geometries and changes no result."""
    n = 17
    gates = circuits.random_gates(n, 300, 88, "all")
    c = Circuit.from_gates(n, gates)
    s0 = _rand_state(n, 3)
    with Simulator(n, fuse=3, profile=True, tile_bits=10) as sim, Simulator(n, fuse=0) as ref:
        ref.write(s0); ref.run(c)
        want = ref.read()                                           # the per-gate kernels, each verified against the oracle
        lib = _lib.load()
        lib.qsim_tune_table_clear()
        rep = sim.tune(c, max_candidates=4, budget_ms=0, dense_start=True)
        assert rep["passes_tuned"] >= 1 and lib.qsim_tune_table_size() == rep["passes_tuned"]
        sim.write(s0); sim.reset_stats(); sim.run(c)
        assert np.max(np.abs(sim.read() - want)) < TOL
        dense_masks = {hm for k, _, hm, _ in sim.launch_log() if k == "tile"}
        rep2 = sim.tune(c, max_candidates=4, budget_ms=0, dense_start=True)
        assert rep2["already_known"] == rep2["tile_passes"]          # every geometry of the dense-start schedule is planned
        sim.reset(); sim.reset_stats(); sim.run(c); sim.sync()
        sparse_masks = {hm for k, _, hm, _ in sim.launch_log() if k == "tile"}
        assert sparse_masks != dense_masks                           # after a reset the first passes are chosen differently
        lib.qsim_tune_table_clear()


def test_planning_keeps_the_cheaper_of_the_two_schedules(oracle, tmp_path):
    """SchedConfig::commute: clusters that are block-diagonal in every qubit they share may overtake each other.  Both ways
    are valid schedules; qsim_tune_circuit plans a circuit under a few dozen scheduler settings (both orders among them),
    remembers the one its pass-time model likes best (bytes moved, plus 11 % of a sweep for every merged block beyond four),
    and the run after it follows that choice.  Amplitudes equal the oracle's."""
    n = 20
    lib = _lib.load()
    seen = set()
    for seed in (40, 41, 42, 59, 61, 65):  # the last three are cheaper in the round-1 order (plan figures, CPU)
        gates = circuits.random_gates(n, 500, seed, "all")
        path = circuits.write_qasm(str(tmp_path / "c.qasm"), n, gates)
        _, want, _, _ = oracle.run_qasm(path)
        c = Circuit.from_file(path)
        costs = {}
        for variant in (1, 0):
            if variant == 0:
                os.environ["QSIM_SCHED_NOCOMMUTE"] = "1"
            try:
                costs[variant] = c.plan(fuse=3, tile_bits=10, tile_low_bits=3)["algorithmic_bytes"]
            finally:
                os.environ.pop("QSIM_SCHED_NOCOMMUTE", None)
        with Simulator(n, fuse=3, profile=True, tile_bits=10, tile_max_ops=32) as sim:
            lib.qsim_tune_table_clear()
            sim.run(c)
            assert np.max(np.abs(sim.read() - want)) < TOL               # default: clusters may overtake
            sim.tune(c, max_candidates=1, budget_ms=0)
            sim.reset_stats(); sim.run(c)
            got = sim.read()
            assert np.max(np.abs(got - want)) < TOL
            moved = sim.stats()["algorithmic_bytes"]
            best = 0 if costs[0] < 0.995 * costs[1] else 1
            seen.add(best)
            # the planned run moves no more bytes than the better of the two orders, give or take what the model trades for
            # leaner passes (a few per cent)
            assert moved <= costs[best] * 1.03, (seed, costs, moved)
    lib.qsim_tune_table_clear()
    assert seen == {0, 1}  # both outcomes were exercised


def test_plan_cache_does_not_trust_its_key(oracle, tmp_path):
    """the plan cache found plans by a 64-bit FNV-1a key alone.  With QSIM_OPT_DEBUG_PLAN_KEY every queue gets
    the SAME key; two different circuits of the same length must still each get their own, correct schedule — a cached
    plan is only replayed after its gate list, options, support and scheduler overrides have been compared."""
    n, depth = 16, 300
    wants, circs = [], []
    for seed in (71, 72, 73):
        gates = circuits.random_gates(n, depth, seed, "all")
        path = circuits.write_qasm(str(tmp_path / f"k{seed}.qasm"), n, gates)
        wants.append(oracle.run_qasm(path)[1])
        circs.append(Circuit.from_file(path))
    with Simulator(n, fuse=3, debug_plan_key=12345) as sim:
        for rep in range(3):
            for c, want in zip(circs, wants):
                sim.reset(); sim.run(c)
                assert np.max(np.abs(sim.read() - want)) < TOL, rep
        st = sim.plan_cache_stats()
        assert st["plans"] == 3 and st["replays"] == 6     # every circuit planned once, replayed twice
        assert st["key_collisions"] >= 6                    # ... and each replay had to walk past the others' plans
        os.environ["QSIM_SCHED_NOCOMMUTE"] = "1"            # a scheduler override is part of a plan's identity too
        try:
            sim.reset(); sim.run(circs[0])
            assert np.max(np.abs(sim.read() - wants[0])) < TOL
            assert sim.plan_cache_stats()["replays"] == 6
        finally:
            os.environ.pop("QSIM_SCHED_NOCOMMUTE", None)


def test_support_api_and_zero_shard_semantics(oracle, tmp_path):
    """qsim_set_support / qsim_get_support / the all-zero vector: what the receiving side of a sparse exchange relies on.
    A state declared zero outside a support, with NaNs in that memory, behaves exactly like the zero-padded state; gates on
    a shard that holds nothing are dropped and it reads back as zeros."""
    import torch
    n = 15
    gates = circuits.random_gates(n, 300, 91, "all")
    path = circuits.write_qasm(str(tmp_path / "s.qasm"), n, gates)
    c = Circuit.from_file(path)
    rng = np.random.default_rng(5)
    support = sum(1 << b for b in (0, 1, 2, 3, 5, 6, 9, 10, 12))
    idx = np.arange(1 << n)
    inside = (idx & ~support) == 0
    init = np.where(inside, rng.standard_normal(1 << n) + 1j * rng.standard_normal(1 << n), 0)
    init /= np.linalg.norm(init)
    want = init.copy()
    for g in gates:
        if g[0] == "cx":
            oracle.apply_cx(want, n, g[1], g[2])
        else:
            tok = f"rz({g[1]!r})" if g[0] == "rz" else g[0]
            oracle.apply_1q(want, n, np.ascontiguousarray(gate_matrix(tok).T), g[-1])
    for pingpong in (0, 2):
        with Simulator(n, fuse=3, pingpong=pingpong) as sim:
            poisoned = np.where(inside, init, np.nan + 1j * np.nan)
            sim.write(poisoned)
            sim.set_support(support)
            assert sim.get_support() == (support, False, 0.0)
            sim.run(c)
            got = sim.read()
            assert not np.isnan(got).any()
            assert np.max(np.abs(got - want)) < TOL
            assert sim.get_support()[0] == (1 << n) - 1
            # a shard that holds nothing: gates are dropped, reads give zeros, the norm is 0
            sim.reset(holds_index0=False)
            sim.run(c)
            sim.flush()
            assert sim.get_support() == (0, True, 0.0)
            assert sim.norm2() == 0.0 and not sim.read().any()


@pytest.mark.parametrize("shards", [2, 4, 8])
def test_cluster_exchanges_leave_out_what_is_zero(oracle, tmp_path, shards):
    """Support carried through exchanges (csrc/dist.cpp roles_of): the first exchanges of a run move only the blocks that can
    be non-zero, shards that hold nothing do no work, and the shards keep visiting only their support afterwards — with
    stale amplitudes of an earlier, unrelated run in every buffer (state and exchange scratch).  Same amplitudes as the oracle."""
    from gpu_quantum_simulator_amd import Cluster
    n = 18
    stale = Circuit.from_file(circuits.random_circuit_file(str(tmp_path / "stale.qasm"), n, 500, 7, "all"))
    for seed, depth in ((301, 700), (302, 80)):
        path = circuits.random_circuit_file(str(tmp_path / f"c{seed}.qasm"), n, depth, seed + shards, "all")
        _, want, _, _ = oracle.run_qasm(path)
        c = Circuit.from_file(path)
        with Cluster(n, shards, devices=[0] * shards) as cl:
            cl.run(stale)  # leaves dense garbage in the state and scratch buffers of every shard
            cl.run(stale)
            moved0, (ex0, dense0) = cl.exchange_bytes_moved(), cl.exchange_stats()
            cl.run(c)
            got = cl.read()
            assert np.max(np.abs(got - want)) < TOL
            assert abs(cl.norm2() - 1.0) < 1e-10
            ex, dense_per_shard = cl.exchange_stats()
            ex, dense_per_shard = ex - ex0, dense_per_shard - dense0
            if depth == 700:
                assert ex >= 2
            moved = cl.exchange_bytes_moved() - moved0
            assert (moved > 0) == (ex > 0)
            assert ex == 0 or moved < dense_per_shard * shards  # something stayed home: the first exchange has shards that hold nothing
            fused, separate = cl.pack_counts()
            assert fused + separate > 0 and (depth != 700 or fused > 0)  # most re-layouts ride on the last tile pass before the exchange


@pytest.mark.parametrize("planned", [False, True])
def test_support_after_is_what_the_engine_ends_up_with(tmp_path, planned):
    """qsim_support_after predicts, without running anything, the support the engine tracks after a circuit — with the default
    schedule and with the one the planning step chose — from a reset and from a partial support."""
    n = 20
    for seed, depth in ((1, 6), (2, 14), (3, 40), (4, 200)):
        gates = [g for g in circuits.random_gates(n, depth, 900 + seed, "all") if seed == 4 or max(g[1:] if g[0] == "cx" else g[-1:]) < n - 2]
        c = Circuit.from_gates(n, gates)
        with Simulator(n, fuse=3) as sim:
            if planned:
                sim.choose_schedule(c)
            want = sim.support_after(c, 0)
            sim.run(c)
            sim.flush()
            got, pending, _ = sim.get_support()
            assert not pending and got == want, (seed, hex(got), hex(want))
            if got != (1 << n) - 1:  # some qubit is still |0>: a second circuit on top of the partial state
                c2 = Circuit.from_gates(n, circuits.random_gates(n, 10, 950 + seed, "all"))
                want2 = sim.support_after(c2, got)
                sim.run(c2)
                sim.flush()
                assert sim.get_support()[0] == want2


@pytest.mark.parametrize("shards", [2, 4])
def test_planned_cluster_lets_the_last_pass_do_every_relayout(oracle, tmp_path, shards):
    """qsim_cluster_plan goes through the steps in order and tells every exchange what its senders' engines will have written
    under the schedules it chose (qsim_support_after), so the last tile pass in front of an exchange can take the re-layout
    also where the chosen schedule leaves other qubits untouched than the planner's default one.  Same amplitudes as the
    oracle; with planning at most as many separate pack sweeps as without."""
    from gpu_quantum_simulator_amd import Cluster
    n = 18
    for seed in (11, 12, 13):
        path = circuits.random_circuit_file(str(tmp_path / f"p{seed}.qasm"), n, 600, seed, "all")
        _, want, _, _ = oracle.run_qasm(path)
        c = Circuit.from_file(path)
        counts = []
        for plan in (False, True):
            with Cluster(n, shards, devices=[0] * shards) as cl:
                if plan:
                    cl.plan(c)
                cl.run(c)
                assert np.max(np.abs(cl.read() - want)) < TOL
                counts.append(cl.pack_counts())
                cl.run(c)  # the cached plan (refined by the planning step) a second time
                assert np.max(np.abs(cl.read() - want)) < TOL
        assert counts[1][1] <= counts[0][1], counts


def _pack_src_index(n, bits):
    k = len(bits)
    d = np.arange(1 << n, dtype=np.int64)
    rest, blk = d & ((1 << (n - k)) - 1), d >> (n - k)
    keep = [b for b in range(n) if b not in bits]
    src = np.zeros_like(d)
    for i, b in enumerate(keep):
        src |= ((rest >> i) & 1) << b
    for i, b in enumerate(bits):
        src |= ((blk >> i) & 1) << b
    return src


@pytest.mark.parametrize("bits", [(5,), (3, 11), (0, 9), (4, 12, 13), (1, 2, 6), (12, 13, 14)])
@pytest.mark.parametrize("pingpong", [0, 2])
def test_flush_pack_last_pass_does_the_relayout(oracle, tmp_path, bits, pingpong):
    """qsim_flush_pack: the last tile pass of the queue writes the state straight into the packed layout of an exchange
    (k_tile PACK, a permutation of index bits applied to every store address).  Same amplitudes in the same places as
    qsim_flush followed by qsim_pack_bits — on a dense state, with out-of-place passes in front, and into a larger buffer
    with the blocks steered by to_bits / konst (the form a cluster with one allocation for all shards uses)."""
    import torch
    n = 15
    gates = circuits.random_gates(n, 400, 77, "all")
    path = circuits.write_qasm(str(tmp_path / "fp.qasm"), n, gates)
    _, want, _, _ = oracle.run_qasm(path)
    c = Circuit.from_file(path)
    src = _pack_src_index(n, bits)
    k = len(bits)
    with Simulator(n, fuse=3, pingpong=pingpong) as sim:
        out = torch.full((1 << n, 2), float("nan"), dtype=torch.float64, device="cuda")
        big = torch.full((4 << n, 2), float("nan"), dtype=torch.float64, device="cuda")
        torch.cuda.synchronize()
        for rep in range(2):  # second time: the cached plan is replayed
            sim.reset()
            sim.run(c)
            at, fused = sim.flush_pack(bits, out.data_ptr())
            sim.sync()
            assert fused and at == out.data_ptr()
            got = out.cpu().numpy().reshape(-1).view(np.complex128)
            assert np.max(np.abs(got - want[src])) < TOL, rep
        # blocks steered into a buffer four times the size: block b -> index bits to_bits, plus a constant offset
        to_bits = [n + 1 - j for j in range(k)] if k <= 2 else [n - k + j for j in range(k)]
        konst = (1 << n) if k == 1 else 0
        sim.reset()
        sim.run(c)
        at, fused = sim.flush_pack(bits, big.data_ptr(), to_bits=to_bits, konst=konst)
        sim.sync()
        assert fused
        gotb = big.cpu().numpy().reshape(-1).view(np.complex128)
        blk = 1 << (n - k)
        for b in range(1 << k):
            start = konst | sum(((b >> j) & 1) << to_bits[j] for j in range(k))
            assert np.max(np.abs(gotb[start:start + blk] - want[src][b * blk:(b + 1) * blk])) < TOL, b
        assert np.isnan(gotb).sum() == (3 << n)  # nothing else was written


def test_flush_pack_on_partial_states_and_fallbacks(oracle, tmp_path):
    """The fused re-layout only writes what its pass visits: with a partially written state it is used when that covers what
    the receivers need and replaced by the pack kernel (which writes the zeros) when it does not; queues that do not end in
    a tile pass, empty queues and tiny registers take the pack kernel too.  Always the same packed amplitudes."""
    import torch
    n = 16
    bits = (3, 14)
    src = _pack_src_index(n, bits)
    # gates on the low 11 qubits only: the state stays partial (support = bits 0..10 + padding of one tile)
    gates = [g for g in circuits.random_gates(11, 200, 5, "all")]
    path = circuits.write_qasm(str(tmp_path / "p.qasm"), n, gates)
    _, want, _, _ = oracle.run_qasm(path)
    c = Circuit.from_file(path)
    low = (1 << 11) - 1
    with Simulator(n, fuse=3) as sim:
        out = torch.full((1 << n, 2), float("nan"), dtype=torch.float64, device="cuda")
        torch.cuda.synchronize()
        sim.run(c)
        sup_needed = low
        at, fused = sim.flush_pack(bits, out.data_ptr(), needed=sup_needed)
        sim.sync()
        assert fused
        got = out.cpu().numpy().reshape(-1).view(np.complex128)
        inside = (src & ~low) == 0
        assert np.max(np.abs(got[inside] - want[src][inside])) < TOL  # where the receivers look
        # needed = everything: the pass would leave most of the buffer unwritten -> pack kernel, zeros included
        out.fill_(float("nan")); torch.cuda.synchronize()
        sim.reset(); sim.run(c)
        at, fused = sim.flush_pack(bits, out.data_ptr())
        sim.sync()
        assert not fused
        got = out.cpu().numpy().reshape(-1).view(np.complex128)
        assert np.max(np.abs(got - want[src])) < TOL
        # a queue that ends in a single-gate kernel (fuse 0), and an empty queue on a dense state
        sim.set_option(_lib.OPT_FUSE, 0)
        sim.reset(); sim.run(c)
        at, fused = sim.flush_pack(bits, out.data_ptr())
        sim.sync()
        assert not fused
        assert np.max(np.abs(out.cpu().numpy().reshape(-1).view(np.complex128) - want[src])) < TOL
        at, fused = sim.flush_pack(bits, out.data_ptr())
        sim.sync()
        assert not fused
        assert np.max(np.abs(out.cpu().numpy().reshape(-1).view(np.complex128) - want[src])) < TOL
    # tiles smaller than the production shape: no fused variant, same result
    n2 = 10
    g2 = circuits.random_gates(n2, 150, 9, "all")
    p2 = circuits.write_qasm(str(tmp_path / "s.qasm"), n2, g2)
    _, w2, _, _ = oracle.run_qasm(p2)
    with Simulator(n2, fuse=3) as sim:
        out = torch.zeros((1 << n2, 2), dtype=torch.float64, device="cuda")
        torch.cuda.synchronize()
        sim.run(Circuit.from_file(p2))
        at, fused = sim.flush_pack((2, 7), out.data_ptr())
        sim.sync()
        assert not fused
        assert np.max(np.abs(out.cpu().numpy().reshape(-1).view(np.complex128) - w2[_pack_src_index(n2, (2, 7))])) < TOL


@pytest.mark.parametrize("bits", [(1, 5, 9, 13), (0, 3, 4, 11, 14)])
def test_flush_pack_of_wide_exchanges_takes_the_pack_kernel(oracle, tmp_path, bits):
    """Exchanges of 4 and 5 qubits — groups of 16 and 32 shards, what the planner emits for P = 16 / 32 — go through the same call
    as the narrow ones (exchange_rccl, rank_exchange: qsim_flush_pack into the scratch buffer).  A tile pass re-lays out at most
    three bits, so these take flush + the pack kernel inside that call, with the sparse roles kept: same amplitudes in the same
    places as the index permutation says, blocks in `skip_blocks` untouched, and the steered forms (to_bits / konst) refused
    rather than mis-executed (the call used to fail with "4 bits unsupported" on every 16-rank run)."""
    import torch
    n = 15
    gates = circuits.random_gates(n, 300, 177, "all")
    path = circuits.write_qasm(str(tmp_path / "w.qasm"), n, gates)
    _, want, _, _ = oracle.run_qasm(path)
    c = Circuit.from_file(path)
    src = _pack_src_index(n, bits)
    k = len(bits)
    blk = 1 << (n - k)
    with Simulator(n, fuse=3) as sim:
        out = torch.full((1 << n, 2), float("nan"), dtype=torch.float64, device="cuda")
        torch.cuda.synchronize()
        sim.run(c)
        at, fused = sim.flush_pack(bits, out.data_ptr())
        sim.sync()
        assert not fused and at == out.data_ptr()
        got = out.cpu().numpy().reshape(-1).view(np.complex128)
        assert np.max(np.abs(got - want[src])) < TOL
        skip = 0b1010010010 & ((1 << (1 << k)) - 1)
        out.fill_(float("nan")); torch.cuda.synchronize()
        sim.reset(); sim.run(c)
        at, fused = sim.flush_pack(bits, out.data_ptr(), skip_blocks=skip)
        sim.sync()
        got = out.cpu().numpy().reshape(-1).view(np.complex128)
        for b in range(1 << k):
            piece = got[b * blk:(b + 1) * blk]
            if skip >> b & 1:
                assert np.isnan(piece).all(), b
            else:
                assert np.max(np.abs(piece - want[src][b * blk:(b + 1) * blk])) < TOL, b
        with pytest.raises(_lib.QsimError):
            sim.flush_pack(bits, out.data_ptr(), to_bits=[n - k + j for j in range(k)], konst=1 << n)


def test_cluster_runs_one_circuit_per_reset(oracle, tmp_path):
    """qsim_cluster_run_circuit is compute_state_vector's contract (quantum_simulator.c:115-254): one circuit, from |0...0>.  The plan
    relies on it (free first placement, exchanges that leave out what is still zero), so a second circuit without a reset in
    between is refused — it used to run and, after a circuit whose qubit map ended as the identity, silently dropped amplitudes
    (ADVICE r03).  With the reset both runs give the oracle's amplitudes."""
    from gpu_quantum_simulator_amd import Cluster
    n = 14
    lib = _lib.load()
    first = circuits.write_qasm(str(tmp_path / "a.qasm"), n, [("h", 0), ("cx", 0, 1), ("t", 1)])  # no exchange: the map stays the identity
    second = circuits.write_qasm(str(tmp_path / "b.qasm"), n, circuits.random_gates(n, 200, 31, "all"))
    ca, cb = Circuit.from_file(first), Circuit.from_file(second)
    with Cluster(n, 4, devices=[0] * 4) as cl:
        assert lib.qsim_cluster_run_circuit(cl._h, ca._h) != 0  # a fresh cluster has not been reset either
        assert b"reset" in lib.qsim_cluster_error()
        cl.run(ca)  # reset + run
        _, want, _, _ = oracle.run_qasm(first)
        assert np.max(np.abs(cl.read() - want)) < TOL
        assert lib.qsim_cluster_run_circuit(cl._h, cb._h) != 0
        assert b"reset" in lib.qsim_cluster_error()
        assert np.max(np.abs(cl.read() - want)) < TOL  # the refused call touched nothing
        cl.run(cb)
        _, want, _, _ = oracle.run_qasm(second)
        assert np.max(np.abs(cl.read() - want)) < TOL
