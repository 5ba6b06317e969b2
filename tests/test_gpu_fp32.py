"""fp32 state (qsim_create_f32): the precision of the reference's CUDA variants (cuFloatComplex, naive.cu:38).

NOT the parity configuration (that is fp64 within 1e-10, tests/test_gpu_parity.py).  The same kernels are instantiated for
float amplitudes; these tests hold them to the fp64 oracle within fp32 rounding.  Tolerance: each gate adds at most a few
ulp(fp32) = 6e-8 relative per amplitude; over a few hundred gates on unit-norm states 2e-5 abs is a loose bound and far
below any indexing error (which shows up at the 1e-2..1 level).
"""
import os

import numpy as np
import pytest

from gpu_quantum_simulator_amd import Simulator, circuits, run_qasm
from helpers import random_unitary

pytestmark = pytest.mark.gpu
TOL32 = 2e-5


def _rand_state(n, seed):
    rng = np.random.default_rng(seed)
    s = rng.standard_normal(1 << n) + 1j * rng.standard_normal(1 << n)
    return (s / np.linalg.norm(s)).astype(np.complex128)


def test_precision_flag_and_io_round_trip():
    n = 12
    s = _rand_state(n, 3)
    with Simulator(n, precision=32) as sim:
        assert sim.precision == 32
        from gpu_quantum_simulator_amd import _lib
        assert _lib.load().qsim_precision_bits(sim._h) == 32
        zero = sim.read()
        assert zero[0] == 1.0 and not zero[1:].any()
        sim.write(s)
        got = sim.read()
        assert np.array_equal(got, s.astype(np.complex64).astype(np.complex128))  # one rounding, nothing else
        part = sim.read(100, 37)
        assert np.array_equal(part, got[100:137])
        assert abs(sim.norm2() - 1.0) < 1e-6
    with Simulator(n) as sim:
        from gpu_quantum_simulator_amd import _lib
        assert _lib.load().qsim_precision_bits(sim._h) == 64
    with pytest.raises(ValueError):
        Simulator(4, precision=16)


@pytest.mark.parametrize("fuse", [0, 1, 2, 3])
@pytest.mark.parametrize("name", ["entanglement", "grover_3_18", "rand_n10_all", "rand_n12_all",
                                  "rand_n12_clifford_t_physical"])
def test_golden_fixtures_fp32(golden_dir, name, fuse):
    want = np.load(os.path.join(golden_dir, name + ".npy"), allow_pickle=False).view(np.complex128).reshape(-1)
    got = run_qasm(os.path.join(golden_dir, name + ".qasm"), fuse=fuse, precision=32)
    assert np.max(np.abs(got - want)) < TOL32


def test_every_target_bit_and_pair_fp32(oracle):
    """Per-gate kernels (fuse=0): dense/diagonal 1q on every bit, cx on every ordered pair, against the oracle."""
    n = 13
    rng = np.random.default_rng(5)
    with Simulator(n, fuse=0, precision=32) as sim:
        for q in range(n):
            for U in (random_unitary(2, rng), np.diag(np.exp(1j * rng.uniform(0, 6.28, 2)))):
                s = _rand_state(n, 300 + q)
                sim.write(s)
                sim.apply_1q(U, q)
                want = s.copy()
                oracle.apply_1q(want, n, U.T, q)
                assert np.max(np.abs(sim.read() - want)) < TOL32, q
        for c in range(n):
            for t in range(n):
                if c == t:
                    continue
                s = _rand_state(n, 400 + c * n + t)
                sim.write(s)
                sim.apply_cx(c, t)
                want = s.copy()
                oracle.apply_cx(want, n, c, t)
                assert np.max(np.abs(sim.read() - want)) < TOL32, (c, t)


@pytest.mark.parametrize("n,depth,seed,vocab,opts", [
    (14, 400, 31, "all", {}),
    (16, 500, 32, "clifford_t", {}),
    (18, 400, 33, "all", {"tile_bits": 13, "tile_low_bits": 6}),
    (20, 300, 34, "all", {"tile_bits": 11, "tile_low_bits": 5, "tile_max_ops": 3}),
    (19, 300, 35, "all", {"tile_bits": 9, "tile_low_bits": 2}),
    (20, 300, 36, "all", {"grid_cap": 64}),
    (22, 400, 37, "all", {}),
])
def test_random_circuits_fp32(oracle, tmp_path, n, depth, seed, vocab, opts):
    path = circuits.random_circuit_file(str(tmp_path / "c.qasm"), n, depth, seed, vocab)
    _, want, _, _ = oracle.run_qasm(path)
    got = run_qasm(path, fuse=3, precision=32, **opts)
    assert np.max(np.abs(got - want)) < TOL32


@pytest.mark.parametrize("order_seed", [1, 2])
def test_tile_bit_order_fp32(oracle, tmp_path, order_seed):
    """Shuffled tile-bit orders (QSIM_OPT_DEBUG_TILE_ORDER) through the fp32 instantiation of the tile kernel."""
    n = 17
    path = circuits.random_circuit_file(str(tmp_path / "c.qasm"), n, 500, 91, "all")
    _, want, _, _ = oracle.run_qasm(path)
    got = run_qasm(path, fuse=3, precision=32, debug_tile_order=order_seed)
    assert np.max(np.abs(got - want)) < TOL32


def test_randomised_geometry_sweep_fp32(oracle, tmp_path):
    rng = np.random.default_rng(77)
    for case in range(30):
        n = int(rng.integers(2, 20))
        depth = int(rng.integers(20, 300))
        tile_bits = int(rng.integers(8, 14))
        tile_low = int(rng.integers(max(2, tile_bits - 10), min(6, tile_bits - 2) + 1))
        opts = {"tile_bits": tile_bits, "tile_low_bits": tile_low, "tile_max_ops": int(rng.integers(1, 40)),
                "tile_pad_from": int(rng.integers(-1, 20))}
        fuse = int(rng.choice([0, 1, 2, 3, 3, 3]))
        path = circuits.random_circuit_file(str(tmp_path / f"g{case}.qasm"), n, depth, 7000 + case, "all")
        _, want, _, _ = oracle.run_qasm(path)
        got = run_qasm(path, fuse=fuse, precision=32, **opts)
        err = float(np.max(np.abs(got - want)))
        assert err < TOL32, (case, n, depth, fuse, opts, err)


def test_dense_2q_and_sparse_blocks_fp32():
    from helpers import np_apply_2q
    n = 14
    rng = np.random.default_rng(9)
    with Simulator(n, precision=32) as sim:
        for hi, lo in [(13, 0), (5, 2), (12, 7), (1, 0), (9, 3)]:
            U = random_unitary(4, rng)
            s = _rand_state(n, 500 + hi)
            sim.write(s)
            sim.apply_2q(U, hi, lo)
            want = np_apply_2q(s.copy(), n, U, hi, lo)
            assert np.max(np.abs(sim.read() - want)) < TOL32, (hi, lo)


def test_sampling_and_pack_fp32():
    import torch
    n = 14
    s = _rand_state(n, 60)
    with Simulator(n, precision=32) as sim:
        sim.write(s)
        s32 = s.astype(np.complex64)
        r = np.array([0.0, 0.1, 0.5, 0.999, 0.25, 1.0])
        idx = sim.sample(r)
        # sums run in fp64 over the widened amplitudes: identical to an fp64 state holding the rounded values
        with Simulator(n) as wide:
            wide.write(s32.astype(np.complex128))
            assert np.array_equal(idx, wide.sample(r))
        dst = torch.zeros((1 << n, 2), dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()  # the fill runs on torch's stream, the pack on the engine's: order them
        sim.pack_bits((3, 11), dst.data_ptr())
        sim.sync()
        got = dst.cpu().numpy().reshape(-1).view(np.complex64)
    d = np.arange(1 << n, dtype=np.int64)
    rest, blk = d & ((1 << (n - 2)) - 1), d >> (n - 2)
    keep = [b for b in range(n) if b not in (3, 11)]
    src = np.zeros_like(d)
    for i, b in enumerate(keep):
        src |= ((rest >> i) & 1) << b
    for i, b in enumerate((3, 11)):
        src |= ((blk >> i) & 1) << b
    assert np.array_equal(got, s32[src])


def test_round_trip_at_n31_fp32():
    """2^31 fp32 amplitudes (16 GiB, as many bytes as the n=30 fp64 headline): circuit then its inverse returns |0...0>."""
    from gpu_quantum_simulator_amd import Circuit
    from test_gpu_parity import _inverse
    n = 31
    gates = circuits.random_gates(n, 150, 4242, "all")
    fwd = Circuit.from_gates(n, gates)
    bwd = Circuit.from_gates(n, _inverse(gates))
    with Simulator(n, precision=32) as sim:
        sim.run(fwd)
        assert abs(sim.norm2() - 1.0) < 1e-4
        assert abs(sim.read(0, 4)[0]) < 0.999
        sim.run(bwd)
        head = sim.read(0, 1 << 12)
        tail = sim.read((1 << n) - 4096, 4096)
        assert abs(sim.norm2() - 1.0) < 1e-4
    assert abs(head[0] - 1.0) < 1e-4 and np.max(np.abs(head[1:])) < 1e-4 and np.max(np.abs(tail)) < 1e-4
