"""Python mirror of the C host: thin objects over the C ABI handles, nothing computed in Python.

`Circuit` = qsim_circuit (the tokenizer of quantum_simulator.c:115-254 and the gate table :184-211),
`Simulator` = qsim_state (state vector in HBM; execute_single_qubit_gate :81-92 -> apply_1q,
execute_cnot :94-106 -> apply_cx, kernel_gate_4 quantum_simulator_4x4.cu:109-146 -> apply_2q).
"""
from __future__ import annotations

import ctypes
from ctypes import byref, c_double, c_int, c_void_p
from typing import Iterable, Optional, Sequence

import numpy as np

from . import _lib
from ._lib import QsimStats, check


def _dp(a: np.ndarray):
    return a.ctypes.data_as(ctypes.POINTER(c_double))


def _mat(U, dim: int) -> np.ndarray:
    m = np.ascontiguousarray(np.asarray(U, dtype=np.complex128).reshape(dim * dim))
    return m.view(np.float64)


def gate_matrix(token: str) -> Optional[np.ndarray]:
    """2x2 for a gate token of the reference's vocabulary (None for cx / unknown)."""
    u = np.zeros(8)
    kind = _lib.load().qsim_gate_matrix(token.encode(), _dp(u))
    return u.view(np.complex128).reshape(2, 2).copy() if kind == _lib.GATE_U1 else None


class Circuit:
    def __init__(self, handle: c_void_p):
        self._h = handle

    @classmethod
    def empty(cls, num_q: int) -> "Circuit":
        h = c_void_p()
        check(_lib.load().qsim_circuit_create(num_q, byref(h)))
        return cls(h)

    @classmethod
    def from_file(cls, path: str) -> "Circuit":
        h = c_void_p()
        rc = _lib.load().qsim_circuit_parse_file(path.encode(), byref(h))
        if rc:
            raise _lib.QsimError(rc, (_lib.load().qsim_circuit_error() or b"").decode())
        return cls(h)

    @classmethod
    def from_text(cls, text: str) -> "Circuit":
        h = c_void_p()
        raw = text.encode()
        rc = _lib.load().qsim_circuit_parse_text(raw, len(raw), byref(h))
        if rc:
            raise _lib.QsimError(rc, (_lib.load().qsim_circuit_error() or b"").decode())
        return cls(h)

    @classmethod
    def from_gates(cls, num_q: int, gates: Iterable[Sequence]) -> "Circuit":
        """gates in the tuple form of circuits.random_gates."""
        c = cls.empty(num_q)
        cache = {}
        for g in gates:
            if g[0] == "cx":
                c.append_cx(g[1], g[2])
            elif g[0] == "rz":
                c.append_1q(gate_matrix(f"rz({g[1]!r})"), g[2])
            else:
                if g[0] not in cache:
                    cache[g[0]] = gate_matrix(g[0])
                c.append_1q(cache[g[0]], g[1])
        return c

    def append_1q(self, U, target: int) -> None:
        check(_lib.load().qsim_circuit_append_1q(self._h, _dp(_mat(U, 2)), target))

    def append_cx(self, control: int, target: int) -> None:
        check(_lib.load().qsim_circuit_append_cx(self._h, control, target))

    def append_2q(self, U, q_hi: int, q_lo: int) -> None:
        check(_lib.load().qsim_circuit_append_2q(self._h, _dp(_mat(U, 4)), q_hi, q_lo))

    @property
    def num_qubits(self) -> int:
        return _lib.load().qsim_circuit_num_qubits(self._h)

    def __len__(self) -> int:
        return int(_lib.load().qsim_circuit_num_gates(self._h))

    def gate(self, i: int):
        kind, q0, q1 = c_int(), c_int(), c_int()
        u = np.zeros(32)
        check(_lib.load().qsim_circuit_gate(self._h, i, byref(kind), byref(q0), byref(q1), _dp(u)))
        if kind.value == _lib.GATE_U1:
            return ("u1", q0.value, u[:8].view(np.complex128).reshape(2, 2).copy())
        if kind.value == _lib.GATE_CX:
            return ("cx", q0.value, q1.value)
        return ("u2", q0.value, q1.value, u.view(np.complex128).reshape(4, 4).copy())

    def plan(self, fuse: int = 3, tile_bits: int = 12, tile_low_bits: int = 3, initial_support: int = 0) -> dict:
        """Launches and algorithmic bytes of the schedule (host only).  initial_support: index bits that may be 1 in the state
        the circuit finds (0: fresh from a reset; all ones: dense)."""
        st = QsimStats()
        check(_lib.load().qsim_plan_circuit_from(self._h, fuse, tile_bits, tile_low_bits, initial_support, byref(st)))
        return st.as_dict()

    def passes(self, fuse: int = 3, tile_bits: int = 12, tile_low_bits: int = 3, initial_support: int = 0) -> list:
        """qsim_plan_passes: the schedule pass by pass (host only) — kernel class, blocks, the index bits inside the tile, the
        fraction of the register visited, algorithmic bytes and the bytes-equivalent of the pass-time model."""
        from ctypes import c_int
        cap = 4096
        buf = (_lib.QsimPassInfo * cap)()
        n = c_int()
        check(_lib.load().qsim_plan_passes(self._h, fuse, tile_bits, tile_low_bits, initial_support, buf, cap, byref(n)))
        return [{"kernel": _lib.K_NAMES[buf[i].kernel_class], "blocks": int(buf[i].blocks), "tile_mask": int(buf[i].tile_mask),
                 "visited": float(buf[i].visited), "bytes": float(buf[i].bytes), "cost_bytes": float(buf[i].cost_bytes)}
                for i in range(min(n.value, cap))]

    def schedule(self, fuse: int = 3, tile_bits: int = 12, tile_low_bits: int = 3, tile_max_ops: int = 32) -> list:
        """Fused blocks in launch order: (pass, kernel_class, kind, qubits, matrix|None, gates_folded); kind is
        "u1" / "cx" / "u2" ... "u8", qubits most significant first (cx: control, target)."""
        out = []

        def cb(_user, pass_i, kclass, kind, qubits, nq, U, folded):
            qs = tuple(qubits[i] for i in range(nq))
            m = None
            if kind != _lib.GATE_CX:
                d = 1 << nq
                m = np.ctypeslib.as_array(U, shape=(2 * d * d,)).copy().view(np.complex128).reshape(d, d)
            out.append((pass_i, _lib.K_NAMES[kclass], {1: "u1", 2: "cx", 3: "u2", 4: "u3", 5: "u4", 6: "u5", 7: "u6", 8: "u7", 9: "u8"}[kind], qs, m, folded))

        check(_lib.load().qsim_schedule_circuit(self._h, fuse, tile_bits, tile_low_bits, tile_max_ops,
                                                _lib.SCHED_CB(cb), None))
        return out

    def close(self) -> None:
        if self._h:
            _lib.load().qsim_circuit_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Simulator:
    """One state vector resident on one GPU."""

    def __init__(self, num_q: int, device: int = 0, *, fuse: Optional[int] = None, profile: bool = False,
                 external_ptr: Optional[int] = None, precision: int = 64, **options):
        self._h = c_void_p()
        lib = _lib.load()
        if precision not in (32, 64):
            raise ValueError("precision must be 64 (fp64 complex, the parity configuration) or 32")
        self.precision = precision
        if precision == 32:
            if external_ptr is not None:
                raise ValueError("external buffers are fp64 only")
            check(lib.qsim_create_f32(byref(self._h), num_q, device))
        elif external_ptr is None:
            check(lib.qsim_create(byref(self._h), num_q, device))
        else:
            check(lib.qsim_create_external(byref(self._h), num_q, device, c_void_p(external_ptr)))
        self.num_qubits = num_q
        if fuse is not None:
            self.set_option(_lib.OPT_FUSE, fuse)
        if profile:
            self.set_option(_lib.OPT_PROFILE, int(profile))  # True / 1: HIP events per launch; 2: block forms as well (launch_log_blocks)
        names = {"tile_bits": _lib.OPT_TILE_BITS, "tile_low_bits": _lib.OPT_TILE_LOW_BITS,
                 "max_pending": _lib.OPT_MAX_PENDING, "tile_max_ops": _lib.OPT_TILE_MAX_OPS,
                 "grid_cap": _lib.OPT_GRID_CAP, "tile_threads": _lib.OPT_TILE_THREADS,
                 "tile_pad_from": _lib.OPT_TILE_PAD_FROM, "debug_skip_ops": _lib.OPT_DEBUG_SKIP_OPS,
                 "debug_skip_mem": _lib.OPT_DEBUG_SKIP_MEM, "debug_tile_order": _lib.OPT_DEBUG_TILE_ORDER, "plan_cache": _lib.OPT_PLAN_CACHE,
                 "pingpong": _lib.OPT_PINGPONG, "sparse_start": _lib.OPT_SPARSE_START, "debug_plan_key": _lib.OPT_DEBUG_PLAN_KEY}
        # tile_low_bits first when shrinking, tile_bits first when growing: keep every intermediate valid
        for key in sorted(options, key=lambda k: k != "tile_low_bits"):
            self.set_option(names[key], options[key])

    # -- options / lifecycle
    def set_option(self, opt: int, value: int) -> None:
        check(_lib.load().qsim_set_option(self._h, opt, value))

    def reset(self, holds_index0: bool = True) -> None:
        check(_lib.load().qsim_reset_shard(self._h, 1 if holds_index0 else 0))

    def close(self) -> None:
        if self._h:
            _lib.load().qsim_destroy(self._h)
            self._h = c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- gates
    def apply_1q(self, U, target: int) -> None:
        check(_lib.load().qsim_apply_1q(self._h, _dp(_mat(U, 2)), target))

    def apply_cx(self, control: int, target: int) -> None:
        check(_lib.load().qsim_apply_cx(self._h, control, target))

    def apply_2q(self, U, q_hi: int, q_lo: int) -> None:
        check(_lib.load().qsim_apply_2q(self._h, _dp(_mat(U, 4)), q_hi, q_lo))

    def run(self, circuit: Circuit, first: int = 0, count: int = -1) -> None:
        check(_lib.load().qsim_run_circuit(self._h, circuit._h, first, count))

    def flush(self) -> None:
        check(_lib.load().qsim_flush(self._h))

    def plan_cache_stats(self) -> dict:
        """qsim_plan_cache_stats: plans held, replays, key matches rejected by the identity check."""
        from ctypes import c_uint64
        a, b, c = c_uint64(), c_uint64(), c_uint64()
        check(_lib.load().qsim_plan_cache_stats(self._h, byref(a), byref(b), byref(c)))
        return {"plans": int(a.value), "replays": int(b.value), "key_collisions": int(c.value)}

    def choose_schedule(self, circuit: Circuit) -> None:
        """qsim_choose_schedule: the schedule choice of the planning step alone (no timing)."""
        check(_lib.load().qsim_choose_schedule(self._h, circuit._h))

    def tune(self, circuit: Circuit, max_candidates: int = 32, budget_ms: float = 6000.0, dense_start: bool = False) -> dict:
        """qsim_tune_circuit(_from): times every pass of the circuit's schedule under candidate orders of its tile bits and
        keeps the fastest per geometry in the library's process-wide table (planning; leaves the state reset).
        dense_start: the circuit will run on a state that is not fresh from a reset (its first passes are scheduled differently)."""
        rep = _lib.QsimTuneReport()
        check(_lib.load().qsim_tune_circuit_from(self._h, circuit._h, max_candidates, budget_ms, byref(rep), 1 if dense_start else 0))
        return rep.as_dict()

    def flush_pack(self, bits: Sequence[int], out_ptr: Optional[int] = None, to_bits: Optional[Sequence[int]] = None, konst: int = 0,
                   needed: int = (1 << 64) - 1, skip_blocks: int = 0):
        """qsim_flush_pack: everything queued is launched and the state ends re-laid-out in `out_ptr` (default: the spare
        buffer); returns (buffer holding the packed state, whether the last tile pass did the re-layout)."""
        k = len(bits)
        arr = (c_int * k)(*bits)
        to = (c_int * k)(*to_bits) if to_bits is not None else None
        at, fused = c_void_p(), c_int()
        check(_lib.load().qsim_flush_pack(self._h, arr, k, to, konst, c_void_p(out_ptr or 0), needed, skip_blocks, byref(at), byref(fused)))
        return int(at.value or 0), bool(fused.value)

    def pack_bits_to(self, bits: Sequence[int], dst_ptrs: Sequence[int]) -> None:
        """qsim_pack_bits_to: block b of the packed layout goes to dst_ptrs[b]."""
        arr = (c_int * len(bits))(*bits)
        ptrs = (c_void_p * len(dst_ptrs))(*dst_ptrs)
        check(_lib.load().qsim_pack_bits_to(self._h, arr, len(bits), ptrs))

    def set_support(self, support: int) -> None:
        """qsim_set_support: amplitudes whose index has a bit outside `support` are zero by definition from now on."""
        check(_lib.load().qsim_set_support(self._h, support))

    def support_after(self, circuit: Circuit, support: int = 0) -> int:
        """qsim_support_after: where the state can be non-zero after `circuit` ran on a state with this support, as the engine
        will track it (scheduled as a flush would schedule it; nothing runs)."""
        from ctypes import c_uint64
        out = c_uint64()
        check(_lib.load().qsim_support_after(self._h, circuit._h, support & 0xFFFFFFFFFFFFFFFF, byref(out)))
        return int(out.value)

    def get_support(self):
        """(support mask, pending basis state?, its amplitude) — qsim_get_support."""
        from ctypes import c_uint64
        m, k, a = c_uint64(), c_int(), c_double()
        check(_lib.load().qsim_get_support(self._h, byref(m), byref(k), byref(a)))
        return int(m.value), bool(k.value), float(a.value)

    def set_spare_buffer(self, ptr: Optional[int]):
        """qsim_set_spare_buffer: lends the state a second device buffer for out-of-place tile passes (None takes it back)."""
        check(_lib.load().qsim_set_spare_buffer(self._h, c_void_p(ptr or 0)))

    def swap_buffer(self, ptr: int) -> int:
        """qsim_swap_buffer: the state takes `ptr` as its amplitude buffer; returns the buffer it held before."""
        p = c_void_p(ptr)
        check(_lib.load().qsim_swap_buffer(self._h, byref(p)))
        return int(p.value)

    def block_prob_masked(self, hi_mask: int, lo_mask: int) -> np.ndarray:
        out = np.zeros(1 << bin(hi_mask).count("1"), dtype=np.float64)
        check(_lib.load().qsim_block_prob_masked(self._h, hi_mask, lo_mask, _dp(out)))
        return out

    def gather_masked(self, base: int, lo_mask: int) -> np.ndarray:
        out = np.zeros(2 << bin(lo_mask).count("1"), dtype=np.float64)
        check(_lib.load().qsim_gather_masked(self._h, base, lo_mask, _dp(out)))
        return out.view(np.complex128)

    def pack_bits(self, bits: Sequence[int], dst_ptr: int) -> None:
        """Shard re-layout ahead of a global<->local qubit exchange (qsim_pack_bits)."""
        arr = (c_int * len(bits))(*bits)
        check(_lib.load().qsim_pack_bits(self._h, arr, len(bits), c_void_p(dst_ptr)))

    def scale(self, z: complex) -> None:
        check(_lib.load().qsim_scale(self._h, float(z.real), float(z.imag)))

    def sync(self) -> None:
        check(_lib.load().qsim_sync(self._h))

    # -- amplitudes
    def read(self, first: int = 0, count: Optional[int] = None) -> np.ndarray:
        if count is None:
            count = (1 << self.num_qubits) - first
        out = np.empty(2 * count, dtype=np.float64)
        check(_lib.load().qsim_read(self._h, first, count, _dp(out)))
        return out.view(np.complex128)

    def write(self, amps: np.ndarray, first: int = 0) -> None:
        a = np.ascontiguousarray(amps, dtype=np.complex128)
        check(_lib.load().qsim_write(self._h, first, a.size, _dp(a.view(np.float64))))

    def sample(self, randoms) -> np.ndarray:
        """Basis index measurement() (quantum_simulator.c:270-283) returns for each random number in [0, 1]."""
        from ctypes import c_uint64
        r = np.ascontiguousarray(randoms, dtype=np.float64)
        out = np.zeros(r.size, dtype=np.uint64)
        check(_lib.load().qsim_sample(self._h, _dp(r), r.size, out.ctypes.data_as(ctypes.POINTER(c_uint64))))
        return out

    def norm2(self) -> float:
        v = c_double()
        check(_lib.load().qsim_norm2(self._h, byref(v)))
        return v.value

    @property
    def device_ptr(self) -> int:
        return int(_lib.load().qsim_device_ptr(self._h) or 0)

    @property
    def stream(self) -> int:
        return int(_lib.load().qsim_stream(self._h) or 0)

    # -- stats
    def stats(self) -> dict:
        st = QsimStats()
        check(_lib.load().qsim_get_stats(self._h, byref(st)))
        return st.as_dict()

    def launch_log(self) -> list:
        """[(kernel, n_ops, high_mask, ms)] per launch since reset_stats (profile mode)."""
        from ctypes import c_uint64
        lib = _lib.load()
        n = lib.qsim_launch_log(self._h, -1, None, None, None, None)
        out = []
        for i in range(max(n, 0)):
            k, o, hm, ms = c_int(), c_int(), c_uint64(), c_double()
            lib.qsim_launch_log(self._h, i, byref(k), byref(o), byref(hm), byref(ms))
            out.append((_lib.K_NAMES[k.value], o.value, hm.value, ms.value))
        return out

    def launch_log_orders(self) -> list:
        """Per launch since reset_stats: the tile pass's high tile bits in tile-local order ([] for other kernels)."""
        lib = _lib.load()
        n = lib.qsim_launch_log(self._h, -1, None, None, None, None)
        out = []
        for i in range(max(n, 0)):
            order, cnt = (c_int * 10)(), c_int()
            check(lib.qsim_launch_log_order(self._h, i, order, byref(cnt)))
            out.append([order[j] for j in range(cnt.value)])
        return out

    def launch_log_blocks(self) -> list:
        """Per launch since reset_stats: [(entries_per_row, tile_qubits, mostly_identity, selector_qubits)] of a tile pass's blocks
        ([] for other kernels) — qsim_launch_log_blocks, the data behind the pass-time model."""
        from ctypes import c_ubyte
        lib = _lib.load()
        n = lib.qsim_launch_log(self._h, -1, None, None, None, None)
        out = []
        for i in range(max(n, 0)):
            codes, cnt = (c_ubyte * 64)(), c_int()
            check(lib.qsim_launch_log_blocks(self._h, i, codes, 64, byref(cnt)))
            out.append([(1 << (codes[j] & 3), (codes[j] >> 2) & 7, bool(codes[j] & 32), codes[j] >> 6) for j in range(min(cnt.value, 64))])
        return out

    def reset_stats(self) -> None:
        check(_lib.load().qsim_reset_stats(self._h))


class Cluster:
    """qsim_cluster: P shards driven by this one process (the C host's multi-GPU path; repeat a device for virtual shards)."""

    def __init__(self, num_q: int, num_shards: int, devices: Optional[Sequence[int]] = None, **options):
        self._h = c_void_p()
        lib = _lib.load()
        arr = (c_int * num_shards)(*devices) if devices is not None else None
        rc = lib.qsim_cluster_create(byref(self._h), num_q, num_shards, arr)
        if rc:
            raise _lib.QsimError(rc, (lib.qsim_cluster_error() or b"").decode())
        self.num_qubits, self.num_shards = num_q, num_shards
        names = {"fuse": _lib.OPT_FUSE, "tile_bits": _lib.OPT_TILE_BITS, "tile_low_bits": _lib.OPT_TILE_LOW_BITS,
                 "tile_max_ops": _lib.OPT_TILE_MAX_OPS, "profile": _lib.OPT_PROFILE, "pingpong": _lib.OPT_PINGPONG}
        for key in sorted(options, key=lambda k: k != "tile_low_bits"):
            self._check(lib.qsim_cluster_set_option(self._h, names[key], int(options[key])))

    def _check(self, rc):
        if rc:
            raise _lib.QsimError(rc, (_lib.load().qsim_cluster_error() or b"").decode())

    def run(self, circuit: Circuit) -> None:
        lib = _lib.load()
        self._check(lib.qsim_cluster_reset(self._h))
        self._check(lib.qsim_cluster_run_circuit(self._h, circuit._h))
        self._check(lib.qsim_cluster_sync(self._h))

    def plan(self, circuit: Circuit, max_candidates: int = 1, budget_ms: float = 0.0) -> None:
        """qsim_cluster_plan: schedule choice (and, with max_candidates > 1, measured tile-bit orders) for every shard's local steps."""
        self._check(_lib.load().qsim_cluster_plan(self._h, circuit._h, max_candidates, budget_ms))

    def read(self, first: int = 0, count: Optional[int] = None) -> np.ndarray:
        if count is None:
            count = (1 << self.num_qubits) - first
        out = np.empty(2 * count, dtype=np.float64)
        self._check(_lib.load().qsim_cluster_read(self._h, first, count, _dp(out)))
        return out.view(np.complex128)

    def norm2(self) -> float:
        v = c_double()
        self._check(_lib.load().qsim_cluster_norm2(self._h, byref(v)))
        return v.value

    def sample(self, randoms) -> np.ndarray:
        """measurement() (quantum_simulator.c:270-283) on the sharded state, indices in logical order."""
        from ctypes import c_uint64
        r = np.ascontiguousarray(randoms, dtype=np.float64)
        out = np.zeros(r.size, dtype=np.uint64)
        self._check(_lib.load().qsim_cluster_sample(self._h, _dp(r), r.size, out.ctypes.data_as(ctypes.POINTER(c_uint64))))
        return out

    def exchange_stats(self):
        from ctypes import c_uint64
        n, b = c_uint64(), c_double()
        _lib.load().qsim_cluster_exchange_stats(self._h, byref(n), byref(b))
        return int(n.value), float(b.value)

    def pack_counts(self):
        """(re-layouts done by the last tile pass in front of an exchange, by the separate pack kernel) per shard and exchange."""
        from ctypes import c_uint64
        a, b = c_uint64(), c_uint64()
        _lib.load().qsim_cluster_pack_counts(self._h, byref(a), byref(b))
        return int(a.value), int(b.value)

    def exchange_bytes_moved(self) -> float:
        """Bytes all shards together really sent (blocks of / for shards that hold nothing stay home)."""
        b = c_double()
        _lib.load().qsim_cluster_exchange_bytes_moved(self._h, byref(b))
        return float(b.value)

    @property
    def exchange_mode(self) -> str:
        """"rccl" | "direct" | "copies" | "none" (qsim_cluster_exchange_mode)."""
        return (_lib.load().qsim_cluster_exchange_mode(self._h) or b"").decode()

    def close(self) -> None:
        if self._h:
            _lib.load().qsim_cluster_destroy(self._h)
            self._h = c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ShardPlanHandle:
    """qsim_shard_plan: the C++ planner's output for one circuit and shard count (host only)."""

    def __init__(self, circuit: Circuit, num_shards: int):
        self._h = c_void_p()
        lib = _lib.load()
        rc = lib.qsim_shard_plan_create(byref(self._h), circuit._h, num_shards)
        if rc:
            raise _lib.QsimError(rc, (lib.qsim_cluster_error() or b"").decode())
        self.num_shards = num_shards
        self.num_qubits = circuit.num_qubits
        self.num_steps = lib.qsim_shard_plan_num_steps(self._h)

    def step(self, i: int):
        """("local",) or ("exchange", shard_bits, local_positions)."""
        kind, k = c_int(), c_int()
        J, L = (c_int * 16)(), (c_int * 16)()
        check(_lib.load().qsim_shard_plan_step(self._h, i, byref(kind), byref(k), J, L))
        if kind.value == 0:
            return ("local",)
        return ("exchange", tuple(J[: k.value]), tuple(L[: k.value]))

    def local_ops(self, i: int, shard: int) -> list:
        """[("u1", pos, U) | ("cx", a, b) | ("scale", z)] for one shard."""
        out = []

        def cb(_u, kind, a, b, m):
            if kind == 1:
                out.append(("u1", a, np.ctypeslib.as_array(m, shape=(8,)).copy().view(np.complex128).reshape(2, 2)))
            elif kind == 2:
                out.append(("cx", a, b))
            else:
                out.append(("scale", complex(m[0], m[1])))

        check(_lib.load().qsim_shard_plan_local_ops(self._h, i, shard, _lib.LOCAL_OP_CB(cb), None))
        return out

    def apply_local(self, i: int, shard: int, sim: "Simulator") -> None:
        rc = _lib.load().qsim_shard_plan_apply_local(self._h, i, shard, sim._h)
        if rc:
            raise _lib.QsimError(rc, (_lib.load().qsim_cluster_error() or b"").decode())

    def step_support(self, i: int):
        """(mixed_local, mixed_rank) of an exchange step: where the register can be non-zero just before it."""
        from ctypes import c_uint64
        a, b = c_uint64(), c_uint64()
        check(_lib.load().qsim_shard_plan_step_support(self._h, i, byref(a), byref(b)))
        return int(a.value), int(b.value)

    def exchange_roles(self, i: int, shard: int) -> dict:
        """qsim_shard_plan_exchange_roles: what one shard sends, receives and holds afterwards in the exchange of step i."""
        r = _lib.QsimExchangeRoles()
        check(_lib.load().qsim_shard_plan_exchange_roles(self._h, i, shard, byref(r)))
        return r.as_dict()

    def final_pos(self) -> list:
        pos = (c_int * self.num_qubits)()
        check(_lib.load().qsim_shard_plan_final_pos(self._h, pos))
        return list(pos)

    def tune(self, shard: int, sim: "Simulator", max_candidates: int = 32, budget_ms: float = 6000.0) -> dict:
        """qsim_shard_plan_tune: geometry planning for one shard's local steps (leaves `sim` reset)."""
        rep = _lib.QsimTuneReport()
        rc = _lib.load().qsim_shard_plan_tune(self._h, shard, sim._h, max_candidates, budget_ms, byref(rep))
        if rc:
            raise _lib.QsimError(rc, (_lib.load().qsim_cluster_error() or b"").decode())
        return rep.as_dict()

    def predict(self, link_gbps: float = 50.0, pack_gbps: float = 5000.0):
        """(bytes each rank sends, seconds spent in exchanges) under the planner's cost model (qsim_shard_plan_predict)."""
        b, t = c_double(), c_double()
        check(_lib.load().qsim_shard_plan_predict(self._h, link_gbps, pack_gbps, byref(b), byref(t)))
        return b.value, t.value

    def close(self) -> None:
        if self._h:
            _lib.load().qsim_shard_plan_free(self._h)
            self._h = c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class RankComm:
    """qsim_rank_comm: this rank's end of the exchanges in a one-process-per-GPU job — RCCL send/recv issued by libqsim
    on the shard's own stream.  `unique_id()` (rank 0) produces the bytes every rank passes to the constructor."""

    @staticmethod
    def unique_id() -> bytes:
        buf = ctypes.create_string_buffer(128)
        rc = _lib.load().qsim_rccl_unique_id(buf)
        if rc:
            raise _lib.QsimError(rc, (_lib.load().qsim_cluster_error() or b"").decode())
        return buf.raw

    def __init__(self, sim: "Simulator", device: int, world: int, rank: int, uid: bytes, scratch_ptr: Optional[int] = None):
        self._h = c_void_p()
        lib = _lib.load()
        buf = ctypes.create_string_buffer(bytes(uid), 128)
        rc = lib.qsim_rank_comm_create(byref(self._h), sim._h, device, world, rank, buf,
                                       c_void_p(scratch_ptr) if scratch_ptr else None)
        if rc:
            raise _lib.QsimError(rc, (lib.qsim_cluster_error() or b"").decode())

    def exchange(self, shard_bits: Sequence[int], local_bits: Sequence[int]) -> None:
        k = len(shard_bits)
        rc = _lib.load().qsim_rank_comm_exchange(self._h, (c_int * k)(*shard_bits), (c_int * k)(*local_bits), k)
        if rc:
            raise _lib.QsimError(rc, (_lib.load().qsim_cluster_error() or b"").decode())

    def exchange_step(self, plan: "ShardPlanHandle", step: int) -> None:
        """qsim_rank_comm_exchange_step: the exchange of one plan step, leaving out what the plan knows to be zero."""
        rc = _lib.load().qsim_rank_comm_exchange_step(self._h, plan._h, step)
        if rc:
            raise _lib.QsimError(rc, (_lib.load().qsim_cluster_error() or b"").decode())

    def loopback(self, count: int) -> None:
        """`count` doubles of the shard through ncclSend -> ncclRecv to this same rank (qsim_rank_comm_loopback)."""
        rc = _lib.load().qsim_rank_comm_loopback(self._h, count)
        if rc:
            raise _lib.QsimError(rc, (_lib.load().qsim_cluster_error() or b"").decode())

    def pack_counts(self):
        """(re-layouts done by the last tile pass in front of an exchange, by the separate pack kernel)."""
        from ctypes import c_uint64
        a, b = c_uint64(), c_uint64()
        _lib.load().qsim_rank_comm_pack_counts(self._h, byref(a), byref(b))
        return int(a.value), int(b.value)

    def stats(self, reset: bool = False):
        """(exchanges, bytes sent by this rank, seconds its stream spent in exchanges)."""
        from ctypes import c_uint64
        n, b, t = c_uint64(), c_double(), c_double()
        rc = _lib.load().qsim_rank_comm_stats(self._h, byref(n), byref(b), byref(t), 1 if reset else 0)
        if rc:
            raise _lib.QsimError(rc, (_lib.load().qsim_cluster_error() or b"").decode())
        return int(n.value), float(b.value), float(t.value)

    def close(self) -> None:
        if self._h:
            _lib.load().qsim_rank_comm_destroy(self._h)
            self._h = c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def run_qasm(path: str, device: int = 0, **sim_options) -> np.ndarray:
    """QASM file in, amplitude vector out (the drop-in path of BASELINE.json's north_star)."""
    c = Circuit.from_file(path)
    with Simulator(c.num_qubits, device, **sim_options) as sim:
        sim.run(c)
        return sim.read()
