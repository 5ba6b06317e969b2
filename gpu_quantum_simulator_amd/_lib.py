"""ctypes binding of libqsim.so (include/qsim.h).  Loading fails loudly: there is no Python or CPU
fallback for any entry point — if the HIP extension is missing the package is unusable by design."""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_char_p, c_double, c_int, c_long, c_size_t, c_uint64, c_void_p

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(PKG_DIR, "libqsim.so")
CLI_PATH = os.path.join(PKG_DIR, "bin", "qsim")

QSIM_OK, ERR_ARG, ERR_ALLOC, ERR_DEVICE, ERR_OPEN, ERR_PARSE = range(6)
GATE_U1, GATE_CX, GATE_U2, GATE_U3 = 1, 2, 3, 4
OPT_FUSE, OPT_PROFILE, OPT_TILE_BITS, OPT_TILE_LOW_BITS, OPT_MAX_PENDING, OPT_TILE_MAX_OPS, OPT_GRID_CAP, OPT_TILE_THREADS = range(1, 9)
OPT_TILE_PAD_FROM = 9
OPT_DEBUG_SKIP_OPS = 10
OPT_DEBUG_SKIP_MEM = 11
OPT_DEBUG_TILE_ORDER = 12
OPT_PLAN_CACHE = 13
OPT_PINGPONG = 14
OPT_SPARSE_START = 15
OPT_DEBUG_PLAN_KEY = 16
K_NAMES = ("init", "gate1", "gate1_lo", "phase", "cx", "gate2", "tile", "pack")
K_COUNT = len(K_NAMES)


class QsimStats(ctypes.Structure):
    _fields_ = [("gates", c_uint64), ("launches", c_uint64), ("algorithmic_bytes", c_double),
                ("k_launches", c_uint64 * K_COUNT), ("k_bytes", c_double * K_COUNT), ("k_ms", c_double * K_COUNT)]

    def as_dict(self) -> dict:
        return {"gates": int(self.gates), "launches": int(self.launches),
                "algorithmic_bytes": float(self.algorithmic_bytes),
                "kernels": {K_NAMES[k]: {"launches": int(self.k_launches[k]), "bytes": float(self.k_bytes[k]),
                                         "ms": float(self.k_ms[k])} for k in range(K_COUNT)}}


class QsimPassInfo(ctypes.Structure):
    _fields_ = [("kernel_class", ctypes.c_int32), ("blocks", ctypes.c_int32), ("tile_mask", c_uint64), ("visited", c_double),
                ("bytes", c_double), ("cost_bytes", c_double)]


class QsimTuneReport(ctypes.Structure):
    _fields_ = [("tile_passes", c_int), ("already_known", c_int), ("passes_tuned", c_int), ("passes_reordered", c_int),
                ("candidates_timed", c_int), ("ms_ascending", c_double), ("ms_best", c_double), ("seconds", c_double)]

    def as_dict(self) -> dict:
        return {k: getattr(self, k) for k, _ in self._fields_}


class QsimExchangeRoles(ctypes.Structure):
    _fields_ = [("mine", c_int), ("empty_before", c_int), ("empty_after", c_int), ("keep_own", c_int),
                ("send", ctypes.c_uint32), ("recv", ctypes.c_uint32), ("unread", ctypes.c_uint32), ("new_support", c_uint64)]

    def as_dict(self) -> dict:
        return {k: int(getattr(self, k)) for k, _ in self._fields_}


SCHED_CB = ctypes.CFUNCTYPE(None, c_void_p, c_int, c_int, c_int, POINTER(c_int), c_int, POINTER(c_double), c_int)

LOCAL_OP_CB = ctypes.CFUNCTYPE(None, c_void_p, c_int, c_int, c_int, POINTER(c_double))

# every symbol include/qsim.h declares: name -> (restype, argtypes)
_DP = POINTER(c_double)
SIGNATURES = {
    "qsim_device_count": (c_int, []),
    "qsim_device_init": (c_int, [c_int]),
    "qsim_last_error": (c_char_p, []),
    "qsim_create": (c_int, [POINTER(c_void_p), c_int, c_int]),
    "qsim_create_f32": (c_int, [POINTER(c_void_p), c_int, c_int]),
    "qsim_create_async": (c_int, [POINTER(c_void_p), c_int, c_int, c_int]),
    "qsim_precision_bits": (c_int, [c_void_p]),
    "qsim_create_external": (c_int, [POINTER(c_void_p), c_int, c_int, c_void_p]),
    "qsim_destroy": (None, [c_void_p]),
    "qsim_reset": (c_int, [c_void_p]),
    "qsim_reset_shard": (c_int, [c_void_p, c_int]),
    "qsim_num_qubits": (c_int, [c_void_p]),
    "qsim_set_option": (c_int, [c_void_p, c_int, c_long]),
    "qsim_get_option": (c_long, [c_void_p, c_int]),
    "qsim_apply_1q": (c_int, [c_void_p, _DP, c_int]),
    "qsim_apply_cx": (c_int, [c_void_p, c_int, c_int]),
    "qsim_apply_2q": (c_int, [c_void_p, _DP, c_int, c_int]),
    "qsim_flush": (c_int, [c_void_p]),
    "qsim_sync": (c_int, [c_void_p]),
    "qsim_plan_cache_stats": (c_int, [c_void_p, POINTER(c_uint64), POINTER(c_uint64), POINTER(c_uint64)]),
    "qsim_read": (c_int, [c_void_p, c_uint64, c_uint64, _DP]),
    "qsim_write": (c_int, [c_void_p, c_uint64, c_uint64, _DP]),
    "qsim_norm2": (c_int, [c_void_p, _DP]),
    "qsim_device_ptr": (c_void_p, [c_void_p]),
    "qsim_stream": (c_void_p, [c_void_p]),
    "qsim_sample": (c_int, [c_void_p, _DP, c_long, POINTER(c_uint64)]),
    "qsim_draw_randn": (c_double, []),
    "qsim_putb": (None, [ctypes.c_longlong, c_int, c_char_p]),
    "qsim_pack_bits": (c_int, [c_void_p, POINTER(c_int), c_int, c_void_p]),
    "qsim_pack_bits_to": (c_int, [c_void_p, POINTER(c_int), c_int, POINTER(c_void_p)]),
    "qsim_swap_buffer": (c_int, [c_void_p, POINTER(c_void_p)]),
    "qsim_set_spare_buffer": (c_int, [c_void_p, c_void_p]),
    "qsim_set_support": (c_int, [c_void_p, c_uint64]),
    "qsim_get_support": (c_int, [c_void_p, POINTER(c_uint64), POINTER(c_int), _DP]),
    "qsim_holds_nothing": (c_int, [c_void_p]),
    "qsim_pack_bits_sparse": (c_int, [c_void_p, POINTER(c_int), c_int, c_void_p, POINTER(c_void_p), ctypes.c_uint32]),
    "qsim_state_buffer": (c_void_p, [c_void_p]),
    "qsim_flush_pack": (c_int, [c_void_p, POINTER(c_int), c_int, POINTER(c_int), c_uint64, c_void_p, c_uint64, ctypes.c_uint32, POINTER(c_void_p), POINTER(c_int)]),
    "qsim_cluster_pack_counts": (c_int, [c_void_p, POINTER(c_uint64), POINTER(c_uint64)]),
    "qsim_rank_comm_pack_counts": (c_int, [c_void_p, POINTER(c_uint64), POINTER(c_uint64)]),
    "qsim_cluster_exchange_bytes_moved": (c_int, [c_void_p, _DP]),
    "qsim_shard_plan_step_support": (c_int, [c_void_p, c_int, POINTER(c_uint64), POINTER(c_uint64)]),
    "qsim_rank_comm_exchange_step": (c_int, [c_void_p, c_void_p, c_int]),
    "qsim_shard_plan_exchange_roles": (c_int, [c_void_p, c_int, c_int, POINTER(QsimExchangeRoles)]),
    "qsim_block_prob_masked": (c_int, [c_void_p, c_uint64, c_uint64, _DP]),
    "qsim_gather_masked": (c_int, [c_void_p, c_uint64, c_uint64, _DP]),
    "qsim_scale": (c_int, [c_void_p, c_double, c_double]),
    "qsim_cluster_create": (c_int, [POINTER(c_void_p), c_int, c_int, POINTER(c_int)]),
    "qsim_cluster_destroy": (None, [c_void_p]),
    "qsim_cluster_num_shards": (c_int, [c_void_p]),
    "qsim_cluster_shard": (c_void_p, [c_void_p, c_int]),
    "qsim_cluster_set_option": (c_int, [c_void_p, c_int, c_long]),
    "qsim_cluster_reset": (c_int, [c_void_p]),
    "qsim_cluster_run_circuit": (c_int, [c_void_p, c_void_p]),
    "qsim_cluster_sync": (c_int, [c_void_p]),
    "qsim_cluster_read": (c_int, [c_void_p, c_uint64, c_uint64, _DP]),
    "qsim_cluster_norm2": (c_int, [c_void_p, _DP]),
    "qsim_cluster_sample": (c_int, [c_void_p, _DP, c_long, POINTER(c_uint64)]),
    "qsim_cluster_exchange_stats": (c_int, [c_void_p, POINTER(c_uint64), _DP]),
    "qsim_cluster_error": (c_char_p, []),
    "qsim_cluster_exchange_mode": (c_char_p, [c_void_p]),
    "qsim_shard_plan_tune": (c_int, [c_void_p, c_int, c_void_p, c_int, c_double, POINTER(QsimTuneReport)]),
    "qsim_shard_plan_predict": (c_int, [c_void_p, c_double, c_double, _DP, _DP]),
    "qsim_rccl_unique_id": (c_int, [c_void_p]),
    "qsim_rank_comm_create": (c_int, [POINTER(c_void_p), c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "qsim_rank_comm_destroy": (None, [c_void_p]),
    "qsim_rank_comm_exchange": (c_int, [c_void_p, POINTER(c_int), POINTER(c_int), c_int]),
    "qsim_rank_comm_stats": (c_int, [c_void_p, POINTER(c_uint64), _DP, _DP, c_int]),
    "qsim_rank_comm_loopback": (c_int, [c_void_p, c_uint64]),
    "qsim_shard_plan_create": (c_int, [POINTER(c_void_p), c_void_p, c_int]),
    "qsim_shard_plan_free": (None, [c_void_p]),
    "qsim_shard_plan_num_steps": (c_int, [c_void_p]),
    "qsim_shard_plan_step": (c_int, [c_void_p, c_int, POINTER(c_int), POINTER(c_int), POINTER(c_int), POINTER(c_int)]),
    "qsim_shard_plan_final_pos": (c_int, [c_void_p, POINTER(c_int)]),
    "qsim_shard_plan_local_ops": (c_int, [c_void_p, c_int, c_int, LOCAL_OP_CB, c_void_p]),
    "qsim_shard_plan_apply_local": (c_int, [c_void_p, c_int, c_int, c_void_p]),
    "qsim_get_stats": (c_int, [c_void_p, POINTER(QsimStats)]),
    "qsim_reset_stats": (c_int, [c_void_p]),
    "qsim_launch_log": (c_long, [c_void_p, c_long, POINTER(c_int), POINTER(c_int), POINTER(c_uint64), POINTER(c_double)]),
    "qsim_launch_log_order": (c_int, [c_void_p, c_long, POINTER(c_int), POINTER(c_int)]),
    "qsim_launch_log_visited": (c_int, [c_void_p, c_long, _DP]),
    "qsim_launch_log_blocks": (c_int, [c_void_p, c_long, c_void_p, c_int, POINTER(c_int)]),
    "qsim_tune_circuit": (c_int, [c_void_p, c_void_p, c_int, c_double, POINTER(QsimTuneReport)]),
    "qsim_choose_schedule": (c_int, [c_void_p, c_void_p]),
    "qsim_choose_schedule_while_allocating": (c_int, [c_void_p, c_void_p]),
    "qsim_tune_circuit_from": (c_int, [c_void_p, c_void_p, c_int, c_double, POINTER(QsimTuneReport), c_int]),
    "qsim_choose_schedule_for": (c_int, [c_void_p, c_void_p, c_uint64]),
    "qsim_support_after": (c_int, [c_void_p, c_void_p, c_uint64, POINTER(c_uint64)]),
    "qsim_tune_circuit_support": (c_int, [c_void_p, c_void_p, c_int, c_double, POINTER(QsimTuneReport), c_uint64]),
    "qsim_cluster_plan": (c_int, [c_void_p, c_void_p, c_int, c_double]),
    "qsim_tune_table_size": (c_long, []),
    "qsim_tune_table_clear": (None, []),
    "qsim_tune_table_save": (c_int, [c_char_p]),
    "qsim_tune_table_load": (c_long, [c_char_p]),
    "qsim_circuit_parse_file": (c_int, [c_char_p, POINTER(c_void_p)]),
    "qsim_circuit_parse_text": (c_int, [c_char_p, c_size_t, POINTER(c_void_p)]),
    "qsim_circuit_create": (c_int, [c_int, POINTER(c_void_p)]),
    "qsim_circuit_free": (None, [c_void_p]),
    "qsim_circuit_num_qubits": (c_int, [c_void_p]),
    "qsim_circuit_num_gates": (c_long, [c_void_p]),
    "qsim_circuit_append_1q": (c_int, [c_void_p, _DP, c_int]),
    "qsim_circuit_append_cx": (c_int, [c_void_p, c_int, c_int]),
    "qsim_circuit_append_2q": (c_int, [c_void_p, _DP, c_int, c_int]),
    "qsim_circuit_gate": (c_int, [c_void_p, c_long, POINTER(c_int), POINTER(c_int), POINTER(c_int), _DP]),
    "qsim_circuit_error": (c_char_p, []),
    "qsim_run_circuit": (c_int, [c_void_p, c_void_p, c_long, c_long]),
    "qsim_gate_matrix": (c_int, [c_char_p, _DP]),
    "qsim_plan_circuit": (c_int, [c_void_p, c_int, c_int, c_int, POINTER(QsimStats)]),
    "qsim_plan_circuit_from": (c_int, [c_void_p, c_int, c_int, c_int, c_uint64, POINTER(QsimStats)]),
    "qsim_plan_passes": (c_int, [c_void_p, c_int, c_int, c_int, c_uint64, POINTER(QsimPassInfo), c_int, POINTER(c_int)]),
    "qsim_schedule_circuit": (c_int, [c_void_p, c_int, c_int, c_int, c_int, SCHED_CB, c_void_p]),
}
# include/qsim_legacy.h
LEGACY_SYMBOLS = ("compute_state_vector", "execute_single_qubit_gate", "execute_cnot")

_lib = None


class QsimError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"qsim error {code}: {message}")
        self.code = code


def load() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `make -C gpu_quantum_simulator_amd/csrc` "
                "(or __graft_entry__.build()).  There is no fallback implementation.")
        # torch wheels bundle their own HIP runtime (torch/lib/libamdhip64.so) next to the system one libqsim
        # links (/opt/rocm/lib/libamdhip64.so.7).  Both can live in one process, but only if torch's copy is
        # loaded FIRST (measured on MI355X / ROCm 7.2: the other order leaves torch with "No HIP GPUs").
        # Anything that will also use torch (bench.py, the sharded path, the tests) therefore gets torch
        # imported here; a process that never imports torch is unaffected.
        import sys
        if "torch" not in sys.modules and os.environ.get("QSIM_NO_TORCH_PRELOAD", "") == "":
            try:
                import torch  # noqa: F401
            except ImportError:
                pass
        # QSIM_LIB: another build of the same library (A/B runs of kernel variants from tools/); never a different backend
        lib = ctypes.CDLL(os.environ.get("QSIM_LIB") or LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError here = header and library disagree
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def check(rc: int) -> None:
    if rc != QSIM_OK:
        lib = load()
        msg = lib.qsim_last_error() or b""
        if rc in (ERR_OPEN, ERR_PARSE) and not msg:
            msg = lib.qsim_circuit_error() or b""
        raise QsimError(rc, msg.decode(errors="replace"))
