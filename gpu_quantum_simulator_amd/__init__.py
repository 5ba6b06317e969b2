"""gpu_quantum_simulator_amd — MI355X-native state-vector simulator, drop-in for the QASM-file-in path of
RiccardoFiorentini/GPU_quantum_simulator's quantum_simulator.c.

The product is libqsim.so (hand-written HIP kernels for gfx950 + C/C++ host, C ABI in include/qsim.h) and the
C host `bin/qsim`.  This Python package is the host-side mirror of that ABI used by tests and bench.py;
importing it never loads anything from oracle/.
"""
from . import circuits  # noqa: F401  (pure Python, no native code)

__all__ = ["circuits", "Simulator", "Circuit", "Cluster", "run_qasm", "gate_matrix", "ShardPlanHandle", "RankComm"]


def __getattr__(name):  # lazy: `import gpu_quantum_simulator_amd.circuits` must work before the library is built
    if name in ("Simulator", "Circuit", "Cluster", "run_qasm", "gate_matrix", "ShardPlanHandle", "RankComm"):
        from . import simulator
        return getattr(simulator, name)
    raise AttributeError(name)
