#!/usr/bin/env python3
"""Regenerates tests/golden/ from the REAL reference (container only).

Recipe: `make -C oracle ref` compiles /root/reference/quantum_simulator.c, from where it lies, into
oracle/_ref/libqsref.so (main renamed away).  This script calls its compute_state_vector()
(quantum_simulator.c:115) through ctypes on each circuit below and stores the returned amplitudes as
.npy (float64, shape [2^n, 2] = re, im) next to the circuit text.  The two reference sample circuits
are copied in as data; everything else is produced by gpu_quantum_simulator_amd.circuits with fixed
seeds.  Only inputs and expected outputs are stored here — no reference source.

The GPU box has no /root/reference: tests read the committed files and never call this script.
"""
import ctypes
import json
import os
import shutil
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from gpu_quantum_simulator_amd import circuits  # noqa: E402

REFERENCE = os.environ.get("REFERENCE", "/root/reference")

# name, n, depth, seed, vocabulary, text options
RANDOM_CASES = [
    ("rand_n3_all_lf", 3, 60, 11, "all", {}),
    ("rand_n5_all_crlf_physical", 5, 120, 12, "all", {"crlf": True, "physical": True}),
    ("rand_n8_all_suffix", 8, 200, 13, "all", {"qubit_style": "suffix"}),
    ("rand_n9_clifford_t", 9, 300, 14, "clifford_t", {"crlf": True}),
    ("rand_n10_all", 10, 400, 15, "all", {}),
    ("rand_n12_all", 12, 500, 16, "all", {}),
    ("rand_n12_clifford_t_physical", 12, 500, 17, "clifford_t", {"physical": True}),
]


def main():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref"])
    lib = ctypes.CDLL(os.path.join(ROOT, "oracle", "_ref", "libqsref.so"))
    lib.compute_state_vector.restype = ctypes.c_void_p
    lib.compute_state_vector.argtypes = [ctypes.c_char_p, ctypes.POINTER(ctypes.c_int)]
    libc = ctypes.CDLL(None)
    libc.free.argtypes = [ctypes.c_void_p]

    def run_reference(path):
        n = ctypes.c_int(0)
        p = lib.compute_state_vector(path.encode(), ctypes.byref(n))
        if not p:
            raise RuntimeError(f"reference rejected {path}")
        a = np.ctypeslib.as_array(ctypes.cast(p, ctypes.POINTER(ctypes.c_double)), shape=((1 << n.value), 2)).copy()
        libc.free(p)
        return n.value, a

    manifest = {}
    for name in ("entanglement", "grover_3_18"):
        dst = os.path.join(HERE, name + ".qasm")
        shutil.copyfile(os.path.join(REFERENCE, name + ".qasm"), dst)
        n, amps = run_reference(dst)
        np.save(os.path.join(HERE, name + ".npy"), amps)
        manifest[name] = {"n": n, "source": "reference sample circuit (data file)"}

    # grover_3_18 with the register widened to 18 qubits: gates touch q0..q5 only, so only
    # the first 64 amplitudes are non-zero; store those plus the exact-zero count.
    wide = os.path.join(HERE, "grover_3_18_n18.qasm")
    with open(os.path.join(REFERENCE, "grover_3_18.qasm"), newline="") as f:
        text = f.read().replace("qubit[6] q;", "qubit[18] q;")
    with open(wide, "w", newline="") as f:
        f.write(text)
    n, amps = run_reference(wide)
    assert n == 18 and not amps[64:].any()
    np.save(os.path.join(HERE, "grover_3_18_n18.first64.npy"), amps[:64])
    manifest["grover_3_18_n18"] = {"n": 18, "stored": "first 64 amplitudes; all others are exactly 0",
                                   "source": "grover_3_18.qasm with line 3 rewritten to qubit[18] q;"}

    for name, n, depth, seed, vocab, opts in RANDOM_CASES:
        path = os.path.join(HERE, name + ".qasm")
        circuits.random_circuit_file(path, n, depth, seed, vocab, **opts)
        rn, amps = run_reference(path)
        assert rn == n
        np.save(os.path.join(HERE, name + ".npy"), amps)
        manifest[name] = {"n": n, "depth": depth, "seed": seed, "vocabulary": vocab, "text": opts}

    with open(os.path.join(HERE, "MANIFEST.json"), "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)
    print("golden fixtures written:", ", ".join(sorted(manifest)))


if __name__ == "__main__":
    main()
