"""The cold figures on their own: bin/qsim as a child process on the n-qubit bench circuit.  Usage: python tools/one_shot.py [n] [reps]"""
import json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import bench
from gpu_quantum_simulator_amd import circuits
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
gates = circuits.random_gates(n, 1000, 20240117 + n, "all")
for _ in range(int(sys.argv[2]) if len(sys.argv) > 2 else 2):
    print(json.dumps(bench.one_shot_cli(n, gates)), flush=True)
