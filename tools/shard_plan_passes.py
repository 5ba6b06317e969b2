"""Host-only: passes and swept bytes of every local step of a shard plan (what each shard's engine will schedule).
Usage: python tools/shard_plan_passes.py [n] [P...]   (QSIM_SHARD_TAIL=0 for the plain planner)"""
import sys
sys.path.insert(0, '.')
from gpu_quantum_simulator_amd import Circuit, circuits, gate_matrix
from gpu_quantum_simulator_amd.distributed import ShardPlan, normalize_gates
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
Ps = [int(x) for x in sys.argv[2:]] or [2, 4, 8]
seed = 20240117 + n
gates = circuits.random_gates(n, 1000, seed, "all")
single = Circuit.from_gates(n, gates).plan()
S = 32.0 * (1 << n)
print(f"single: launches {single['launches']} sweeps {single['algorithmic_bytes'] / S:.2f}")
norm = normalize_gates(gates, gate_matrix)
for P in Ps:
    p = P.bit_length() - 1
    m = n - p
    tot = 0.0
    for r in sorted({0, P - 1}):
        pl = ShardPlan(n, p, norm, r)
        sup, empty = 0, r != 0
        row, sweeps = [], 0.0
        for i, st in enumerate(pl.steps):
            if st[0] == "exchange":
                ro = pl.handle.exchange_roles(i, r)
                empty, sup = bool(ro["empty_after"]), ro["new_support"]
                row.append("X")
                continue
            if empty:
                row.append("-")
                continue
            c = Circuit.empty(m)
            for op in st[1]:
                if op[0] == "cx": c.append_cx(op[1], op[2])
                elif op[0] == "u1": c.append_1q(op[2], op[1])
                else: c.append_1q([[op[1], 0], [0, op[1]]], 0)
            stt = c.plan(initial_support=sup)
            sw = stt["algorithmic_bytes"] / (32.0 * (1 << m))
            row.append(f"{stt['launches']}p/{sw:.2f}")
            sweeps += sw
            sup = (1 << m) - 1
        print(f"P={P} shard {r}: sweeps {sweeps:.2f}  " + " ".join(row))
