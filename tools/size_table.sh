#!/bin/bash
# gate-applies/s and tile-kernel bandwidth across BASELINE's sizes (1 GPU).
for cfg in "24 all 5" "26 all 5" "28 all 5" "28 clifford_t 5" "30 all 5" "32 all 2" "33 all 2"; do
  set -- $cfg
  python3 bench.py --qubits $1 --vocabulary $2 --steps $3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']; p = d.get('roofline_1q_probe') or {}
print(f\"n={d['config']['qubits']:2d} {d['config']['workload'].split('(')[1].split(')')[0]:10s} {d['value']:12.1f} gate-applies/s  {d['ms_per_step']:9.2f} ms/step  launches={d['launches_per_step']:.0f}  {r['kernel']} {r['achieved']:.0f} GB/s ({r['frac']:.2f})  1q-probe: \" + ' '.join(f\"{k}={v['achieved']:.0f}\" for k, v in p.items()))
"
done
