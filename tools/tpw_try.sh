#!/bin/bash
# tiles per workgroup (through --grid-cap) at the smaller registers: launch tail / ramp vs exposed first loads
for n in 26 28; do
  for cap in 0 4096 8192 16384 32768 65536; do
    python3 bench.py --qubits $n --steps 5 --warmup 2 --sizes= --no-cpu-baseline --no-tune --grid-cap $cap 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('n=$n grid_cap=$cap', '| value', round(d['value']), 'ms', round(d['ms_per_step'],3), 'launches', d['launches_per_step'], 'frac', round(d['roofline']['frac'],3), 'full sweeps ms', round(d['sparse_start']['with_full_sweeps']['ms_per_step'],3))
"
  done
done
