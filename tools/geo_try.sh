#!/bin/bash
# tile-shape variants of the n=30 bench step (steady state and with full sweeps)
mkdir -p gpurun_out/r03
for v in "" "--tile-bits 13" "--tile-bits 13 --tile-low-bits 4" "--tile-bits 13 --tile-threads 512" "--tile-bits 12 --tile-low-bits 4"; do
  python3 bench.py --steps 3 --warmup 1 --sizes= --no-cpu-baseline --no-precision32 --no-one-shot $v 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$v', '| value', round(d['value']), 'ms', round(d['ms_per_step'],2), 'launches', d['launches_per_step'], 'frac', round(d['roofline']['frac'],3), 'full sweeps', round(d['sparse_start']['with_full_sweeps']['value']), round(d['sparse_start']['with_full_sweeps']['ms_per_step'],2))
"
done
