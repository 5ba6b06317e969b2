#!/bin/bash
# runs the n=30 bench step with every tools/ab/libqsim_*.so (and the tree's own), twice, alternating
cd "$(dirname "$0")/.."
for rep in 1 2; do
  for lib in "" tools/ab/libqsim_*.so; do
    QSIM_LIB=${lib:+$PWD/$lib} python3 bench.py --steps 3 --warmup 1 --sizes= --no-cpu-baseline --no-precision32 --no-one-shot --tune-ms 3000 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('${lib:-tree}', '| value', round(d['value']), 'ms', round(d['ms_per_step'],2), 'frac', round(d['roofline']['frac'],4), 'full sweeps', round(d['sparse_start']['with_full_sweeps']['value']), round(d['sparse_start']['with_full_sweeps']['ms_per_step'],2))
"
  done
done
