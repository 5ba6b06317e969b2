#!/bin/bash
out=gpurun_out/${1:-r02y}; mkdir -p $out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $out/pytest_gpu.log 2>&1; rc=$?; echo "pytest exit $rc" >> $out/pytest_gpu.log
tail -5 $out/pytest_gpu.log
[ $rc -ne 0 ] && exit 1
python bench.py --steps 5 --warmup 1 --no-cpu-baseline > $out/bench.json 2> $out/bench.err; echo "bench $?"
python - <<PY
import json
d=json.loads(open('$out/bench.json').read().strip().splitlines()[-1])
print(round(d['value']), round(d['ms_per_step'],2), round(d['roofline']['frac'],3), [(s['qubits'], round(s['value']), round(s['ms_per_step'],3)) for s in d['sizes']])
PY
python tools/small_n.py > $out/small_n.log 2>&1; tail -8 $out/small_n.log
