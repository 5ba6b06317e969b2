#!/bin/bash
# round-2 opening run: the whole -m gpu suite, then the baseline bench and a geometry sweep with the unchanged kernels
set -o pipefail
mkdir -p gpurun_out/r02a
python -m pytest tests -m gpu -x -q > gpurun_out/r02a/pytest_gpu.log 2>&1; echo "pytest exit $?" >> gpurun_out/r02a/pytest_gpu.log
tail -3 gpurun_out/r02a/pytest_gpu.log
B="python bench.py --steps 4 --warmup 1 --no-cpu-baseline"
$B > gpurun_out/r02a/bench_default.json 2> gpurun_out/r02a/bench_default.err; echo "default $?"
for geo in "13 3" "13 4" "12 4" "12 5" "11 3"; do
  set -- $geo
  timeout -k 10 200 $B --tile-bits $1 --tile-low-bits $2 > gpurun_out/r02a/bench_b$1_l$2.json 2> gpurun_out/r02a/bench_b$1_l$2.err; echo "geo $geo $?"
done
QSIM_SCHED_LOCAL=3 QSIM_SCHED_LOOKAHEAD=1 timeout -k 10 200 $B > gpurun_out/r02a/bench_local3.json 2> gpurun_out/r02a/bench_local3.err; echo "local3 $?"
QSIM_SCHED_LOCAL=3 QSIM_SCHED_LOOKAHEAD=1 timeout -k 10 200 $B --tile-bits 13 --tile-low-bits 4 > gpurun_out/r02a/bench_local3_b13_l4.json 2> gpurun_out/r02a/bench_local3_b13_l4.err; echo "local3 13/4 $?"
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r02a/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        r=d['roofline']
        print(f.split('/')[-1], round(d['value']), 'ms/step', round(d['ms_per_step'],2), 'launches', d['launches_per_step'], 'tile ms', round(r['avg_launch_ms'],3), 'frac', round(r['frac'],3))
    except Exception as e:
        print(f, 'ERR', e)
PY
