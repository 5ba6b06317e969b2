#!/bin/bash
# quick A/B of the bench only (no test suite): new build vs tools/ab/libqsim_base.so, plus variants given as extra env lines
set -o pipefail
out=gpurun_out/${1:-r02q}; mkdir -p $out
B="python bench.py --steps 4 --warmup 1 --no-cpu-baseline --sizes="
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "golden or sparse_block or geometry_sweep or selectors or random_circuits" > $out/pytest_subset.log 2>&1; echo "pytest subset exit $?"; tail -2 $out/pytest_subset.log
$B > $out/bench_new.json 2> $out/bench_new.err; echo "new $?"
QSIM_SCHED_LOCAL=3 QSIM_SCHED_LOOKAHEAD=1 $B > $out/bench_new_local3.json 2> $out/bench_new_local3.err; echo "new local3 $?"
$B --tile-threads 1024 > $out/bench_new_t1024.json 2> $out/bench_new_t1024.err; echo "new t1024 $?"
$B --tile-bits 12 --tile-low-bits 4 > $out/bench_new_l4.json 2> $out/bench_new_l4.err; echo "new l4 $?"
$B --precision 32 > $out/bench_new_f32.json 2> $out/bench_new_f32.err; echo "new f32 $?"
QSIM_LIB=$PWD/tools/ab/libqsim_base.so $B --precision 32 > $out/bench_base_f32.json 2> $out/bench_base_f32.err; echo "base f32 $?"
python - <<PY
import json,glob
for f in sorted(glob.glob('$out/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        r=d['roofline']
        print(f.split('/')[-1], round(d['value']), 'ms/step', round(d['ms_per_step'],2), 'launches', d['launches_per_step'], 'tile ms', round(r['avg_launch_ms'],3), 'frac', round(r['frac'],3), 'norm2', d['norm2'])
    except Exception as e:
        print(f, 'ERR', e)
PY
