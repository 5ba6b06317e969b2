#!/bin/bash
# A/B: the -m gpu suite on the new build, then the bench on the new and the base build (tools/ab/libqsim_base.so)
set -o pipefail
out=gpurun_out/${1:-r02b}; mkdir -p $out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $out/pytest_gpu.log 2>&1; rc=$?; echo "pytest exit $rc" >> $out/pytest_gpu.log
tail -4 $out/pytest_gpu.log
[ $rc -ne 0 ] && exit 1
B="python bench.py --steps 4 --warmup 1 --no-cpu-baseline --sizes="
$B > $out/bench_new.json 2> $out/bench_new.err; echo "new $?"
QSIM_LIB=$PWD/tools/ab/libqsim_base.so $B > $out/bench_base.json 2> $out/bench_base.err; echo "base $?"
QSIM_SCHED_LOCAL=3 QSIM_SCHED_LOOKAHEAD=1 $B > $out/bench_new_local3.json 2> $out/bench_new_local3.err; echo "new local3 $?"
$B --tile-threads 1024 > $out/bench_new_t1024.json 2> $out/bench_new_t1024.err; echo "new t1024 $?"
$B --tile-bits 12 --tile-low-bits 4 > $out/bench_new_l4.json 2> $out/bench_new_l4.err; echo "new l4 $?"
$B --qubits 28 --vocabulary clifford_t > $out/bench_new_n28ct.json 2> $out/bench_new_n28ct.err; echo "new n28 $?"
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
