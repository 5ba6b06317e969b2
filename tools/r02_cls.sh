#!/bin/bash
out=gpurun_out/${1:-r02k}; mkdir -p $out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $out/pytest_gpu.log 2>&1; rc=$?; echo "pytest exit $rc" >> $out/pytest_gpu.log
tail -5 $out/pytest_gpu.log
[ $rc -ne 0 ] && exit 1
python tools/phase_split.py 30 > $out/phase.log 2>&1; tail -1 $out/phase.log
QSIM_SCHED_LOCAL=3 QSIM_SCHED_LOOKAHEAD=1 python tools/phase_split.py 30 > $out/phase_local3.log 2>&1; tail -1 $out/phase_local3.log
python tools/phase_split.py 24 > $out/phase_n24.log 2>&1; tail -1 $out/phase_n24.log
B="python bench.py --steps 4 --warmup 1 --no-cpu-baseline --sizes="
$B > $out/bench_new.json 2> $out/bench_new.err; echo "new $?"
QSIM_SCHED_LOCAL=3 QSIM_SCHED_LOOKAHEAD=1 $B > $out/bench_new_local3.json 2> $out/bench_new_local3.err; echo "new local3 $?"
$B --precision 32 > $out/bench_new_f32.json 2> $out/bench_new_f32.err; echo "f32 $?"
$B --qubits 28 --vocabulary clifford_t > $out/bench_new_n28ct.json 2> $out/bench_new_n28ct.err
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
