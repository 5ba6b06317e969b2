#!/bin/bash
out=gpurun_out/${1:-r02r}; mkdir -p $out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $out/pytest_gpu.log 2>&1; rc=$?; echo "pytest exit $rc" >> $out/pytest_gpu.log
tail -5 $out/pytest_gpu.log
[ $rc -ne 0 ] && exit 1
B="python bench.py --steps 4 --warmup 1 --no-cpu-baseline --sizes="
run() { name=$1; shift; "$@" > $out/bench_$name.json 2> $out/bench_$name.err; echo "$name $?"; }
run notune $B --no-tune
run tune48 $B
run tune128 $B --tune-candidates 128 --tune-ms 20000
run n28 $B --qubits 28
run n32 $B --qubits 32 --steps 2
run f32 $B --precision 32
python - <<PY
import json,glob
for f in sorted(glob.glob('$out/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        r=d['roofline']
        print(f.split('/')[-1], round(d['value']), 'ms/step', round(d['ms_per_step'],2), 'launches', d['launches_per_step'], 'tile ms', round(r['avg_launch_ms'],3), 'frac', round(r['frac'],3), 'plan', d.get('geometry_planning'))
    except Exception as e:
        print(f, 'ERR', e)
PY
