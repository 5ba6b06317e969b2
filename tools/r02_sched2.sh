#!/bin/bash
out=gpurun_out/${1:-r02m}; mkdir -p $out
B="python bench.py --steps 4 --warmup 1 --no-cpu-baseline --sizes="
run() { name=$1; shift; "$@" > $out/bench_$name.json 2> $out/bench_$name.err; echo "$name $?"; }
export QSIM_SCHED_LOCAL=3 QSIM_SCHED_LOOKAHEAD=1
run base $B --tile-max-ops 24
QSIM_SCHED_LOOKAHEAD=2 run la2 $B --tile-max-ops 24
QSIM_SCHED_LOOKAHEAD=3 run la3 $B --tile-max-ops 24
QSIM_SCHED_LOCAL=6 QSIM_SCHED_LOOKAHEAD=2 run loc6_la2 $B --tile-max-ops 24
QSIM_SCHED_ROLLOUT=16 run roll16 $B --tile-max-ops 24
QSIM_SCHED_OBJ=1 run obj1 $B --tile-max-ops 24
QSIM_SCHED_LOOKAHEAD=2 run la2_mo22 $B --tile-max-ops 22
QSIM_SCHED_LOOKAHEAD=2 run la2_mo26 $B --tile-max-ops 26
QSIM_SCHED_LOOKAHEAD=2 QSIM_SCHED_WINDOW=1024 run la2_win1024 $B --tile-max-ops 24
python - <<PY
import json,glob
for f in sorted(glob.glob('$out/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        r=d['roofline']
        print(f.split('/')[-1], round(d['value']), 'ms/step', round(d['ms_per_step'],2), 'launches', d['launches_per_step'], 'tile ms', round(r['avg_launch_ms'],3), 'frac', round(r['frac'],3))
    except Exception as e:
        print(f, 'ERR', e)
PY
