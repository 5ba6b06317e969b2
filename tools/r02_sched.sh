#!/bin/bash
out=gpurun_out/${1:-r02l}; mkdir -p $out
B="python bench.py --steps 4 --warmup 1 --no-cpu-baseline --sizes="
run() { name=$1; shift; "$@" > $out/bench_$name.json 2> $out/bench_$name.err; echo "$name $?"; }
run default $B
run mo24 $B --tile-max-ops 24
QSIM_SCHED_LOCAL=3 QSIM_SCHED_LOOKAHEAD=1 run local3_mo24 $B --tile-max-ops 24
QSIM_SCHED_LOCAL=3 QSIM_SCHED_LOOKAHEAD=1 run local3_mo32 $B
QSIM_SCHED_LOCAL=3 QSIM_SCHED_LOOKAHEAD=1 run local3_mo28 $B --tile-max-ops 28
QSIM_SCHED_LOCAL=3 QSIM_SCHED_LOOKAHEAD=2 run local3_la2_mo24 $B --tile-max-ops 24
QSIM_SCHED_LOCAL=3 QSIM_SCHED_LOOKAHEAD=1 run n28_local3_mo24 $B --tile-max-ops 24 --qubits 28
run n28_default $B --qubits 28
QSIM_SCHED_LOCAL=3 QSIM_SCHED_LOOKAHEAD=1 run n26_local3_mo24 $B --tile-max-ops 24 --qubits 26
run n26_default $B --qubits 26
QSIM_SCHED_LOCAL=3 QSIM_SCHED_LOOKAHEAD=1 run n32_local3_mo24 $B --tile-max-ops 24 --qubits 32 --steps 2
run n32_default $B --qubits 32 --steps 2
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
