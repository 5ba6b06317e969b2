#!/bin/bash
# Round-4 evidence, part A: tests, smoke, then the judged profile set (bench line, probes, kernel trace, PMC traffic, SQ counters)
tag=${1:-r04}
out=gpurun_out/$tag; mkdir -p $out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $out/pytest_gpu.log 2>&1; rc=$?; echo "pytest exit $rc" >> $out/pytest_gpu.log
tail -4 $out/pytest_gpu.log
[ $rc -ne 0 ] && exit 1
python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; echo "smoke $?"
rm -f $out/probe_1q_n30.jsonl $out/wisdom.txt
bash tools/gpu_profile.sh $tag > $out/profile.log 2>&1; echo "profile $?"
bash tools/pmc_tile.sh $tag/pmc_tile $tag > $out/pmc_tile.log 2>&1; echo "pmc $?"
python - <<PY
import json
d=json.loads(open('$out/bench_n30.json').read().strip().splitlines()[-1])
print(round(d['value']), round(d['ms_per_step'],2), d['launches_per_step'], d['roofline'])
print([(s['qubits'], round(s['value']), round(s['roofline']['frac'],3)) for s in d['sizes']])
print(d['one_shot']['cli'], d['one_shot']['in_process_first_step_ms'])
print(d['cpu_baseline'])
PY
find $out -name "*.csv" -size +8M -delete
