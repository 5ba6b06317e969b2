#!/bin/bash
out=gpurun_out/${1:-r02i}; mkdir -p $out
python -c "
import sys; sys.path.insert(0,'.')
from gpu_quantum_simulator_amd import launch
print('sysfs gpus:', launch.count_gpus())
import torch; print('torch device_count:', torch.cuda.device_count())
"
timeout -k 10 300 python bench.py --gpus 2 --steps 1 --warmup 0 > $out/bench_gpus2.json 2> $out/bench_gpus2.err; echo "plain --gpus 2 rc=$?"; tail -3 $out/bench_gpus2.err; cat $out/bench_gpus2.json | head -c 300
timeout -k 10 300 python bench.py --force-sharded --steps 2 --warmup 1 --sizes 24 --no-cpu-baseline > $out/bench_forced.json 2> $out/bench_forced.err; echo "force-sharded rc=$?"; tail -2 $out/bench_forced.err; python - <<PY
import json
d=json.loads(open('$out/bench_forced.json').read().strip().splitlines()[-1])
print(round(d['value']), d['n_ranks_seen'], d.get('exchange'), [ (s['qubits'], round(s['value'])) for s in d.get('sizes',[])])
PY
QSIM_SHARDS=4 QSIM_STATS=1 QSIM_MEASURE=1 gpu_quantum_simulator_amd/bin/qsim tests/golden/rand_n12_all.qasm 5; echo "cli sharded rc=$?"
