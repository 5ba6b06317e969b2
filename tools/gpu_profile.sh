#!/bin/bash
# Round evidence on the GPU box: bench line, single-qubit probes, rocprofv3 kernel-trace stats and PMC traffic.
# Usage (via gpurun): bash tools/gpu_profile.sh r01
set -e
TAG=${1:-r01}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python3 bench.py --steps 5 --warmup 1 --wisdom $OUT/wisdom.txt > $OUT/bench_n30.json 2> $OUT/bench_n30.err   # plans the geometries and saves them
tail -c 2000 $OUT/bench_n30.json
for q in 0 5 6 12 20 29; do
  python3 bench.py --probe $q --depth 30 --steps 2 --warmup 1 --no-cpu-baseline --sizes= >> $OUT/probe_1q_n30.jsonl 2>> $OUT/probe.err
done
cat $OUT/probe_1q_n30.jsonl | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); r = d['roofline']
    print(d['config']['workload'], '->', round(r['achieved'],1), 'GB/s', round(r['frac'],3), r['kernel'], round(r['avg_launch_ms'],3), 'ms')
"
# kernel trace + stats of the same bench command
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/rocprof_trace -o bench -- python3 $OLDPWD/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-full-sweeps --no-precision32 --no-one-shot --sizes= --wisdom $OUT/wisdom.txt > $OUT/rocprof_trace.log 2>&1 || echo "rocprof trace failed"
# PMC: separate passes (FETCH_SIZE and WRITE_SIZE do not fit one pass); short run to keep the CSV small
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/rocprof_fetch -o bench -- python3 $OLDPWD/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-full-sweeps --no-precision32 --no-one-shot --sizes= --wisdom $OUT/wisdom.txt > $OUT/rocprof_fetch.log 2>&1 || echo "rocprof fetch failed"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/rocprof_write -o bench -- python3 $OLDPWD/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-full-sweeps --no-precision32 --no-one-shot --sizes= --wisdom $OUT/wisdom.txt > $OUT/rocprof_write.log 2>&1 || echo "rocprof write failed"
cd $OLDPWD
find $OUT -name "*.csv" | head -20
