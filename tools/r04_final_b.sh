#!/bin/bash
# Round-4 evidence, part B: phase splits at three sizes, sharding overhead without links (n = 30 and the n = 33 local leg), stress parity
tag=${1:-r04}
out=gpurun_out/$tag; mkdir -p $out
for n in 28 30 32; do timeout -k 10 300 python3 tools/phase_split.py $n > $out/phase_split_n$n.log 2>&1; tail -1 $out/phase_split_n$n.log; done
timeout -k 10 300 python3 tools/cluster_bench.py > $out/cluster_bench_virtual_shards.log 2>&1; tail -4 $out/cluster_bench_virtual_shards.log
timeout -k 10 400 python3 tools/cluster_bench.py 33 model 8 > $out/cluster_bench_n33.log 2>&1; tail -2 $out/cluster_bench_n33.log
timeout -k 10 600 python3 tests/stress_parity.py 800 404 > $out/stress_parity_800_cases.log 2>&1; tail -3 $out/stress_parity_800_cases.log
STRESS_KIND=cluster timeout -k 10 500 python3 tests/stress_parity.py 400 405 > $out/stress_parity_400_sharded_cases.log 2>&1; tail -3 $out/stress_parity_400_sharded_cases.log
