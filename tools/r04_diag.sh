#!/bin/bash
# Round-4 first look (VERDICT r03 "Next round" #1, #5 and the inputs for #2): the two sharded configs that had never run at
# full size, the phase split at n = 28 / 30 / 32, the n = 33 / P = 8 local leg, and counters that say what k_tile waits for
# (clock under load, scalar-cache and instruction-cache hit rates).
tag=${1:-r04a}
out=$PWD/gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
rocprofv3 -L > $out/counters.txt 2>&1 || true
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -s -k "full_size_sharded or largest_sharded" > $out/pytest_sharded.log 2>&1; echo "pytest sharded exit $?" | tee -a $out/pytest_sharded.log
tail -3 $out/pytest_sharded.log
python3 bench.py --steps 5 --warmup 1 --wisdom $out/wisdom.txt --no-cpu-baseline --no-precision32 --no-one-shot --sizes= > $out/bench_n30.json 2> $out/bench_n30.err; echo "bench $?"
tail -c 600 $out/bench_n30.json
for n in 28 30 32; do
  timeout -k 10 400 python3 tools/phase_split.py $n > $out/phase_split_n$n.log 2>&1; echo "phase_split $n: $?"; tail -1 $out/phase_split_n$n.log
done
timeout -k 10 500 python3 tools/cluster_bench.py 33 model 8 > $out/cluster_bench_n33.log 2>&1; echo "cluster_bench n33: $?"; tail -2 $out/cluster_bench_n33.log
cd /tmp
B="python3 $OLDPWD/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-full-sweeps --no-precision32 --no-one-shot --sizes= --wisdom $out/wisdom.txt"
rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT --kernel-trace --output-format csv -d $out/pmc_clk -o t -- $B > $out/pmc_clk.log 2>&1 || echo "fail clk"
rocprofv3 --pmc SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES --output-format csv -d $out/pmc_sqc -o t -- $B > $out/pmc_sqc.log 2>&1 || echo "fail sqc"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA --output-format csv -d $out/pmc_sq1 -o t -- $B > $out/pmc_sq1.log 2>&1 || echo "fail sq1"
rocprofv3 --pmc SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_IFETCH SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_LDS SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $out/pmc_sq2 -o t -- $B > $out/pmc_sq2.log 2>&1 || echo "fail sq2"
cd $OLDPWD
python3 - <<PY
import csv, collections, glob
for d in ("pmc_clk", "pmc_sqc", "pmc_sq1", "pmc_sq2"):
    for f in glob.glob(f"$out/{d}/*counter_collection.csv"):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "k_tile" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            print(f"{d:8s} {k:28s} n={len(v):3d} mean={sum(v)/len(v):18.1f}  max={max(v):18.1f}")
    for f in glob.glob(f"$out/{d}/*kernel_trace.csv"):
        rows = [r for r in csv.DictReader(open(f)) if "k_tile" in r["Kernel_Name"]]
        for r in rows:
            print(d, "dispatch", r.get("Dispatch_Id"), "ns", int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
PY
find $out -name "*.csv" -size +8M -delete
