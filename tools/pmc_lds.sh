#!/bin/bash
# LDS bank conflicts of the tile kernel on the bench circuit for one precision: tools/pmc_lds.sh <tag> <64|32>
OUT=$PWD/gpurun_out/${1:-pmc_lds}; mkdir -p $OUT; export TMPDIR=/tmp; HERE=$PWD; cd /tmp
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_LDS --output-format csv -d $OUT/a -o t -- python3 $HERE/bench.py --precision ${2:-64} --steps 1 --warmup 0 --no-cpu-baseline --no-full-sweeps --no-precision32 --no-one-shot --sizes= --no-tune > $OUT/a.log 2>&1 || echo fail
cd $HERE
python3 - <<PY
import csv, collections, glob
for f in glob.glob("gpurun_out/${1:-pmc_lds}/a/*counter_collection.csv"):
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "k_tile" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in agg.items(): print(f"{k:28s} n={len(v):3d} mean={sum(v)/len(v):16.1f}")
PY
