#!/bin/bash
# SQ counters of the tile kernel on the bench circuit (one rocprofv3 --pmc pass, kernel-trace only).
OUT=$PWD/gpurun_out/${1:-pmc_tile}; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/a -o t -- python3 $OLDPWD/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-full-sweeps --no-precision32 --no-one-shot --sizes= --wisdom $OLDPWD/gpurun_out/${2:-r03}/wisdom.txt > $OUT/a.log 2>&1 || echo fail a
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU --output-format csv -d $OUT/b -o t -- python3 $OLDPWD/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-full-sweeps --no-precision32 --no-one-shot --sizes= --wisdom $OLDPWD/gpurun_out/${2:-r03}/wisdom.txt > $OUT/b.log 2>&1 || echo fail b
cd $OLDPWD
python3 - <<PY
import csv, collections, glob
for d in ("a","b"):
    for f in glob.glob(f"gpurun_out/${1:-pmc_tile}/{d}/*counter_collection.csv"):
        rows=list(csv.DictReader(open(f)))
        agg=collections.defaultdict(list)
        for r in rows:
            if "k_tile" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k,v in agg.items():
            print(f"{k:28s} n={len(v):3d} mean={sum(v)/len(v):16.1f}")
PY
