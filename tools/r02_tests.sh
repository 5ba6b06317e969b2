#!/bin/bash
# the whole -m gpu suite + smoke + a default bench line
out=gpurun_out/${1:-r02f}; mkdir -p $out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $out/pytest_gpu.log 2>&1; rc=$?; echo "pytest exit $rc" >> $out/pytest_gpu.log
tail -15 $out/pytest_gpu.log
[ $rc -ne 0 ] && exit 1
python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; echo "smoke $?"; tail -2 $out/smoke.log
python tools/cluster_bench.py > $out/cluster_bench.log 2>&1; echo "cluster bench $?"; tail -6 $out/cluster_bench.log
python bench.py --steps 3 --warmup 1 > $out/bench.json 2> $out/bench.err; echo "bench $?"; tail -c 3000 $out/bench.json
