#!/bin/bash
# GPU parity suite + the phase split of the n=30 bench schedule (the yardstick of the block-phase work)
tag=${1:-r04t}; out=gpurun_out/$tag; mkdir -p $out
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $out/pytest_gpu.log 2>&1; rc=$?; echo "pytest exit $rc" >> $out/pytest_gpu.log
tail -5 $out/pytest_gpu.log
python3 tools/phase_split.py 30 > $out/phase_split_n30.log 2>&1; tail -9 $out/phase_split_n30.log
[ $rc -ne 0 ] && exit 1
exit 0
