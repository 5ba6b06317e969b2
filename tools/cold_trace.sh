#!/bin/bash
# One cold run of bin/qsim (n = 30, the bench circuit) under rocprofv3's HIP API + kernel trace, then five plain runs with QSIM_STATS:
# where the printed seconds of the reference's own protocol go (context, queue creation, hipMalloc, first copy, code object, passes).
# Usage (on the GPU box): bash tools/cold_trace.sh <tag>      -> gpurun_out/cold_<tag>/
set -e
TAG=${1:-r04}
OUT=gpurun_out/cold_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 -c "
import sys; sys.path.insert(0,'.')
from gpu_quantum_simulator_amd import circuits
circuits.write_qasm('/tmp/c30.qasm',30,circuits.random_gates(30,1000,20240117+30,'all'))
"
for i in 1 2 3 4 5; do QSIM_STATS=1 gpu_quantum_simulator_amd/bin/qsim /tmp/c30.qasm 1 2>> $OUT/stats.jsonl >> $OUT/printed.txt; done
rocprofv3 --hip-trace --kernel-trace --output-format csv -d $OUT/trace -- gpu_quantum_simulator_amd/bin/qsim /tmp/c30.qasm 1
python3 tools/cold_timeline.py $OUT/trace > $OUT/timeline.txt
cat $OUT/printed.txt $OUT/timeline.txt
