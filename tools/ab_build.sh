#!/bin/bash
# A/B builds of the kernel file with different compiler options: tools/ab/libqsim_<tag>.so (run with QSIM_LIB=..., tools/ab_run.sh)
# usage: tools/ab_build.sh tag "extra hipcc flags" [tag "flags" ...]
set -e
cd "$(dirname "$0")/../gpu_quantum_simulator_amd/csrc"
mkdir -p ../../tools/ab
make -s ARCH=gfx950
build() {
  tag=$1; shift
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-result $@ -c kernels.hip -o ../../tools/ab/kernels_$tag.o
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../../tools/ab/libqsim_$tag.so ../../tools/ab/kernels_$tag.o engine.o scheduler.o dist.o qasm.o legacy.o -L/opt/rocm/lib -lrccl -lm
  rm -f ../../tools/ab/kernels_$tag.o
  echo built $tag
}
while [ $# -gt 1 ]; do
  build "$1" $2 &
  shift 2
done
wait
