#!/bin/bash
# phase split (tools/phase_split.py) of the n=30 bench schedule with every tools/ab/libqsim_*.so and the tree's own library:
# what the blocks-only time of a pass loses when one ingredient of the block phase is taken out (kernels_impl.inc QSIM_EXP_*)
out=gpurun_out/${1:-ab}; mkdir -p $out
for lib in "" tools/ab/libqsim_*.so; do
  tag=$(basename "${lib:-tree}" .so)
  QSIM_LIB=${lib:+$PWD/$lib} python3 tools/phase_split.py ${2:-30} > $out/phase_$tag.log 2>&1
  echo "$tag: $(tail -1 $out/phase_$tag.log)"
  grep "blocks= 6" $out/phase_$tag.log | head -3
done
