#!/bin/bash
out=gpurun_out/${1:-r02j}; mkdir -p $out
for n in 24 26; do for cap in 0 2048 1024 512; do python tools/phase_split.py $n grid_cap=$cap > $out/phase_n${n}_cap$cap.log 2>&1; echo "n=$n cap=$cap $(tail -1 $out/phase_n${n}_cap$cap.log)"; done; done
for n in 20 22; do for cap in 0 512 256; do python tools/phase_split.py $n grid_cap=$cap > $out/phase_n${n}_cap$cap.log 2>&1; echo "n=$n cap=$cap $(tail -1 $out/phase_n${n}_cap$cap.log)"; done; done
