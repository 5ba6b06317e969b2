#!/bin/bash
out=gpurun_out/${1:-r02h}; mkdir -p $out
for L in 4 5 6; do python tools/phase_split.py 30 tile_bits=12 tile_low_bits=$L > $out/phase_l$L.log 2>&1; echo "L=$L $(tail -1 $out/phase_l$L.log)"; done
for cap in 65536 16384 8192 4096; do python tools/phase_split.py 30 grid_cap=$cap > $out/phase_cap$cap.log 2>&1; echo "cap=$cap $(tail -1 $out/phase_cap$cap.log)"; done
