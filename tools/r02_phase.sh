#!/bin/bash
out=gpurun_out/${1:-r02d}; mkdir -p $out
python tools/phase_split.py 30 > $out/phase_owner.log 2>&1; echo "owner $?"; tail -1 $out/phase_owner.log
QSIM_LIB=$PWD/tools/ab/libqsim_rows.so python tools/phase_split.py 30 > $out/phase_rows.log 2>&1; echo "rows $?"; tail -1 $out/phase_rows.log
python tools/phase_split.py 30 tile_bits=12 tile_low_bits=4 > $out/phase_owner_l4.log 2>&1; tail -1 $out/phase_owner_l4.log
python tools/phase_split.py 30 tile_bits=11 tile_low_bits=3 > $out/phase_owner_b11.log 2>&1; tail -1 $out/phase_owner_b11.log
QSIM_SCHED_LOCAL=3 QSIM_SCHED_LOOKAHEAD=1 python tools/phase_split.py 30 > $out/phase_owner_local3.log 2>&1; tail -1 $out/phase_owner_local3.log
