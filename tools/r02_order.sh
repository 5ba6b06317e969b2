#!/bin/bash
out=gpurun_out/${1:-r02q}; mkdir -p $out
python - > $out/order_parity.log 2>&1 <<'PY'
import sys; sys.path.insert(0,'.')
import numpy as np, tempfile, os
from gpu_quantum_simulator_amd import Circuit, Simulator, circuits
from oracle import oracle
for n,depth,k in ((16,400,3),(20,600,5),(13,300,9)):
    with tempfile.TemporaryDirectory() as d:
        path=circuits.random_circuit_file(os.path.join(d,'c.qasm'),n,depth,77+n,'all')
        _,want,_,_=oracle.run_qasm(path)
        c=Circuit.from_file(path)
        with Simulator(n,0,fuse=3,debug_tile_order=k) as sim:
            sim.run(c); got=sim.read()
            print(n, 'shuffled tile order: max err', float(np.max(np.abs(got-want))))
PY
tail -3 $out/order_parity.log
timeout -k 10 800 python tools/geom_probe4.py 21 3000 48 > $out/geom_probe4.log 2>&1; grep -v "part B" $out/geom_probe4.log | tail -45
