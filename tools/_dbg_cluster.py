import sys
sys.path.insert(0,'.')
from gpu_quantum_simulator_amd import Circuit, Cluster, circuits
n=18
stale = Circuit.from_gates(n, circuits.random_gates(n, 500, 7, "all"))
c = Circuit.from_gates(n, circuits.random_gates(n, 700, 303, "all"))
with Cluster(n, 2, devices=[0,0]) as cl:
    print(cl.exchange_stats(), cl.exchange_bytes_moved())
    cl.run(stale); print(cl.exchange_stats(), cl.exchange_bytes_moved())
    cl.run(stale); print(cl.exchange_stats(), cl.exchange_bytes_moved())
    cl.run(c); print(cl.exchange_stats(), cl.exchange_bytes_moved(), cl.exchange_mode)
