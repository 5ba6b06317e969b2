"""Overhead of the sharded path itself, links excluded: P virtual shards on ONE device (exchanges = HBM copies)."""
import sys, time
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from gpu_quantum_simulator_amd import Circuit, Cluster, Simulator, circuits
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
MEASURED = len(sys.argv) > 2 and sys.argv[2] == "measured"  # planning with timing (qsim_tune_circuit / qsim_cluster_plan with candidates) instead of the model alone
SHARDS = tuple(int(x) for x in sys.argv[3].split(",")) if len(sys.argv) > 3 else (2, 4, 8)  # e.g. `33 model 8`: configs[4]'s local leg
c = Circuit.from_gates(n, circuits.random_gates(n, 1000, 20240117 + n, "all"))
with Simulator(n, **({"pingpong": 0} if n >= 33 else {})) as sim:  # n = 33: 128 GiB in place, the pools of the cluster need the rest
    sim.choose_schedule(c)
    if MEASURED:
        sim.tune(c, 16, 6000.0)
    def body():
        sim.reset(); sim.run(c); sim.sync()
    body()
    t0 = time.perf_counter(); body(); body(); dt = (time.perf_counter() - t0) / 2
    print(f"single      : {dt*1e3:8.2f} ms/iter  launches/iter={sim.stats()['launches']//3}", flush=True)
for P in SHARDS:
    with Cluster(n, P, devices=[0] * P) as cl:
        cl.plan(c, 16 if MEASURED else 1, 12000.0)  # the schedule choice per shard and local step, like the single state above (outside the clock)
        cl.run(c)
        t0 = time.perf_counter(); cl.run(c); cl.run(c); dt = (time.perf_counter() - t0) / 2
        ex, nb = cl.exchange_stats()
        print(f"{P} virtual   : {dt*1e3:8.2f} ms/iter  exchanges/iter={ex//3}  GiB per shard/iter if every block travelled={nb/3/2**30:.2f}  "
              f"GiB really moved (all shards)/iter={cl.exchange_bytes_moved()/3/2**30:.2f}  packs fused/separate={cl.pack_counts()}  "
              f"local ms per shard and iter (the shards take turns on this one device)={dt*1e3/P:.2f}", flush=True)
