"""Where a tile pass spends its time (n=30 bench schedule): every pass timed three ways — as it runs, with its blocks
skipped (QSIM_OPT_DEBUG_SKIP_OPS: HBM -> LDS -> HBM only) and with its memory traffic skipped (QSIM_OPT_DEBUG_SKIP_MEM:
blocks on zero tiles only).  full ~ max(mem, ops) means the two overlap; full ~ mem + ops means they serialise."""
import sys
sys.path.insert(0, '.')
from gpu_quantum_simulator_amd import Circuit, Simulator, circuits, _lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
opts = dict(kv.split("=") for kv in sys.argv[2:])
opts = {k: int(v) for k, v in opts.items()}
precision = opts.pop("precision", 64)
c = Circuit.from_gates(n, circuits.random_gates(n, 1000, 20240117 + n, "all"))


def log(sim):
    def body():
        sim.reset(); sim.run(c); sim.flush()
    body(); sim.sync(); sim.reset_stats()
    body(); sim.sync()
    return sim.launch_log()


with Simulator(n, fuse=3, profile=True, precision=precision, **opts) as sim:
    sim.choose_schedule(c)  # the schedule bench.py's planning step picks for this circuit
    full = log(sim)
    sim.set_option(_lib.OPT_DEBUG_SKIP_OPS, 1)
    mem = log(sim)
    sim.set_option(_lib.OPT_DEBUG_SKIP_OPS, 0)
    sim.set_option(_lib.OPT_DEBUG_SKIP_MEM, 1)
    ops = log(sim)
    sim.set_option(_lib.OPT_DEBUG_SKIP_MEM, 0)
    t = [0.0, 0.0, 0.0]
    for (k, nops, hm, ms), (_, _, _, m0), (_, _, _, o0) in zip(full, mem, ops):
        bits = [b for b in range(40) if hm >> b & 1]
        print(f"{k:6s} blocks={nops:2d} full={ms:7.3f} mem_only={m0:7.3f} ops_only={o0:7.3f} high={bits}", flush=True)
        t[0] += ms; t[1] += m0; t[2] += o0
    print(f"total full={t[0]:.2f} mem_only={t[1]:.2f} ops_only={t[2]:.2f}  passes={len(full)}", flush=True)
