"""Memory time vs compute time of tile passes (n=30): the bench schedule with and without its blocks applied
(QSIM_OPT_DEBUG_SKIP_OPS), then synthetic passes over chosen high-qubit sets."""
import sys
sys.path.insert(0, '.')
from gpu_quantum_simulator_amd import Circuit, Simulator, circuits
n = 30
B, L, T = (int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "12,3,512").split(','))
c = Circuit.from_gates(n, circuits.random_gates(n, 1000, 20240117 + n, "all"))


def log(sim, circ):
    def body():
        sim.reset(); sim.run(circ); sim.flush()
    body(); sim.sync(); sim.reset_stats()
    body(); sim.sync()
    return sim.launch_log()


with Simulator(n, fuse=3, profile=True, tile_bits=B, tile_low_bits=L, tile_threads=T, tile_max_ops=64) as sim:
    full = log(sim, c)
    sim.set_option(10, 1)
    mem = log(sim, c)
    sim.set_option(10, 0)
    tf = tm = 0.0
    for (k, nops, hm, ms), (_, _, _, ms0) in zip(full, mem):
        bits = [b for b in range(40) if hm >> b & 1]
        print(f"{k:6s} ops={nops:2d} full={ms:7.3f} mem_only={ms0:7.3f} high={bits}", flush=True)
        tf += ms; tm += ms0
    print(f"total full={tf:.2f} mem_only={tm:.2f}", flush=True)
    H = B - L
    sets = {"contiguous_low": list(range(L, L + H)), "contiguous_top": list(range(n - H, n)),
            "mid": list(range(10, 10 + H)), "spread": [L + 3 * i for i in range(H)],
            "two_halves": list(range(L, L + H // 2)) + list(range(n - (H - H // 2), n))}
    sim.set_option(10, 1)
    for name, qs in sets.items():
        # x on q[0] first so the pass after it is not a generating (write-only) pass
        others = [q for q in range(L, n) if q not in qs][:H]
        gates = [("h", q) for q in qs] + [("h", q) for q in others] + [("h", q) for q in qs]  # third pass = the probe
        cc = Circuit.from_gates(n, gates)
        ll = log(sim, cc)
        print(name, qs, " ".join(f"{ms:.3f}" for _, _, _, ms in ll), flush=True)
