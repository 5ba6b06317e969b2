"""Where a workgroup's block phase spends its cycles: the QSIM_EXP_STAMP build (tools/ab_build.sh stamp "-DQSIM_EXP_STAMP",
run with QSIM_LIB=tools/ab/libqsim_stamp.so) stamps the shader clock at fixed points of every block of every tile pass of the
n=30 bench schedule, blocks only (QSIM_OPT_DEBUG_SKIP_MEM); this prints the average cycles per segment by block form.  The stamps
serialise what the real kernel overlaps: shares, not sums."""
import sys, collections
sys.path.insert(0, '.')
import numpy as np
from gpu_quantum_simulator_amd import Circuit, Simulator, circuits, _lib
n = 30
c = Circuit.from_gates(n, circuits.random_gates(n, 1000, 20240117 + n, "all"))
with Simulator(n, fuse=3, pingpong=0) as sim:
    sim.choose_schedule(c)
    sim.set_option(_lib.OPT_DEBUG_SKIP_MEM, 1)
    for _ in range(2):
        sim.reset(); sim.run(c); sim.flush(); sim.sync()
    names = ["switch", "smem A", "lds reads", "fma class0", "smem B", "fma class1", "mid barrier", "lds writes", "prepare next", "end barrier"]
    agg = collections.defaultdict(lambda: [np.zeros(10), 0])
    clocks = []
    for region in range(64):
        raw = sim.read(region << 20, 1 << 20).view(np.uint64).reshape(2048, 1024)  # 2048 workgroups x 1024 stamps
        ok = raw[:, 1021] == 0x51534D5453544D50
        if not ok.any():
            continue
        wg = raw[ok]
        nops = int(wg[0, 1020])
        per_pass = collections.defaultdict(lambda: [np.zeros(9), 0])
        for b in range(min(nops, 32)):
            st = wg[:, 32 * b: 32 * b + 16].astype(np.int64)
            form = st[:, 15]
            valid = (st[:, 9] > st[:, 0]) & (st[:, 1] > 0)
            for f in np.unique(form[valid]):
                m = valid & (form == f)
                info = int(f)
                K, T, flags = (info >> 8) & 255, 1 << ((info >> 1) & 3), (info >> 3) & 3  # flags: bit 0 skips, bit 1 barrier between reads and writes
                key = (K, T, flags)
                if T != 4:
                    d = (st[m, 9] - st[m, 0]).mean()
                    agg[key][0][9] += d * m.sum(); agg[key][1] += m.sum()
                    continue
                order = [0, 1, 2, 3, 4, 5, 6, 7, 8, 10, 9]
                seg = np.diff(st[m][:, order], axis=1).mean(axis=0)
                agg[key][0] += seg * m.sum(); agg[key][1] += m.sum()
        print(f"region {region}: {ok.sum()} workgroups, {nops} blocks", flush=True)
    for key, (tot, cnt) in sorted(agg.items()):
        seg = tot / max(cnt, 1)
        if key[1] == 4:
            print(f"K={key[0]} T=4 flags={key[2]} samples={cnt}: total {seg.sum():7.0f} cycles | " + " | ".join(f"{nm} {v:6.0f}" for nm, v in zip(names, seg)))
        else:
            print(f"K={key[0]} T={key[1]} flags={key[2]} samples={cnt}: total {seg[9]:7.0f} cycles")
