"""Markdown tables for DESIGN.md from a bench line (default profiles/r03/bench_n30.json): the exchange model (section 6) and the
current numbers (section 7).  Usage: python tools/model_table.py [bench.json]"""
import json, sys
path = sys.argv[1] if len(sys.argv) > 1 else "profiles/r03/bench_n30.json"
d = json.loads(open(path).read().strip().splitlines()[-1])
m = d["exchange_model"]
print("| config | exchanges (qubits) | local passes / sweeps of the shard | bytes sent by the busiest rank | exchange ms | local ms | step ms | ideal ms | vs ideal |")
print("|---|---|---|---|---|---|---|---|---|")
for r in m["configs"]:
    print(f"| n = {r['qubits']}, P = {r['ranks']} | {r['exchanges']} ({', '.join(map(str, r['qubits_swapped']))}) | {r['local_passes']} / {r['local_sweeps_of_the_shard']:.2f} | "
          f"{r['bytes_sent_per_rank'] / 2**30:.1f} GiB | {r['predicted_exchange_ms']:.1f} | {r['predicted_local_ms']:.1f} | {r['predicted_step_ms']:.1f} | {r['ideal_ms']:.1f} | {r['vs_ideal']:.2f} |")
print()
print("assumptions:", json.dumps(m["assumptions"]))
print()
fs = d["sparse_start"]["with_full_sweeps"]
print("| n | workload | gate-applies/s | ms / step | launches | with full sweeps | k_tile of HBM peak | CPU (1 core) |")
print("|---|---|---|---|---|---|---|---|")
print(f"| 30 | {d['config']['workload']} | **{d['value']:.0f}** | {d['ms_per_step']:.2f} | {d['launches_per_step']:.0f} | {fs['value']:.0f} ({fs['ms_per_step']:.1f} ms) | {d['roofline']['frac']:.3f} | {d['cpu_baseline']['value']:.3f} |")
for r in d["sizes"]:
    cb = r.get("cpu_baseline")
    f = r["with_full_sweeps"]
    print(f"| {r['qubits']} | {r['workload']} | {r['value']:.0f} | {r['ms_per_step']:.2f} | {r['launches_per_step']:.0f} | {f['value']:.0f} ({f['ms_per_step']:.1f} ms) | {r['roofline']['frac']:.3f} | {(str(round(cb['value'], 3))) if cb else '-'} |")
p = d["precision32"]
print(f"| 30 fp32 | {p['workload']} | {p['value']:.0f} | {p['ms_per_step']:.2f} | {p['launches_per_step']:.0f} | {p['with_full_sweeps']['value']:.0f} ({p['with_full_sweeps']['ms_per_step']:.1f} ms) | {p['roofline']['frac']:.3f} | - |")
o = d["one_shot"]
print()
print("one shot:", json.dumps(o["cli"]), "first in-process step", o["in_process_first_step_ms"], "second", o["in_process_second_step_ms"])
