"""GPU busy time from a rocprofv3 kernel trace CSV: union of [start, end) over all kernels vs the span.  Usage: trace_gaps.py file.csv"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows if "qsim" in r["Kernel_Name"])
# last third of the trace = the steady-state iterations
t_lo = iv[0][0] + 2 * (iv[-1][1] - iv[0][0]) // 3
iv = [x for x in iv if x[0] >= t_lo]
busy, cur_s, cur_e = 0, iv[0][0], iv[0][1]
for s, e, _ in iv[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
span = iv[-1][1] - iv[0][0]
tot = sum(e - s for s, e, _ in iv)
print(f"kernels {len(iv)} span {span/1e6:.2f} ms busy(union) {busy/1e6:.2f} ms ({100*busy/span:.1f} %) sum of durations {tot/1e6:.2f} ms (avg concurrency {tot/busy:.2f})")
