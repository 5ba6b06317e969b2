#!/usr/bin/env python3
"""Timeline of one cold bin/qsim run from rocprofv3's hip_api_trace.csv + kernel_trace.csv (tools/cold_trace.sh): every HIP call
longer than 0.3 ms with its thread, then the kernels.  Times in ms from the first HIP call of the process."""
import csv, glob, os, sys

def main(root):
    api = kern = None
    for p in glob.glob(os.path.join(root, "**", "*_hip_api_trace.csv"), recursive=True): api = p
    for p in glob.glob(os.path.join(root, "**", "*_kernel_trace.csv"), recursive=True): kern = p
    rows = list(csv.DictReader(open(api)))
    t0 = min(int(r["Start_Timestamp"]) for r in rows)
    print("start_ms  dur_ms  call (thread)")
    for r in rows:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if e - s > 300000:
            print(f"{(s - t0) / 1e6:8.3f} {(e - s) / 1e6:7.3f}  {r['Function']} ({r['Thread_Id']})")
    tot = 0.0
    print("kernels")
    for r in csv.DictReader(open(kern)):
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        tot += (e - s) / 1e6
        print(f"{(s - t0) / 1e6:8.3f} {(e - s) / 1e6:7.3f}  {r['Kernel_Name'][:48]} grid={r.get('Grid_Size_X', r.get('Grid_Size', ''))}")
    print(f"kernel time {tot:.3f} ms")

if __name__ == "__main__":
    main(sys.argv[1])
