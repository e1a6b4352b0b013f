#!/usr/bin/env python3
"""Do sub-batches overlap?  Reads a rocprofv3 --kernel-trace CSV of scripts/overlap_run.py and prints, for the
steady state of one encode, a timeline of kernel begin / end per queue and how much of the select kernels' time ran
beside a screen kernel.     usage: overlap_trace.py <kernel_trace.csv> [first_row] [rows]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
first = int(sys.argv[2]) if len(sys.argv) > 2 else len(rows) // 2
n = int(sys.argv[3]) if len(sys.argv) > 3 else 40
t0 = int(rows[first]["Start_Timestamp"])
for r in rows[first:first + n]:
    name = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:34]
    print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:9.1f} -> {(int(r['End_Timestamp']) - t0) / 1e3:9.1f} us  q{r.get('Queue_Id', '?'):>3s}  {name}")
# overlap statistics over the second half of the trace
half = rows[len(rows) // 2:]
scr = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in half if "fft_screen" in r["Kernel_Name"]]
sel = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in half if "select" in r["Kernel_Name"] or "scan_refine" in r["Kernel_Name"]]
tot = sum(e - s for s, e in sel)
ov = 0
for s, e in sel:
    for a, b in scr:
        lo, hi = max(s, a), min(e, b)
        if hi > lo:
            ov += hi - lo
iv = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in half]
span = max(e for _, e in iv) - min(s for s, _ in iv)
busy_scr = sum(e - s for s, e in scr)
print(f"second half: span {span / 1e3:.1f} us, screen kernel time {busy_scr / 1e3:.1f} us, select kernel time {tot / 1e3:.1f} us, "
      f"of which beside a screen {ov / 1e3:.1f} us ({ov / max(tot, 1):.2f})")
