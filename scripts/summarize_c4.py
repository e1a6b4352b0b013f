#!/usr/bin/env python3
"""Condense the rocprofv3 passes of scripts/c4_traffic.py (gpurun_out/r01_c4/{fetch,write,kt,sq4}) into
profiles/r01_c4_summary.json."""
import csv, glob, collections, json, os, sys
import numpy as np
src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r01_c4"
dst = sys.argv[2] if len(sys.argv) > 2 else "profiles/r01_c4_summary.json"
new = lambda pat: sorted(glob.glob(pat), key=os.path.getmtime)[-1]
out = {"what": sys.argv[3] if len(sys.argv) > 3 else
       "BASELINE configs[3] shape (4096 x 2048 dictionary, 128 x 131072-sample segments), MP_PATH_FFT, one stream, "
       "K=6: scripts/c4_traffic.py under rocprofv3 (separate --pmc passes and a --kernel-trace --stats pass)"}
for tag, cn in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    v = collections.defaultdict(list)
    for r in csv.DictReader(open(new(f"{src}/{tag}/*/*_counter_collection.csv"))):
        if r["Counter_Name"] == cn and "fft_screen_kernel" in r["Kernel_Name"]:
            v["full_pass" if int(r["Grid_Size"]) > 1e8 else "incremental"].append(float(r["Counter_Value"]))
    out[cn + "_KB_fft_screen_kernel"] = {k: {"launches": len(x), "avg_per_launch": round(float(np.mean(x)), 1)} for k, x in v.items()}
d = collections.defaultdict(list)
for r in csv.DictReader(open(new(f"{src}/kt/*/*_kernel_trace.csv"))):
    n = r["Kernel_Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0]
    if "fft_screen" in n:
        n += " [full pass]" if int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]) > 1e8 else " [incremental]"
    d[n].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
out["kernel_trace"] = [{"kernel": k, "calls": len(v), "avg_us": round(float(np.mean(v)) / 1e3, 2)}
                       for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1]))[:6]]
f, w = (out[c + "_KB_fft_screen_kernel"]["incremental"]["avg_per_launch"] for c in ("FETCH_SIZE", "WRITE_SIZE"))
out["hbm_traffic_bytes_per_incremental_launch_fft_screen_kernel"] = int((2 * f + w) * 1024)  # FETCH_SIZE x2: gfx950 correction
out["algorithmic_bytes_per_incremental_launch"] = int(128 * (4096 // 2 + 1) * 8 * 8192)
v = collections.defaultdict(lambda: collections.defaultdict(list)); dur = collections.defaultdict(list)
for r in csv.DictReader(open(new(f"{src}/sq4/*/*_counter_collection.csv"))):
    if "fft_screen_kernel" in r["Kernel_Name"]:
        kind = "full_pass" if int(r["Grid_Size"]) > 1e8 else "incremental"
        v[kind][r["Counter_Name"]].append(float(r["Counter_Value"]))
        dur[kind].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
out["sq_counters_fft_screen_kernel"] = {}
for kind in v:
    x = {k: float(np.mean(y)) for k, y in v[kind].items()}
    cyc = x["GRBM_GUI_ACTIVE"] / 8
    out["sq_counters_fft_screen_kernel"][kind] = {
        "avg_ms": round(float(np.mean(dur[kind])) / 1e6, 3), "clock_GHz": round(cyc / float(np.mean(dur[kind])), 2),
        "valu_busy_frac_of_simd_cycles": round(x["SQ_ACTIVE_INST_VALU"] * 4 / (cyc * 1024), 3),
        "lds_busy_frac_of_cu_cycles": round(x["SQ_LDS_IDX_ACTIVE"] / (cyc * 256), 3),
        "lds_bank_conflict_frac_of_lds_cycles": round(x["SQ_LDS_BANK_CONFLICT"] / x["SQ_LDS_IDX_ACTIVE"], 3),
        "waves_per_simd": round(x["SQ_WAVE_CYCLES"] * 4 / (cyc * 1024), 2)}
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps(out, indent=1))
