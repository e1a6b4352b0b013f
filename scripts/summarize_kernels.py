#!/usr/bin/env python3
"""Condense the passes of scripts/profile_round4.sh `passes` (gpurun_out/<dir>/{kt,fetch,write,sq,sq2}) into
profiles/<name>_summary.json + profiles/<name>_kernel_stats.csv: per kernel -- calls, average / min / max duration from the
kernel trace; from the counter passes HBM bytes per launch (FETCH_SIZE x 2 + WRITE_SIZE, KB -> bytes: the gfx950 correction of
MI355X_MICROARCH.md), VALU-busy / LDS-busy / MFMA-busy fractions, waves per SIMD, wait fractions, measured clock.
    usage: summarize_kernels.py gpurun_out/r04_lcn profiles/r04_lcn "what was profiled" [kernel substrings ...]"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

src, dst, what = sys.argv[1], sys.argv[2], sys.argv[3]
keep = sys.argv[4:]


def newest(pattern):
    fs = sorted(glob.glob(pattern), key=os.path.getmtime)
    return fs[-1] if fs else None


def short(n):
    return n.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0]


out = {"what": what, "kernels": {}}
ks = newest(os.path.join(src, "kt", "*", "*_kernel_stats.csv"))
if ks:
    shutil.copy(ks, dst + "_kernel_stats.csv")
    for r in csv.DictReader(open(ks)):
        n = short(r["Name"])
        if keep and not any(k in n for k in keep):
            continue
        out["kernels"][n] = {"calls": int(r["Calls"]), "avg_us": round(float(r["AverageNs"]) / 1e3, 2),
                             "min_us": round(float(r["MinNs"]) / 1e3, 2), "max_us": round(float(r["MaxNs"]) / 1e3, 2),
                             "pct_of_gpu_time": float(r["Percentage"])}
# a kernel launched once per encode over everything and then per step over what an event dirtied (full pass / incremental)
# is two populations: split at the geometric mean of its shortest and longest launch when they are 4 x apart
split_at = {}
kt = newest(os.path.join(src, "kt", "*", "*_kernel_trace.csv"))
if kt:
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(kt)):
        n = short(r["Kernel_Name"])
        if n in out["kernels"]:
            per[n].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    for n, v in per.items():
        if len(v) >= 2 and max(v) > 4 * min(v):
            thr = (min(v) * max(v)) ** 0.5
            split_at[n] = thr
            base = out["kernels"].pop(n)
            for tag, sel in (("full pass", [x for x in v if x > thr]), ("incremental", [x for x in v if x <= thr])):
                out["kernels"][f"{n} [{tag}]"] = {"calls": len(sel), "avg_us": round(sum(sel) / len(sel) / 1e3, 2),
                                                  "min_us": round(min(sel) / 1e3, 2), "max_us": round(max(sel) / 1e3, 2),
                                                  "pct_of_gpu_time": round(base["pct_of_gpu_time"] * sum(sel) / sum(v), 2)}


def classed(n, d):
    if n in split_at:
        return f"{n} [{'full pass' if d > split_at[n] else 'incremental'}]"
    return n


ctr = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for tag in ("fetch", "write", "sq", "sq2"):
    f = newest(os.path.join(src, tag, "*", "*_counter_collection.csv"))
    if not f:
        continue
    seen = set()
    for r in csv.DictReader(open(f)):
        n = classed(short(r["Kernel_Name"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        if n not in out["kernels"]:
            continue
        ctr[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if tag == "sq" and r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"])
            dur[n].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for n, c in ctr.items():
    m = {k: sum(v) / len(v) for k, v in c.items()}
    k = out["kernels"][n]
    if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
        k["hbm_bytes_per_launch"] = int((2 * m["FETCH_SIZE"] + m["WRITE_SIZE"]) * 1024)
        k["fetch_kb_raw"], k["write_kb"] = round(m["FETCH_SIZE"], 1), round(m["WRITE_SIZE"], 1)
    if "GRBM_GUI_ACTIVE" in m and dur.get(n):
        cyc = m["GRBM_GUI_ACTIVE"] / 8          # summed over the 8 XCDs
        d = sum(dur[n]) / len(dur[n])
        k["avg_us_under_counters"] = round(d / 1e3, 2)
        k["clock_GHz"] = round(cyc / d, 3)
        if "SQ_ACTIVE_INST_VALU" in m:
            k["valu_busy_frac_of_simd_cycles"] = round(m["SQ_ACTIVE_INST_VALU"] * 4 / (cyc * 1024), 4)
            k["valu_wave_instructions_per_launch"] = round(m.get("SQ_INSTS_VALU", 0.0))
        if "SQ_LDS_IDX_ACTIVE" in m:
            k["lds_busy_frac_of_cu_cycles"] = round(m["SQ_LDS_IDX_ACTIVE"] / (cyc * 256), 4)
            k["lds_bank_conflict_frac_of_lds_cycles"] = round(m.get("SQ_LDS_BANK_CONFLICT", 0.0) / max(m["SQ_LDS_IDX_ACTIVE"], 1.0), 4)
        if "SQ_WAVE_CYCLES" in m:
            k["waves_per_simd"] = round(m["SQ_WAVE_CYCLES"] * 4 / (cyc * 1024), 3)
            k["wait_any_frac_of_wave_cycles"] = round(m.get("SQ_WAIT_ANY", 0.0) / m["SQ_WAVE_CYCLES"], 4)
            k["wait_inst_any_frac_of_wave_cycles"] = round(m.get("SQ_WAIT_INST_ANY", 0.0) / m["SQ_WAVE_CYCLES"], 4)
        if "SQ_VALU_MFMA_BUSY_CYCLES" in m:
            k["mfma_pipe_busy_frac"] = round(m["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024), 4)
            k["mfma_instructions_per_launch"] = round(m.get("SQ_INSTS_VALU_MFMA_F32", 0.0))
json.dump(out, open(dst + "_summary.json", "w"), indent=1)
print(json.dumps(out, indent=1)[:4000])
