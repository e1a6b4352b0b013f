#!/usr/bin/env python3
"""Condense rocprofv3 output directories (gpurun_out/<round>/{kt,fetch,write,sq}) into the small,
tracked summaries under profiles/.   usage: summarize_profiles.py gpurun_out/r01 profiles/r01"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

src, dst = sys.argv[1], sys.argv[2]
os.makedirs(os.path.dirname(dst) or ".", exist_ok=True)


def newest(pattern):
    """gpurun merges every call's files into gpurun_out/: keep the latest run of each pass only"""
    return sorted(glob.glob(pattern), key=os.path.getmtime, reverse=True)[:1]


def short(name):
    for key in ("fft_persistent_kernel", "persist_mark_kernel", "clear_words_kernel", "correlate_persistent_kernel", "correlate_mfma_kernel", "correlate_naive_kernel",
                "fft_screen_kernel", "fft_correlate_kernel", "fft_refine_chain_kernel", "fft_refine_valu_kernel",
                "fft_refine_kernel", "fft_scan_refine_kernel", "fft_select_a_kernel", "fft_select_b_kernel", "fft_select_fused_kernel", "fft_select_quarter_kernel", "fft_window_kernel",
                "fft_dict_kernel", "fft_twiddle_kernel", "fft_mark_overflow_kernel",
                "select_subtract_kernel", "unit_norm_kernel", "init_residual_kernel", "copy_residual_kernel",
                "dict_image_kernel"):
        if key in name:
            tpl = name[name.index(key):].split("(")[0]
            return tpl
    return name[:60]


summary = {}
ks = newest(os.path.join(src, "kt", "*", "*_kernel_stats.csv"))
if ks:
    shutil.copy(ks[0], dst + "_kernel_stats.csv")
    rows = list(csv.DictReader(open(ks[0])))
    summary["kernel_trace_stats"] = [
        {"kernel": short(r["Name"]), "calls": int(r["Calls"]), "avg_us": round(float(r["AverageNs"]) / 1e3, 3),
         "total_ms": round(float(r["TotalDurationNs"]) / 1e6, 3), "pct": float(r["Percentage"])}
        for r in rows[:8]]
    b = os.path.join(src, "bench_under_rocprof_kt.json")
    if os.path.exists(b):
        summary["bench_line_under_kernel_trace"] = json.loads(open(b).read().strip().splitlines()[-1])

for tag, cn in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    fs = newest(os.path.join(src, tag, "*", "*_counter_collection.csv"))
    if not fs:
        continue
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(fs[0])):
        if r["Counter_Name"] != cn:
            continue
        k = short(r["Kernel_Name"])
        agg[k][0] += 1
        agg[k][1] += float(r["Counter_Value"])
    summary[cn + "_KB"] = {k: {"launches": v[0], "avg_per_launch": round(v[1] / v[0], 2)} for k, v in agg.items()}

fs = newest(os.path.join(src, "sq", "*", "*_counter_collection.csv"))
fs2 = newest(os.path.join(src, "sq2", "*", "*_counter_collection.csv"))  # second pass: VALU / LDS counters
if fs:
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    dur = collections.defaultdict(float)
    n = collections.defaultdict(int)
    seen = set()
    for which, path in enumerate(fs + fs2):
        for r in csv.DictReader(open(path)):
            if not any(k in r["Kernel_Name"] for k in ("correlate", "fft_screen", "fft_persistent")):
                continue
            d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            kind = "full_pass" if d > (5e5 if "fft_screen" in r["Kernel_Name"] else 3e6) else "incremental"
            if "fft_persistent" in r["Kernel_Name"]:
                kind = "persistent_launch"
            agg[kind][r["Counter_Name"]] += float(r["Counter_Value"])
            if which == 0 and r["Dispatch_Id"] not in seen:
                seen.add(r["Dispatch_Id"])
                dur[kind] += d
                n[kind] += 1
    sq = {}
    for kind, v in agg.items():
        cyc = v["GRBM_GUI_ACTIVE"] / 8  # summed over the 8 XCDs
        sq[kind] = {
            "launches": n[kind], "avg_us": round(dur[kind] / n[kind] / 1e3, 2),
            "clock_GHz": round(cyc / dur[kind], 3),
            "mfma_pipe_busy_frac": round(v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (cyc * 1024), 4),  # 256 CUs x 4 SIMDs
            "mfma_instructions": v.get("SQ_INSTS_VALU_MFMA_F32", 0.0),
            # SQ_ACTIVE_INST_VALU counts quad-cycles summed over the 1024 SIMDs; SQ_LDS_IDX_ACTIVE cycles over 256 CUs
            "valu_busy_frac_of_simd_cycles": round(v.get("SQ_ACTIVE_INST_VALU", 0.0) * 4 / (cyc * 1024), 4),
            "valu_wave_instructions": v.get("SQ_INSTS_VALU", 0.0) / max(n[kind], 1),
            "lds_busy_frac_of_cu_cycles": round(v.get("SQ_LDS_IDX_ACTIVE", 0.0) / (cyc * 256), 4),
            "lds_bank_conflict_frac_of_lds_cycles": round(v.get("SQ_LDS_BANK_CONFLICT", 0.0) / max(v.get("SQ_LDS_IDX_ACTIVE", 0.0), 1.0), 4),
            "waves_per_simd": round(v["SQ_WAVE_CYCLES"] * 4 / (cyc * 1024), 3),
            "wait_any_frac_of_wave_cycles": round(v["SQ_WAIT_ANY"] / v["SQ_WAVE_CYCLES"], 4),
            "wait_inst_any_frac_of_wave_cycles": round(v["SQ_WAIT_INST_ANY"] / v["SQ_WAVE_CYCLES"], 4),
        }
    summary["sq_counters_dominant_kernel"] = sq

# HBM traffic per launch of the dominant kernel, corrected as MI355X_MICROARCH.md prescribes:
# FETCH_SIZE (KB) is doubled on gfx950 for wide coalesced reads; WRITE_SIZE is exact.
fk = summary.get("FETCH_SIZE_KB", {})
wk = summary.get("WRITE_SIZE_KB", {})
for k in fk:
    if ("correlate" in k or "fft_screen" in k or "fft_persistent" in k) and k in wk:
        summary["hbm_traffic_bytes_per_launch_" + k.split("<")[0]] = int(
            (2 * fk[k]["avg_per_launch"] + wk[k]["avg_per_launch"]) * 1024)
json.dump(summary, open(dst + "_summary.json", "w"), indent=1)
print(json.dumps(summary, indent=1)[:3000])
