#!/bin/bash
# Why does the config-4 screen not get faster in proportion to the tiles the lazy screen skips when whole SEGMENTS skip?
# Two forced-mask runs with the same share of workgroups skipped (MP_TUNE_LAZY_FORCE: 1.5 = every (segment, tile) with
# probability 0.5; 2.5 = every segment all its tiles with probability 0.5) under rocprofv3: kernel trace, SQ and TCC counters.
set -uo pipefail
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
for f in 1.5 2.5; do
  OUT="$PWD/gpurun_out/c4_mode_$f"; mkdir -p "$OUT"
  export C4_FORCE=$f
  C4="python3 scripts/c4_traffic.py 128 12"
  timeout -k 10 200 rocprofv3 --kernel-trace -d "$OUT/kt" --output-format csv -- $C4 > "$OUT/kt.log" 2>&1; echo "force $f kt rc=$?"
  timeout -k 10 200 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY -d "$OUT/sq" --output-format csv -- $C4 > "$OUT/sq.log" 2>&1; echo "force $f sq rc=$?"
  timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE -d "$OUT/fetch" --output-format csv -- $C4 > "$OUT/fetch.log" 2>&1; echo "force $f fetch rc=$?"
  timeout -k 10 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum -d "$OUT/tcc" --output-format csv -- $C4 > "$OUT/tcc.log" 2>&1; echo "force $f tcc rc=$?"
  find "$OUT" -name "*_agent_info.csv" -delete 2>/dev/null
done
python3 - <<'PY'
import csv, glob, collections, os
import numpy as np
for f in ("1.5", "2.5"):
    src = f"gpurun_out/c4_mode_{f}"
    print("== force", f)
    for tag in ("kt", "sq", "fetch", "tcc"):
        files = glob.glob(f"{src}/{tag}/*/*_kernel_trace.csv" if tag == "kt" else f"{src}/{tag}/*/*_counter_collection.csv")
        if not files:
            print("  ", tag, "no output"); continue
        rows = list(csv.DictReader(open(files[0])))
        if tag == "kt":
            d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if "fft_screen_kernel" in r["Kernel_Name"]]
            print("   screen launches (us):", [round(v) for v in d])
        else:
            v = collections.defaultdict(list)
            for r in rows:
                if "fft_screen_kernel" in r["Kernel_Name"]:
                    v[r["Counter_Name"]].append(float(r["Counter_Value"]))
            for k, x in v.items():
                print(f"   {k}: first launches {[round(t) for t in x[:3]]} ... masked launches mean {np.mean(x[3:]):.4g}")
PY
