#!/bin/bash
# rocprofv3 passes over one long-atom encode (scripts/long_atom_one.py): kernel trace, then counters in their own passes.
set -uo pipefail
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
N=${1:-32768}; B=${2:-8}
OUT="$PWD/gpurun_out/long_atom_$N"; mkdir -p "$OUT"
CMD="python3 scripts/long_atom_one.py $N $B"
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d "$OUT/kt" --output-format csv -- $CMD > "$OUT/kt.log" 2>&1; echo "kt rc=$?"
timeout -k 10 200 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY -d "$OUT/sq" --output-format csv -- $CMD > "$OUT/sq.log" 2>&1; echo "sq rc=$?"
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE -d "$OUT/fetch" --output-format csv -- $CMD > "$OUT/fetch.log" 2>&1; echo "fetch rc=$?"
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM -d "$OUT/sq2" --output-format csv -- $CMD > "$OUT/sq2.log" 2>&1; echo "sq2 rc=$?"
find "$OUT" -name "*_agent_info.csv" -delete 2>/dev/null
python3 - "$OUT" <<'PY'
import csv, glob, collections, sys, numpy as np
out = sys.argv[1]
for f in glob.glob(out + "/kt/*/*_kernel_stats.csv"):
    for i, r in enumerate(csv.DictReader(open(f))):
        if i < 8: print(r["Name"][:90], r["Calls"], r["TotalDurationNs"], r["AverageNs"])
for tag in ("sq", "fetch", "sq2"):
    v = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(out + f"/{tag}/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "fft_screen" in k or "refine" in k or "select" in k:
                v[k.split("(")[0][-60:] + " grid " + r["Grid_Size"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, c in v.items():
        print(tag, k, {n: (len(x), round(float(np.mean(x)), 1)) for n, x in c.items()})
PY
