#!/bin/bash
# Round-2 profiling passes (run on the GPU box via gpurun; summaries are then copied into profiles/ by
# scripts/summarize_profiles.py / summarize_c4.py / summarize_stats.py):
#   bench.py on the FFT schedule (persistent default; launch per step) and on the incremental MFMA schedule (kernel
#   trace + separate PMC passes),
#   the config-4 shape (fft_screen_kernel<13>), and kernel-trace --stats of the local-contrast-norm schedule,
#   dictionary_learning_step and the config-5 train step.
set -uo pipefail
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
bash scripts/profile_round.sh r02_persist fft                    # the library default at the headline shape: persistent form
bash scripts/profile_round.sh r02_fft fft "--flags 4096"          # launch-per-step, one stream: the per-step screen kernel
bash scripts/profile_round.sh r02_inc incremental "--flags 4096"
OUT="$PWD/gpurun_out/r02_c4"; mkdir -p "$OUT"
C4="python3 scripts/c4_traffic.py 128 6"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$OUT/kt" --output-format csv -- $C4 > "$OUT/kt.log" 2>&1; echo "c4 kt rc=$?"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE -d "$OUT/fetch" --output-format csv -- $C4 > "$OUT/fetch.log" 2>&1; echo "c4 fetch rc=$?"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE -d "$OUT/write" --output-format csv -- $C4 > "$OUT/write.log" 2>&1; echo "c4 write rc=$?"
timeout -k 10 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY -d "$OUT/sq4" --output-format csv -- $C4 > "$OUT/sq4.log" 2>&1; echo "c4 sq rc=$?"
for w in lcn_time dls_time c5_step; do
  O="$PWD/gpurun_out/r02_$w"; mkdir -p "$O"
  timeout -k 10 400 rocprofv3 --kernel-trace --stats -d "$O/kt" --output-format csv -- python3 scripts/$w.py > "$O/run.log" 2>&1; echo "$w rc=$?"
  f=$(find "$O/kt" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$O/kernel_stats.csv"; rm -rf "$O/kt"
done
# keep only what the summaries need (the raw traces are tens of MB)
find gpurun_out/r02_persist gpurun_out/r02_fft gpurun_out/r02_inc gpurun_out/r02_c4 -name "*_agent_info.csv" -delete 2>/dev/null
du -sh gpurun_out/r02_* 2>/dev/null
