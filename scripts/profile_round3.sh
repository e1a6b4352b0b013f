#!/bin/bash
# Round-3 profiling passes (run on the GPU box via gpurun; summaries are then copied into profiles/ by
# scripts/summarize_profiles.py / summarize_c4.py):
#   bench.py on the FFT schedule -- the persistent default and the launch-per-step form -- under rocprofv3 (kernel trace +
#   separate PMC passes), and the config-4 shape with the lazy screen of the launch-per-step form (fft_screen_kernel<13>,
#   fft_select_fused_kernel<13>): scripts/c4_traffic.py 128 segments x 256 steps, SURVEY 8(d)'s 768 planted events.
set -uo pipefail
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
if [ "${1:-all}" != "c4" ]; then
bash scripts/profile_round.sh r03_persist fft "--no-configs3"
bash scripts/profile_round.sh r03_fft fft "--flags 4096 --no-configs3"
fi
OUT="$PWD/gpurun_out/r03_c4"; mkdir -p "$OUT"
export C4_LAZY=1 C4_EVENTS=768
C4="python3 scripts/c4_traffic.py 128 256"   # (the whole job: the share of tiles skipped is ~90 % in the first steps and ~65 % later)
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$OUT/kt" --output-format csv -- $C4 > "$OUT/kt.log" 2>&1; echo "c4 kt rc=$?"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE -d "$OUT/fetch" --output-format csv -- $C4 > "$OUT/fetch.log" 2>&1; echo "c4 fetch rc=$?"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE -d "$OUT/write" --output-format csv -- $C4 > "$OUT/write.log" 2>&1; echo "c4 write rc=$?"
timeout -k 10 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY -d "$OUT/sq4" --output-format csv -- $C4 > "$OUT/sq4.log" 2>&1; echo "c4 sq rc=$?"
find gpurun_out/r03_persist gpurun_out/r03_fft gpurun_out/r03_c4 -name "*_agent_info.csv" -delete 2>/dev/null
du -sh gpurun_out/r03_* 2>/dev/null
