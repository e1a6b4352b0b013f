#!/bin/bash
# Round-4 profiling passes (run on the GPU box via gpurun; summaries are then copied into profiles/ by
# scripts/summarize_profiles.py / summarize_c4.py / summarize_kernels.py):
#   which = persist | fft | c4 | long | lcn | longest | all
#   persist, fft : bench.py on the FFT schedule (the one-launch default; launch per step on one stream): kernel trace + PMC passes
#   c4           : BASELINE configs[3] with the lazy screen (scripts/c4_traffic.py 128 256, SURVEY 8(d)'s 768 planted events)
#   long         : the split-transform screen, 1024 atoms of 8192 samples, 8 x 32768 (scripts/long_atom_one.py)
#   lcn          : the local-contrast-norm schedule at the headline shape (scripts/lcn_headline.py)
#   longest      : the four-quarter screen, 2048 atoms of 16384 samples, 4 x 32768, 32 steps (scripts/longest_atoms_time.py)
# Counters run in their own passes (--pmc alone), never with a trace.
set -uo pipefail
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
WHICH="${1:-all}"
want() { [ "$WHICH" = all ] || [ "$WHICH" = "$1" ]; }
SQA="SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY"
SQB="SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F32 SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"
passes() {   # passes <outdir> <command...>: kernel trace, FETCH, WRITE, two SQ passes
    local OUT="$PWD/gpurun_out/$1"; shift; mkdir -p "$OUT"
    timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$OUT/kt" --output-format csv -- "$@" > "$OUT/kt.log" 2>&1; echo "$OUT kt rc=$?"
    timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE -d "$OUT/fetch" --output-format csv -- "$@" > "$OUT/fetch.log" 2>&1; echo "$OUT fetch rc=$?"
    timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE -d "$OUT/write" --output-format csv -- "$@" > "$OUT/write.log" 2>&1; echo "$OUT write rc=$?"
    timeout -k 10 300 rocprofv3 --pmc $SQA -d "$OUT/sq" --output-format csv -- "$@" > "$OUT/sq.log" 2>&1; echo "$OUT sq rc=$?"
    timeout -k 10 300 rocprofv3 --pmc $SQB -d "$OUT/sq2" --output-format csv -- "$@" > "$OUT/sq2.log" 2>&1; echo "$OUT sq2 rc=$?"
}
if want persist; then bash scripts/profile_round.sh r04_persist fft "--no-configs3"; fi
if want fft; then bash scripts/profile_round.sh r04_fft fft "--flags 4096 --no-configs3"; fi
if want c4; then
    OUT="$PWD/gpurun_out/r04_c4"; mkdir -p "$OUT"
    export C4_LAZY=1 C4_EVENTS=768
    C4="python3 scripts/c4_traffic.py 128 256"
    timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$OUT/kt" --output-format csv -- $C4 > "$OUT/kt.log" 2>&1; echo "c4 kt rc=$?"
    timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE -d "$OUT/fetch" --output-format csv -- $C4 > "$OUT/fetch.log" 2>&1; echo "c4 fetch rc=$?"
    timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE -d "$OUT/write" --output-format csv -- $C4 > "$OUT/write.log" 2>&1; echo "c4 write rc=$?"
    timeout -k 10 300 rocprofv3 --pmc $SQA -d "$OUT/sq4" --output-format csv -- $C4 > "$OUT/sq4.log" 2>&1; echo "c4 sq rc=$?"
    unset C4_LAZY C4_EVENTS
fi
if want long; then passes r04_long_atom python3 scripts/long_atom_one.py 32768 8; fi
if want lcn; then passes r04_lcn python3 scripts/lcn_headline.py 64 64; fi
if want longest; then passes r04_longest_atom python3 scripts/longest_atoms_time.py 4 32 "fft ("; fi
find gpurun_out/r04_persist gpurun_out/r04_fft gpurun_out/r04_c4 gpurun_out/r04_long_atom gpurun_out/r04_lcn gpurun_out/r04_longest_atom -name "*_agent_info.csv" -delete 2>/dev/null
du -sh gpurun_out/r04_* 2>/dev/null
