#!/bin/bash
# Run on the GPU box (via gpurun): the rocprofv3 passes whose summaries are committed under profiles/.
#   usage: scripts/profile_round.sh <outdir under gpurun_out> <bench path: fft|incremental> [more bench.py arguments]
set -uo pipefail
OUT="$PWD/gpurun_out/$1"; PATHARG="$2"; MORE="${3:-}"
mkdir -p "$OUT"; export TMPDIR=/tmp
B="python3 bench.py --path $PATHARG --no-cpu --no-variants $MORE"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$OUT/kt" --output-format csv -- $B --steps 5 > "$OUT/bench_under_rocprof_kt.json" 2> "$OUT/kt.err"; echo "kt rc=$?"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE -d "$OUT/fetch" --output-format csv -- $B --steps 2 > "$OUT/bench_fetch.json" 2> "$OUT/fetch.err"; echo "fetch rc=$?"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE -d "$OUT/write" --output-format csv -- $B --steps 2 > "$OUT/bench_write.json" 2> "$OUT/write.err"; echo "write rc=$?"
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_F32 -d "$OUT/sq" --output-format csv -- $B --steps 2 > "$OUT/bench_sq.json" 2> "$OUT/sq.err"; echo "sq rc=$?"
timeout -k 10 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS -d "$OUT/sq2" --output-format csv -- $B --steps 2 > "$OUT/bench_sq2.json" 2> "$OUT/sq2.err"; echo "sq2 rc=$?"
