#!/bin/bash
# PMC passes for the screen kernel (run on the GPU box):  scripts/pmc_screen.sh <outdir under gpurun_out>
set -uo pipefail
OUT="$PWD/gpurun_out/$1"; mkdir -p "$OUT"; export TMPDIR=/tmp
B="python3 bench.py --path fft --no-cpu --no-variants --steps 2"
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU" \
           "SQ_INST_CYCLES_VMEM SQ_WAVES SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set -d "$OUT/p$i" --output-format csv -- $B > "$OUT/b$i.json" 2> "$OUT/p$i.err"; echo "set $i rc=$?"
done
