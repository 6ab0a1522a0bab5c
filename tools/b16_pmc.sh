#!/bin/bash
# PMC passes over one bf16 forward of configs[2] (run on the GPU box through gpurun).  Output: gpurun_out/pmc16_*/
# usage: tools/b16_pmc.sh [extra env assignments...]
set -e
REPO=$PWD
export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
mkdir -p $REPO/gpurun_out
cd /tmp
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE" "WRITE_SIZE" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set -d $REPO/gpurun_out/pmc16_$i --output-format csv -- python3 $REPO/bench.py --dtype bf16 --batch 32 --frames 500 --no-cpu-baseline --no-extras --no-profile --steps 1 --warmup 0 > $REPO/gpurun_out/pmc16_$i.log 2>&1 || echo "pass $i failed"
done
cd $REPO
for i in 1 2 3 4 5 6; do f=$(find gpurun_out/pmc16_$i -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp $f gpurun_out/pmc16_pass$i.csv; done
ls gpurun_out/pmc16_pass*.csv
