#!/bin/bash
# MFMA-pipe utilisation and held clock of the fp32 headline (one PMC pass; run through gpurun from the repo root)
set -e
REPO=$PWD
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY -d $REPO/gpurun_out/f32_pmc --output-format csv -- python3 $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-profile "$@" > $REPO/gpurun_out/f32_pmc.log 2>&1
cd $REPO
python3 tools/pmc_summary.py $(find gpurun_out/f32_pmc -name "*counter_collection.csv" | head -1) 30
