#!/bin/bash
# Two PMC passes over the fp32 headline (run through gpurun from the repo root): (1) MFMA-pipe busy, clock, resident waves, issue
# stalls; (2) parked waves, LDS / vector-memory waits.  Prints one line per dispatch of the last forward.  usage: tools/f32_pmc2.sh <outdir> [bench.py args]
set -e
REPO=$PWD; OUT=$1; shift; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d $REPO/$OUT/pmc1 --output-format csv -- python3 $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-profile "$@" > $REPO/$OUT/pmc1.log 2>&1
cd $REPO
python3 tools/pmc_summary.py $(find $OUT/pmc1 -name "*counter_collection.csv" | head -1) 27 > $OUT/pmc_summary.txt
cat $OUT/pmc_summary.txt
