#!/bin/bash
# Two PMC passes (FETCH_SIZE, WRITE_SIZE) over bench.py on the GPU box; writes gpurun_out/<tag>_hbm_traffic.json.
# usage: [PLAN=<letters, see hbm_traffic.py>] tools/hbm_traffic.sh <tag> [bench.py args...]      (run through gpurun from the repo root)
set -e
REPO=$PWD; TAG=$1; shift
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $REPO/gpurun_out/${TAG}_fetch --output-format csv -- python3 $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-profile "$@" > $REPO/gpurun_out/${TAG}_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $REPO/gpurun_out/${TAG}_write --output-format csv -- python3 $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-profile "$@" > $REPO/gpurun_out/${TAG}_write.log 2>&1
cd $REPO
python3 tools/hbm_traffic.py $(find gpurun_out/${TAG}_fetch -name "*counter_collection.csv" | head -1) $(find gpurun_out/${TAG}_write -name "*counter_collection.csv" | head -1) gpurun_out/${TAG}_hbm_traffic.json $PLAN
