#!/bin/bash
# rocprofv3 kernel stats of the fp32 forward at the short / mid shapes the bench line's `grid` quotes (through gpurun, from the repo root):
# gpurun_out/<tag>_kernel_stats_<B>x<T>.csv.  usage: tools/short_shape_profiles.sh <tag> ["1 100" "1 500" ...]
set -e
REPO=$PWD; TAG=${1:-rXX}; shift || true
export TMPDIR=/tmp
cd /tmp
for shape in "${@:-1 100}"; do
  set -- $shape
  rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/${TAG}_prof_$1x$2 --output-format csv -- python3 $REPO/bench.py --batch $1 --frames $2 --steps 10 --warmup 2 --no-cpu-baseline --no-extras > $REPO/gpurun_out/${TAG}_bench_under_rocprof_$1x$2.json 2> $REPO/gpurun_out/${TAG}_prof_$1x$2.log
  cp $(find $REPO/gpurun_out/${TAG}_prof_$1x$2 -name "*kernel_stats.csv" | head -1) $REPO/gpurun_out/${TAG}_kernel_stats_$1x$2.csv
done
ls $REPO/gpurun_out/${TAG}_kernel_stats_*x*.csv
