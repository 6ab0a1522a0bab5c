#!/bin/bash
# Kernel durations of one fp32 MRF step under each forced launch plan (release library): tools/forced_plan_probe.sh <tag> "L C [B]" ...
set -e
REPO=$PWD; TAG=${1:-rXX}; shift
export TMPDIR=/tmp
cd /tmp
for shape in "$@"; do
  name=$(echo $shape | tr ' ' x)
  rocprofv3 --kernel-trace -d $REPO/gpurun_out/${TAG}_plans_$name --output-format csv -- python3 $REPO/tools/forced_plan_probe.py $shape > $REPO/gpurun_out/${TAG}_plans_$name.log 2>&1
  python3 - $(find $REPO/gpurun_out/${TAG}_plans_$name -name "*kernel_trace.csv" | head -1) "$shape" <<'PY'
import csv, sys, re
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "mrf_" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
reps = 5; plans = (0, 1, 2, 3, 5, 6)
print("L C [B] =", sys.argv[2], "-- %d mrf dispatches" % len(rows))
for i, plan in enumerate(plans):
    chunk = rows[i * reps:(i + 1) * reps]
    if not chunk: break
    d = sorted((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in chunk)
    nm = re.sub(r"\(.*", "", chunk[0]["Kernel_Name"]).replace("void iris::", "")
    print("  plan %d: median %8.1f us  min %8.1f   grid %6s  %s" % (plan, d[len(d) // 2], d[0], int(chunk[0]["Grid_Size_X"]) // 256 if "Grid_Size_X" in chunk[0] else "?", nm))
PY
done
