#!/bin/bash
# round 3, call k: where the waves of the fp32 MRF launches wait (SQ PMC pass: parked at waitcnt/barrier vs issue stalls vs LDS)
set -e
REPO=$PWD; O=$REPO/gpurun_out/r03k
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_BUSY_CYCLES -d $O/pmc1 --output-format csv -- python3 $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-profile > $O/pmc1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -d $O/pmc2 --output-format csv -- python3 $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-profile > $O/pmc2.log 2>&1 || echo "second pass failed (counter names?)"
cd $REPO
python3 tools/pmc_summary.py $(find $O/pmc1 -name "*counter_collection.csv" | head -1) 27 | tee $O/pmc1_summary.txt
python3 - $O <<'PY'
import csv, sys, glob, collections
f = glob.glob(sys.argv[1] + "/pmc2/**/*counter_collection.csv", recursive=True)
if f:
    d = collections.OrderedDict()
    for r in csv.DictReader(open(f[0])):
        e = d.setdefault(int(r["Dispatch_Id"]), {"name": r["Kernel_Name"].split("(")[0].replace("void iris::", "")[:44]})
        e[r["Counter_Name"]] = float(r["Counter_Value"])
    for k in list(d)[-27:]:
        v = d[k]
        print(k, v["name"], {c: int(x) for c, x in v.items() if c != "name"})
PY
