#!/bin/bash
# Vector-memory issue pressure of the fp32 headline's kernels (run through gpurun from the repo root): two PMC passes --
# (1) SQ: VMEM instructions, cycles spent issuing them, cycles the issue was blocked by a full TA address / command FIFO, average
#     VMEM instructions in flight;  (2) TA: busy cycles and address stalls.  One line per dispatch of the last forward.
# usage: tools/f32_pmc3.sh <outdir> [bench.py args]
set -e
REPO=$PWD; OUT=$1; shift; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_INSTS_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_LEVEL_VMEM SQ_VALU_MFMA_BUSY_CYCLES -d $REPO/$OUT/pmc_sq --output-format csv -- python3 $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-profile "$@" > $REPO/$OUT/pmc_sq.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_BUFFER_TOTAL_CYCLES_sum -d $REPO/$OUT/pmc_ta --output-format csv -- python3 $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-profile "$@" > $REPO/$OUT/pmc_ta.log 2>&1 || echo "(TA pass failed: see $OUT/pmc_ta.log)"
cd $REPO
python3 - $OUT <<'PY'
import collections, csv, glob, sys
out = sys.argv[1]
def load(d):
    f = glob.glob(f"{out}/{d}/**/*counter_collection.csv", recursive=True)
    rows = list(csv.DictReader(open(f[0]))) if f else []
    t = collections.OrderedDict()
    for r in rows:
        e = t.setdefault(int(r["Dispatch_Id"]), {"name": r["Kernel_Name"].split("(")[0].replace("void iris::", "")[:58],
                                                  "us": (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3})
        e[r["Counter_Name"]] = float(r["Counter_Value"])
    return t
sq, ta = load("pmc_sq"), load("pmc_ta")
lines = []
for k in list(sq)[-26:]:
    v = sq[k]; wc = v.get("SQ_WAVE_CYCLES", 0) or 1
    s = (f"{v['name']:58s} {v['us']:7.1f}us vmem_insts/wave-kcycle={1e3 * v.get('SQ_INSTS_VMEM', 0) / (4 * wc):6.2f} "
         f"issue_cycles_share={v.get('SQ_INST_CYCLES_VMEM', 0) / (4 * wc):.3f} active_vmem={v.get('SQ_ACTIVE_INST_VMEM', 0) / wc:.3f} "
         f"addr_fifo_full={v.get('SQ_VMEM_TA_ADDR_FIFO_FULL', 0) / wc:.3f} cmd_fifo_full={v.get('SQ_VMEM_TA_CMD_FIFO_FULL', 0) / wc:.3f} "
         f"in_flight={v.get('SQ_INST_LEVEL_VMEM', 0) / wc:.2f}")
    lines.append(s)
for i, k in enumerate(list(ta)[-26:]):
    v = ta[k]; cyc = v.get("GRBM_GUI_ACTIVE", 0) / 8 or 1
    extra = f" | ta_busy={v.get('TA_BUSY_sum', 0) / (256 * cyc):.3f} ta_addr_stalled_by_tc={v.get('TA_ADDR_STALLED_BY_TC_CYCLES_sum', 0) / (256 * cyc):.3f}"
    if i < len(lines): lines[i] += extra
open(f"{out}/vmem_pressure.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
