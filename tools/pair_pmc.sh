#!/bin/bash
# PMC passes over one bf16 configs[2] forward (GPU box, through gpurun): where the waves of the MRF kernels spend their cycles.
# usage: tools/pair_pmc.sh [bench args...]   -> gpurun_out/pairpmc_pass*.csv + summary on stdout
set -e
REPO=$PWD
export TMPDIR=/tmp
mkdir -p $REPO/gpurun_out
cd /tmp
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS" \
           "GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_BUSY_CYCLES SQ_INSTS_MFMA"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set -d $REPO/gpurun_out/pairpmc_$i --output-format csv -- python3 $REPO/bench.py --dtype bf16 --batch 32 --frames 500 --no-cpu-baseline --no-extras --no-profile --steps 1 --warmup 0 "$@" > $REPO/gpurun_out/pairpmc_$i.log 2>&1 || echo "pass $i failed: $(tail -2 $REPO/gpurun_out/pairpmc_$i.log)"
done
cd $REPO
for i in 1 2 3; do f=$(find gpurun_out/pairpmc_$i -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp $f gpurun_out/pairpmc_pass$i.csv; done
python3 - <<'PY'
import collections, csv, glob
for path in sorted(glob.glob('gpurun_out/pairpmc_pass*.csv')):
    rows = list(csv.DictReader(open(path)))
    d = collections.OrderedDict()
    for r in rows:
        k = int(r['Dispatch_Id'])
        e = d.setdefault(k, {'name': r['Kernel_Name'], 't': (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3})
        e[r['Counter_Name']] = float(r['Counter_Value'])
    print('==', path)
    for k, v in d.items():
        if 'mfma' not in v['name'] and 'pair' not in v['name']:
            continue
        cyc = v.get('GRBM_GUI_ACTIVE', 0) / 8
        name = v['name'].split('(')[0].replace('void iris::', '')[:52]
        out = [f"{k:3d} {name:52s} {v['t']:7.1f}us clk={cyc / v['t'] / 1e3:.2f}"]
        wc = v.get('SQ_WAVE_CYCLES')
        for c, val in v.items():
            if c in ('name', 't', 'GRBM_GUI_ACTIVE'):
                continue
            if c == 'SQ_VALU_MFMA_BUSY_CYCLES':
                out.append(f"mfma_busy={val / (1024 * cyc):.3f}")
            elif c == 'SQ_WAVE_CYCLES':
                out.append(f"waves/SIMD={val * 4 / (1024 * cyc):.2f}")
            elif c.startswith('SQ_WAIT') or c.startswith('SQ_ACTIVE_INST'):
                out.append(f"{c[3:].lower()}={val / wc:.3f}" if wc else f"{c}={val:.3g}")
            elif c in ('SQ_LDS_BANK_CONFLICT', 'SQ_LDS_IDX_ACTIVE', 'SQ_BUSY_CYCLES', 'SQ_INST_CYCLES_VMEM'):
                out.append(f"{c[3:].lower()}/cyc={val / (256 * cyc):.3f}")
            else:
                out.append(f"{c}={val:.4g}")
        print(' '.join(out))
PY
