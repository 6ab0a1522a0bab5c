#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc counter_collection.csv: per-dispatch MFMA-pipe utilisation, clock, occupancy.
usage: tools/pmc_summary.py <counter_collection.csv> [last_n_dispatches]"""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n_last = int(sys.argv[2]) if len(sys.argv) > 2 else 30
d = collections.OrderedDict()
for r in rows:
    k = int(r['Dispatch_Id'])
    e = d.setdefault(k, {'name': r['Kernel_Name'], 't': (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3,
                         'grid': r['Grid_Size'], 'vgpr': r['VGPR_Count']})
    e[r['Counter_Name']] = float(r['Counter_Value'])
for k in list(d)[-n_last:]:
    v = d[k]
    cyc = v.get('GRBM_GUI_ACTIVE', 0) / 8          # rocprofv3 sums the 8 XCDs
    if cyc <= 0:
        continue
    name = v['name'].split('(')[0].replace('void iris::', '')[:46]
    out = [f"{k:4d} {name:46s} grid={v['grid']:>7s} {v['t']:8.1f}us clk={cyc / v['t'] / 1e3:.2f}GHz"]
    if 'SQ_VALU_MFMA_BUSY_CYCLES' in v:
        out.append(f"mfma_util={v['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * cyc):.3f}")
    if 'SQ_WAVE_CYCLES' in v:
        out.append(f"waves/SIMD={v['SQ_WAVE_CYCLES'] * 4 / (1024 * cyc):.2f}")
    for c in ('SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY'):
        if c in v and v.get('SQ_WAVE_CYCLES'):
            out.append(f"{c[3:].lower()}={v[c] / v['SQ_WAVE_CYCLES']:.2f}")
    for c in ('SQ_LDS_BANK_CONFLICT', 'SQ_LDS_IDX_ACTIVE', 'FETCH_SIZE', 'WRITE_SIZE', 'TCC_HIT_sum', 'TCC_MISS_sum'):
        if c in v:
            out.append(f"{c}={v[c]:.4g}")
    print(' '.join(out))
