"""Kernel durations and the gaps between consecutive kernels from a `rocprofv3 --kernel-trace` CSV (diagnostics).
usage: python tools/kernel_gaps.py <..._kernel_trace.csv> [launches per forward]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
per = int(sys.argv[2]) if len(sys.argv) > 2 else 30
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-per * 5:]                      # the last five forwards
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows)
gaps = [int(b["Start_Timestamp"]) - int(a["End_Timestamp"]) for a, b in zip(rows, rows[1:])]
span = int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])
gaps_in = sorted(g for g in gaps if g < 50000)
print(f"{len(rows)} kernels: busy {busy / 5e3:.1f} us per forward, span {span / 5e3:.1f} us per forward (incl. gaps between forwards)")
print(f"gap between consecutive kernels: median {gaps_in[len(gaps_in) // 2] / 1e3:.2f} us, mean {sum(gaps_in) / len(gaps_in) / 1e3:.2f} us, max {gaps_in[-1] / 1e3:.2f} us")
for r in rows[:per]:
    print(f"  {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:8.2f} us  {r['Kernel_Name'][:90]}")
