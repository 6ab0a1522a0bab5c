#!/usr/bin/env python3
"""HBM traffic per dispatch of the last forward from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE).
usage: tools/hbm_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json> [plan]
plan = one letter per launch of a forward: P conv_pre, U upsample, M MRF launch, O conv_post.  Default: derived from the
kernel names of the last forward in the trace (fp32, batch 1 x 1000: P + 2 x (U + 6 M) + (U + 4 M) + (U + 3 M) + O = 25 launches;
bf16 with fused conv pairs at C <= 128: PUMMMMMMUMMMUMMMUMMMO, 21).
gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE counts 64 B per 128-B request -> read bytes = 2*FETCH_SIZE*1024;
WRITE_SIZE*1024 is exact."""
import collections, csv, json, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from bench import csrc_digest          # the stamp bench.py's committed_traffic checks (kernel sources this pass was taken on)

def load(path, counter):
    d = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            d[int(r["Dispatch_Id"])] = (r["Kernel_Name"], int(r["Grid_Size"]), float(r["Counter_Value"]))
    return list(d.values())

fetch_all, write_all = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
if len(sys.argv) > 4:
    plan = sys.argv[4]
else:
    # the last forward = the dispatches behind the second-to-last conv_post up to the last one; one letter per launch by kernel name
    posts = [i for i, (kn, _, _) in enumerate(fetch_all) if "conv_post" in kn]
    assert len(posts) >= 2, "need at least two forwards in the trace"
    names = [kn for kn, _, _ in fetch_all[posts[-2] + 1:posts[-1] + 1]]
    while names and "iris" not in names[0]:          # (the runtime's own memset kernel in front of a forward is not a launch of ours)
        names.pop(0)
    plan = "".join("O" if "conv_post" in kn else ("M" if "mrf_" in kn else ("P" if i == 0 else "U")) for i, kn in enumerate(names))
n = len(plan)
fetch, write = fetch_all[-n:], write_all[-n:]
assert len(fetch) == len(write) == n, (len(fetch), len(write))
rows, mrf = [], []
mrf_pos = {i for i, c in enumerate(plan) if c == "M"}
for pos, ((kn, grid, f), (kn2, _, w)) in enumerate(zip(fetch, write)):
    assert kn == kn2
    name = kn.split("(")[0].replace("void iris::", "").replace("b16::", "b16::")
    t = 2.0 * f * 1024.0 + w * 1024.0
    rows.append({"kernel": name, "grid": grid, "FETCH_SIZE_KB": f, "WRITE_SIZE_KB": w, "traffic_bytes_corrected": t})
    if pos in mrf_pos:
        mrf.append(t)
out = {"plan": plan, "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (two separate passes); last forward (%d dispatches)" % n,
       "correction": "gfx950: read bytes = 2*FETCH_SIZE*1024; WRITE_SIZE*1024 exact (MI355X_MICROARCH.md, HBM)",
       "csrc_sha16": csrc_digest(), "mrf_launches": len(mrf), "mrf_traffic_bytes_per_launch": (sum(mrf) / len(mrf)) if mrf else None, "per_dispatch": rows}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print("mrf launches %d, traffic/launch %.1f MB, whole forward %.1f MB" % (len(mrf), (sum(mrf) / max(len(mrf), 1)) / 1e6, sum(r["traffic_bytes_corrected"] for r in rows) / 1e6))
