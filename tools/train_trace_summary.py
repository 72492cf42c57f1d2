#!/usr/bin/env python3
"""Per-step kernel breakdown of a rocprofv3 --kernel-trace CSV of tools/train_bench.py: for each batch size (recognised by
the forward convolution's grid) the last 10 steps: kernel time per step, wall time per step, launches, per-kernel sums."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows = [r for r in rows if "lgt_" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# split into runs at changes of the mask kernel's grid (one mask launch per step, grid ~ batch)
runs, curk = [], None
for r in rows:
    if "lgt_mask_kernel" in r["Kernel_Name"]:
        k = r["Grid_Size_X"] if "Grid_Size_X" in r else r["Grid_Size"]
        if k != curk:
            runs.append([])
            curk = k
    if runs:
        runs[-1].append(r)
for run in runs:
    starts = [i for i, r in enumerate(run) if "lgt_mask_kernel" in r["Kernel_Name"]]
    if len(starts) < 12:
        continue
    seg = run[starts[-11]:starts[-1]]
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in seg:
        nm = r["Kernel_Name"].split("lgt_")[1].split("(")[0]
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        agg[nm][0] += 1
        agg[nm][1] += d
    tot = sum(v[1] for v in agg.values())
    wall = (int(seg[-1]["End_Timestamp"]) - int(seg[0]["Start_Timestamp"])) / 1e3
    print("mask grid %s: kernel time/step %.1f us, wall/step %.1f us, launches/step %.1f" % (
        seg[0].get("Grid_Size_X", seg[0].get("Grid_Size")), tot / 10, wall / 10, len(seg) / 10))
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print("  %-36s n/step %5.1f  avg %8.1f us  per-step %8.1f us" % (k, v[0] / 10, v[1] / v[0], v[1] / 10))
