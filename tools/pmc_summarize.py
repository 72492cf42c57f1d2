#!/usr/bin/env python3
"""Folds the rocprofv3 --pmc passes of tools/pmc_final.sh (gpurun_out/pmc/{sq,fetch,write,tcc}) into
profiles/<out> (default r03_pmc_counters.json): per-kernel counter averages per dispatch and the HBM traffic of lg_final_kernel
(FETCH_SIZE doubled per the gfx950 note of MI355X_MICROARCH.md, WRITE_SIZE as is, both KiB).
usage: python tools/pmc_summarize.py <batch> <H> <W> <algorithmic bytes per px of lg_final_kernel for the benched masks> [out.json]"""
import collections
import csv
import glob
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
B, H, W = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
alg = float(sys.argv[4])
px = B * H * W
kern = collections.defaultdict(lambda: collections.defaultdict(list))
for sub in ("sq", "fetch", "write", "tcc"):
    files = glob.glob(os.path.join(REPO, "gpurun_out", "pmc", sub, "**", "*_counter_collection.csv"), recursive=True)
    for f in sorted(files, key=os.path.getmtime)[-1:]:   # newest pass only (older runs may linger in gpurun_out/)
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].split("(")[0]
            if "lg_" in name:
                kern[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {
    "command": "tools/pmc_final.sh: rocprofv3 --kernel-trace --pmc <COUNTERS> --output-format csv -- python bench.py "
               f"--steps 3 --warmup 1 --cpu-frames 0 --batch {B}  (one pass per counter set: FETCH_SIZE | WRITE_SIZE | "
               "TCC_HIT_sum TCC_MISS_sum | SQ_*); folded by tools/pmc_summarize.py",
    "units": "FETCH_SIZE / WRITE_SIZE in KiB per dispatch; gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE "
             "reports 1/2 of the bytes of wide coalesced reads -> doubled in the summary; WRITE_SIZE is exact for "
             "16-B-per-lane stores",
    "batch": B, "pixels_per_launch": px,
    "kernels": {k: {c: sum(v) / len(v) for c, v in sorted(cs.items())} for k, cs in sorted(kern.items())},
}
fk = next((v for k, v in out["kernels"].items() if "lg_final_kernel" in k), None)   # (templated: "void lg_final_kernel<true, true>")
if fk and "FETCH_SIZE" in fk and "WRITE_SIZE" in fk:
    rd, wr = 2.0 * fk["FETCH_SIZE"] * 1024.0, fk["WRITE_SIZE"] * 1024.0
    out["lg_final_kernel_summary"] = {
        "hbm_read_bytes_corrected": rd, "hbm_write_bytes": wr, "traffic_bytes_per_launch": rd + wr,
        "traffic_bytes_per_px": (rd + wr) / px, "algorithmic_bytes_per_px": alg,
        "dense_bytes_per_px": 37.25, "traffic_over_algorithmic": (rd + wr) / px / alg,
        "l2_hit_rate": fk["TCC_HIT_sum"] / (fk["TCC_HIT_sum"] + fk["TCC_MISS_sum"]) if "TCC_HIT_sum" in fk else None,
    }
json.dump(out, open(os.path.join(REPO, "profiles", sys.argv[5] if len(sys.argv) > 5 else "r03_pmc_counters.json"), "w"), indent=1)
print(json.dumps(out.get("lg_final_kernel_summary"), indent=1))
for k, v in out["kernels"].items():
    print(k[:50], {c: round(x) for c, x in v.items() if c in ("FETCH_SIZE", "WRITE_SIZE")})
