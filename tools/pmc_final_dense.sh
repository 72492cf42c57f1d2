# rocprofv3 passes over the dense (every tile on the stencil path) launch of lg_final_kernel: kernel trace + stats, SQ counters,
# FETCH_SIZE, WRITE_SIZE (separate --pmc passes, as MI355X_MICROARCH.md prescribes), folded into profiles-ready files.
# usage (on the GPU box): bash tools/pmc_final_dense.sh [no_skip=1]
set -e
NS=${1:-1}
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_dense
mkdir -p $OUT && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
CMD="python3 tools/final_dense.py 32 3 $NS"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -- $CMD > $OUT/t.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/sq -- $CMD > $OUT/sq.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- $CMD > $OUT/fetch.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- $CMD > $OUT/write.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE --output-format csv -d $OUT/tcc -- $CMD > $OUT/tcc.log 2>&1
python3 - <<PY
import csv, glob, json, collections
px = 32 * 1080 * 1920
res = {"command": "tools/pmc_final_dense.sh $NS: rocprofv3 --kernel-trace --pmc <COUNTERS> -- python3 tools/final_dense.py 32 3 $NS (one pass per counter set)",
       "units": "FETCH_SIZE / WRITE_SIZE in KiB per dispatch; gfx950: FETCH_SIZE reports 1/2 of the bytes of wide coalesced reads -> doubled",
       "pixels_per_launch": px, "counters": {}}
for sub in ("sq", "fetch", "write", "tcc"):
    fs = sorted(glob.glob("$OUT/%s/**/*_counter_collection.csv" % sub, recursive=True))
    if not fs: continue
    d = collections.defaultdict(list); dur = []
    for r in csv.DictReader(open(fs[-1])):
        if "lg_final_kernel" in r["Kernel_Name"]:
            d[r["Counter_Name"]].append(float(r["Counter_Value"]))
            dur.append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    for k, v in d.items():
        res["counters"][k] = sum(v[1:]) / max(1, len(v) - 1)     # skip the first (warm-up) launch
    if dur: res["avg_ns_" + sub] = sum(dur[1:]) / max(1, len(dur) - 1)
c = res["counters"]
if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
    rd, wr = 2.0 * c["FETCH_SIZE"] * 1024.0, c["WRITE_SIZE"] * 1024.0
    res.update({"hbm_read_bytes_corrected": rd, "hbm_write_bytes": wr, "traffic_bytes_per_px": (rd + wr) / px,
                "algorithmic_bytes_per_px": 37.25, "traffic_over_algorithmic": (rd + wr) / px / 37.25})
fs = sorted(glob.glob("$OUT/t/**/*kernel_stats.csv", recursive=True))
if fs:
    res["kernel_stats"] = [{"name": r["Name"][:60], "calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3} for r in csv.DictReader(open(fs[-1])) if "lg_" in r["Name"]]
json.dump(res, open("$OUT/final_dense_$NS.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
