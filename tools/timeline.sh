# Timeline of one steady-state step from rocprofv3 --kernel-trace: every kernel's queue, start offset and duration, the gaps
# between consecutive kernels of the critical chain.   bash tools/timeline.sh <tag> [batch] [extra env assignments...]
TAG=${1:-a}; BATCH=${2:-256}
OUT=$GRAFT_REPO_ROOT/gpurun_out/timeline
mkdir -p $OUT && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/$TAG -- python3 bench.py --batch $BATCH --steps 4 --warmup 2 --node-steps 0 --train-steps 0 --dense-steps 0 --h2d-steps 0 --pipelined 0 --cpu-frames 0 --config-steps 0 > $OUT/$TAG.log 2>&1
python3 - <<PY
import csv, glob
fs = sorted(glob.glob("$OUT/$TAG/**/*_kernel_trace.csv", recursive=True))
rows = list(csv.DictReader(open(fs[-1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last complete step: from the last lg_pack_bits launch but one to the next
packs = [i for i, r in enumerate(rows) if "lg_pack_bits" in r["Kernel_Name"]]
a, b = packs[-2], packs[-1]
t0 = int(rows[a]["Start_Timestamp"])
print("step of %.3f ms (pack to pack), %d kernels" % ((int(rows[b]["Start_Timestamp"]) - t0) / 1e6, b - a))
last_end = t0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"]
    name = name[name.find("lg_"):] if "lg_" in name else name
    print("%9.3f ms  +%8.1f us  q%-3s  gap-from-prev-end %7.1f us  %s" % ((s - t0) / 1e6, (e - s) / 1e3, r.get("Queue_Id", "?"), (s - last_end) / 1e3, name[:70]))
    last_end = max(last_end, e)
PY
