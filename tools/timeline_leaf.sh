# Timeline of one steady-state leaf-stage call (tools/leaf_batch.py) from rocprofv3 --kernel-trace: bash tools/timeline_leaf.sh <tag> [frames]
TAG=${1:-a}; NB=${2:-128}
OUT=$GRAFT_REPO_ROOT/gpurun_out/timeline_leaf
mkdir -p $OUT && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/$TAG -- python3 tools/leaf_batch.py $NB 6 > $OUT/$TAG.log 2>&1
python3 - <<PY
import csv, glob
fs = sorted(glob.glob("$OUT/$TAG/**/*_kernel_trace.csv", recursive=True))
rows = list(csv.DictReader(open(fs[-1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "k_presence_bits" in r["Kernel_Name"]]
a, b = marks[-2], marks[-1]
# include the memsets in front of the presence kernel
while a > 0 and "fillBuffer" in rows[a - 1]["Kernel_Name"]: a -= 1
while b > 0 and "fillBuffer" in rows[b - 1]["Kernel_Name"]: b -= 1
t0 = int(rows[a]["Start_Timestamp"])
print("call of %.3f ms (first memset to first memset), %d kernels" % ((int(rows[b]["Start_Timestamp"]) - t0) / 1e6, b - a))
last_end = t0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"]
    name = name[name.find("k_"):] if "k_" in name else name
    print("%9.3f ms  +%8.1f us  q%-3s  gap-from-prev-end %7.1f us  %s" % ((s - t0) / 1e6, (e - s) / 1e3, r.get("Queue_Id", "?"), (s - last_end) / 1e3, name[:60]))
    last_end = max(last_end, e)
PY
