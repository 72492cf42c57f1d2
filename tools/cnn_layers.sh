# per-layer kernel times of the CNN forward (rocprofv3 --kernel-trace --stats of tools/cnn_time.py) for the default library and
# the variants given.  usage (GPU box): bash tools/cnn_layers.sh [variant.so ...]
OUT=$GRAFT_REPO_ROOT/gpurun_out/cnn_layers
mkdir -p $OUT && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for lib in "" "$@"; do
  i=$((i+1))
  LG_LIB_PATH=$lib timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/v$i -- python3 tools/cnn_time.py 5120 > $OUT/v$i.log 2>&1
  echo "== ${lib:-default}"
  python3 - <<PY
import csv, glob
fs = sorted(glob.glob("$OUT/v$i/**/*kernel_stats.csv", recursive=True))
for r in csv.DictReader(open(fs[-1])):
    if "lg_" in r["Name"]:
        print("%-90s calls %4s avg %9.1f us" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
