# per-layer kernel times of the CNN forward at a given patch count, default library and variants: bash tools/cnn_layers_n.sh <patches> [variant.so ...]
NP=${1:-640}; shift || true
OUT=$GRAFT_REPO_ROOT/gpurun_out/cnn_layers_n
mkdir -p $OUT && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for lib in "" "$@"; do
  i=$((i+1))
  LG_LIB_PATH=$lib timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/n${NP}_v$i -- python3 tools/cnn_time.py $NP > $OUT/n${NP}_v$i.log 2>&1
  echo "== ${lib:-default} patches=$NP"
  python3 - <<PY
import csv, glob
fs = sorted(glob.glob("$OUT/n${NP}_v$i/**/*kernel_stats.csv", recursive=True))
tot = 0
for r in csv.DictReader(open(fs[-1])):
    if "lg_" in r["Name"]:
        print("%-90s calls %4s avg %9.1f us" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3)); tot += float(r["AverageNs"]) / 1e3
print("sum %.1f us" % tot)
PY
done
