# rocprofv3 kernel stats of the batched leaf stage (B = 128 x 1080p): bash tools/leaf_prof.sh <tag>
set -e
TAG=${1:-a}
OUT=$GRAFT_REPO_ROOT/gpurun_out/leaf_prof
mkdir -p $OUT && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 tools/leaf_batch.py 128 10 > $OUT/$TAG.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$TAG -- python3 tools/leaf_batch.py 128 10 >> $OUT/$TAG.log 2>&1
python3 - <<PY
import csv, glob
fs = sorted(glob.glob("$OUT/$TAG/**/*kernel_stats.csv", recursive=True))
out = open("$OUT/${TAG}_summary.txt", "w")
for line in ["# rocprofv3 --kernel-trace --stats -- python3 tools/leaf_batch.py 128 10  (13 batched calls of 128 x 1080p)"] + \
        [f"{r['Name'][:100]:100s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:10.1f} total_ms={float(r['TotalDurationNs'])/1e6:9.2f} pct={r['Percentage']}" for r in csv.DictReader(open(fs[-1]))][:24]:
    print(line); print(line, file=out)
PY
cat $OUT/$TAG.log | grep "ms_per_call"
