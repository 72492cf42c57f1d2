# where the batched leaf stage's time goes: rocprofv3 kernel averages with parts of k_accumulate / k_hist switched off
# (LG_LEAF_ABLATE bits: 1 no list stores, 2 no accumulator atomics, 4 no ray sum, 8 no histogram atomics; wrong results by design)
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/leaf_ablate
mkdir -p $OUT && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for ab in 0 1 2 4 7 8; do
  LG_LEAF_ABLATE=$ab timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ab$ab -- python3 tools/leaf_batch.py 128 6 > $OUT/ab$ab.log 2>&1
  python3 - <<PY
import csv, glob
fs = sorted(glob.glob("$OUT/ab$ab/**/*kernel_stats.csv", recursive=True))
r = {x['Name'].split('(')[1 if x['Name'].startswith('(') else 0][:24] if False else x['Name'][:60]: float(x['AverageNs'])/1e3 for x in csv.DictReader(open(fs[-1]))}
print("ablate $ab:", {k.replace('(anonymous namespace)::','')[:16]: round(v,1) for k, v in r.items() if 'k_' in k})
PY
done
