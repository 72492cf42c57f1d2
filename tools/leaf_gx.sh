# batched leaf stage under rocprofv3 for several workgroup budgets of the streaming passes (LG_LEAF_GX / B workgroups per frame)
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/leaf_gx
mkdir -p $OUT && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for gx in ${1:-2048 4096 8192 16384}; do
  LG_LEAF_GX=$gx timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/gx$gx -- python3 tools/leaf_batch.py 128 6 > $OUT/gx$gx.log 2>&1
  python3 - <<PY
import csv, glob
fs = sorted(glob.glob("$OUT/gx$gx/**/*kernel_stats.csv", recursive=True))
r = {x['Name'][:60]: float(x['AverageNs'])/1e3 for x in csv.DictReader(open(fs[-1]))}
print("gx budget $gx:", {k.replace('(anonymous namespace)::','').replace('void ','')[:14]: round(v,1) for k, v in r.items() if 'k_' in k}, open("$OUT/gx$gx.log").read().strip().splitlines()[-1][:60])
PY
done
