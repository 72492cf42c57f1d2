# LDS bank-conflict cycles of the F(4x4,3x3) kernels per library variant (which access owns them?):
#   bash tools/pmc_cnn_lds.sh <tag> [variant.so ...]      (variants: tools/build_variants.sh, LG_W4_EXP ablations)
TAG=${1:-a}; shift || true
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_cnn_lds
mkdir -p $OUT && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for lib in "" "$@"; do
  i=$((i+1))
  LG_LIB_PATH=$lib timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d $OUT/${TAG}$i -- python3 tools/cnn_run.py 5120 3 > $OUT/${TAG}$i.log 2>&1
  python3 - <<PY
import csv, glob, collections
fs = sorted(glob.glob("$OUT/${TAG}$i/**/*_counter_collection.csv", recursive=True))
d = collections.defaultdict(lambda: collections.defaultdict(list)); dur = collections.defaultdict(list)
for r in csv.DictReader(open(fs[-1])):
    k = r["Kernel_Name"]
    d[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    dur[k].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
print("== ${lib:-default}")
for k, c in d.items():
    if "wino4" in k:
        m = {cn: sum(v) / len(v) for cn, v in c.items()}
        t = sum(dur[k]) / len(dur[k])
        print(f"{k[k.find('lg_'):][:52]:52s} dur={t/1e3:8.1f}us conflict/idx_active={m['SQ_LDS_BANK_CONFLICT']/max(1,m['SQ_LDS_IDX_ACTIVE']):.3f} " + " ".join(f"{a}={b:.4g}" for a, b in sorted(m.items())))
PY
done
