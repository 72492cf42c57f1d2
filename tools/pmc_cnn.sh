# Kernel trace + PMC counters of the CNN kernels (MFMA busy, waits, LDS conflicts).
# usage (on the GPU box): bash tools/pmc_cnn.sh <tag> [ENV=VAL ...]     e.g.  bash tools/pmc_cnn.sh f43   /   bash tools/pmc_cnn.sh f23 LG_CNN_F23=1
set -e
TAG=${1:-a}; shift || true
for kv in "$@"; do export "$kv"; done
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_cnn
mkdir -p $OUT && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}0 -- python3 tools/cnn_run.py 5120 5 > $OUT/${TAG}0.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU --output-format csv -d $OUT/${TAG}1 -- python3 tools/cnn_run.py 5120 3 > $OUT/${TAG}1.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VMEM SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/${TAG}2 -- python3 tools/cnn_run.py 5120 3 > $OUT/${TAG}2.log 2>&1 || true
python3 - <<PY
import csv, glob, collections
out = open("$OUT/${TAG}_summary.txt", "w")
def P(*a):
    print(*a); print(*a, file=out)
fs = sorted(glob.glob("$OUT/${TAG}0/**/*kernel_stats.csv", recursive=True))
if fs:
    P("# rocprofv3 --kernel-trace --stats: python3 tools/cnn_run.py 5120 5   [${TAG}] $@")
    for r in csv.DictReader(open(fs[-1])):
        P(f"{r['Name'][:110]:110s} calls={r['Calls']:>4s} avg_us={float(r['AverageNs'])/1e3:9.1f} pct={r['Percentage']}")
for sub in ("${TAG}1", "${TAG}2"):
    fs = sorted(glob.glob(f"$OUT/{sub}/**/*_counter_collection.csv", recursive=True))
    if not fs: P(sub, "no csv"); continue
    d = collections.defaultdict(lambda: collections.defaultdict(list)); dur = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[-1])):
        k = r["Kernel_Name"]
        d[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        dur[k].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    for k, c in d.items():
        if "wino" in k or "conv" in k:
            t = sum(dur[k]) / len(dur[k])
            m = {cn: sum(v) / len(v) for cn, v in c.items()}
            extra = ""
            if "SQ_VALU_MFMA_BUSY_CYCLES" in m and "GRBM_GUI_ACTIVE" in m:
                extra = f" mfma_busy_share={m['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024 / (m['GRBM_GUI_ACTIVE'] / 8):.3f} clock_GHz={m['GRBM_GUI_ACTIVE'] / 8 / t:.2f}"
            P(f"{sub} {k[k.find('lg_'):][:60]:60s} dur={t/1e3:8.1f}us" + extra + " " + " ".join(f"{a}={b:.4g}" for a, b in sorted(m.items())))
PY
