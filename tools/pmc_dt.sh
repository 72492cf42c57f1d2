# Counters of the distance-transform search kernels (lg_hrun / lg_dtsearch / lg_dtanchor / lg_dtband) in one bench pass:
#   bash tools/pmc_dt.sh <tag> [LG_DT_SEARCH mode] [LG_DT_SEARCH_ALGO]
TAG=${1:-a}; MODE=${2:-1}; ALGO=${3:-2}
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_dt
mkdir -p $OUT && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export LG_DT_SEARCH=$MODE LG_DT_SEARCH_ALGO=$ALGO
ARGS="bench.py --steps 3 --warmup 1 --node-steps 0 --train-steps 0 --dense-steps 0 --h2d-steps 0 --pipelined 0 --cpu-frames 0 --config-steps 0"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/${TAG}1 -- python3 $ARGS > $OUT/${TAG}1.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_RD SQ_INSTS_LDS --output-format csv -d $OUT/${TAG}2 -- python3 $ARGS > $OUT/${TAG}2.log 2>&1
python3 - <<PY
import csv, glob, collections
for i in (1, 2):
    fs = sorted(glob.glob("$OUT/${TAG}%d/**/*_counter_collection.csv" % i, recursive=True))
    if not fs:
        print("pass", i, "no counters"); continue
    d = collections.defaultdict(lambda: collections.defaultdict(list)); dur = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[-1])):
        k = r["Kernel_Name"]
        d[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        dur[k].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    for k, c in d.items():
        if "lg_dt" in k or "lg_hrun" in k:
            m = {cn: sum(v) / len(v) for cn, v in c.items()}
            nc = len(next(iter(c.values())))
            t = sum(dur[k]) / len(dur[k])
            print(f"pass {i} {k[k.find('lg_'):][:40]:40s} n={nc:3d} dur={t/1e3:8.1f}us " + " ".join(f"{a}={b:.4g}" for a, b in sorted(m.items())))
PY
