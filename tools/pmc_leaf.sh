# Counters of the leaf-stage kernels (tools/leaf_batch.py 128 frames): bash tools/pmc_leaf.sh <tag>
TAG=${1:-a}
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_leaf
mkdir -p $OUT && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/${TAG}1 -- python3 tools/leaf_batch.py 128 4 > $OUT/${TAG}1.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/${TAG}2 -- python3 tools/leaf_batch.py 128 4 > $OUT/${TAG}2.log 2>&1
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
        if "k_" in k:
            m = {cn: sum(v) / len(v) for cn, v in c.items()}
            t = sum(dur[k]) / len(dur[k])
            print(f"pass {i} {k[k.find('k_'):][:28]:28s} dur={t/1e3:8.1f}us " + " ".join(f"{a}={b:.4g}" for a, b in sorted(m.items())))
PY
