# PMC counters of the CNN kernels (MFMA busy, waits, LDS conflicts).  usage: bash tools/pmc_cnn.sh <tag> [ENV=VAL ...]
set -e
TAG=${1:-a}; shift || true
for kv in "$@"; do export "$kv"; done
mkdir -p gpurun_out/pmc_cnn && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
CMD="python bench.py --steps 3 --warmup 1 --cpu-frames 0 --batch 128"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU --output-format csv -d gpurun_out/pmc_cnn/${TAG}1 -- $CMD > gpurun_out/pmc_cnn/${TAG}1.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VMEM SQ_LDS_IDX_ACTIVE --output-format csv -d gpurun_out/pmc_cnn/${TAG}2 -- $CMD > gpurun_out/pmc_cnn/${TAG}2.log 2>&1 || true
python - <<PY
import csv, glob, collections
for sub in ("${TAG}1", "${TAG}2"):
    fs = sorted(glob.glob(f"gpurun_out/pmc_cnn/{sub}/**/*_counter_collection.csv", recursive=True))
    if not fs: print(sub, "no csv"); continue
    d = collections.defaultdict(lambda: collections.defaultdict(list)); dur = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[-1])):
        k = r["Kernel_Name"][:70]
        d[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        dur[k].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    for k, c in d.items():
        if "wino" in k or "conv3x3" in k:
            t = sum(dur[k]) / len(dur[k])
            m = {cn: sum(v) / len(v) for cn, v in c.items()}
            print(f"{sub} {k[40:70]:30s} dur={t/1e3:8.1f}us " + " ".join(f"{a}={b:.4g}" for a, b in sorted(m.items())))
PY
