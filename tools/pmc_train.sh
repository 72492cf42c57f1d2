# PMC counters of the training-step convolution kernels (MFMA busy, waits, LDS conflicts).  usage: bash tools/pmc_train.sh [batch]
set -e
NB=${1:-1024}
rm -rf gpurun_out/pmc_train && mkdir -p gpurun_out/pmc_train && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
CMD="python tools/train_bench.py --batches $NB --steps 3 --warmup 1"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU --output-format csv -d gpurun_out/pmc_train/p1 -- $CMD > gpurun_out/pmc_train/p1.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VMEM SQ_LDS_IDX_ACTIVE --output-format csv -d gpurun_out/pmc_train/p2 -- $CMD > gpurun_out/pmc_train/p2.log 2>&1 || true
python - <<PY
import csv, glob, collections
for sub in ("p1", "p2"):
    fs = sorted(glob.glob(f"gpurun_out/pmc_train/{sub}/**/*_counter_collection.csv", recursive=True))
    if not fs: print(sub, "no csv"); continue
    d = collections.defaultdict(lambda: collections.defaultdict(list)); dur = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[-1])):
        k = r["Kernel_Name"].split("::")[-1][:44] + " g" + r["Grid_Size"]
        d[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        dur[k].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    for k, c in d.items():
        if "lgt_conv" in k or "lgt_wgrad" in k:
            t = sum(dur[k]) / len(dur[k])
            m = {cn: sum(v) / len(v) for cn, v in c.items()}
            print(f"{sub} {k:64s} dur={t/1e3:8.1f}us " + " ".join(f"{a}={b:.4g}" for a, b in sorted(m.items())))
PY
