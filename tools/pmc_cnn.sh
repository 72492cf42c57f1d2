set -e
mkdir -p gpurun_out/pmc_cnn && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
CMD="python bench.py --steps 3 --warmup 1 --cpu-frames 0 --batch 128"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU --output-format csv -d gpurun_out/pmc_cnn/a -- $CMD > gpurun_out/pmc_cnn/a.log 2>&1
python - <<PY
import csv, glob, collections
p = sorted(glob.glob("gpurun_out/pmc_cnn/a/runc/*_counter_collection.csv"))[-1]
d = collections.defaultdict(lambda: collections.defaultdict(list)); dur = collections.defaultdict(list)
for r in csv.DictReader(open(p)):
    k = r["Kernel_Name"][:70]
    d[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    dur[k].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
for k, c in d.items():
    if "lg_" in k:
        t = sum(dur[k]) / len(dur[k])
        m = {cn: sum(v) / len(v) for cn, v in c.items()}
        clk = m.get("GRBM_GUI_ACTIVE", 0) / 8 / t if t else 0
        print(f"{k[:60]:60s} dur={t/1e3:8.1f}us clk~{clk:5.2f}GHz mfma_busy={m.get('SQ_VALU_MFMA_BUSY_CYCLES',0):.3g} wave_cyc={m.get('SQ_WAVE_CYCLES',0):.3g} wait_any={m.get('SQ_WAIT_ANY',0):.3g} wait_inst={m.get('SQ_WAIT_INST_ANY',0):.3g} active={m.get('SQ_ACTIVE_INST_ANY',0):.3g} busy={m.get('SQ_BUSY_CYCLES',0):.3g}")
PY
