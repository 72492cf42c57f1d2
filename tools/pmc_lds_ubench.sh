# LDS cycles and bank-conflict cycles per instruction for the access patterns of tools/ubench/lds_conflict.hip
OUT=$GRAFT_REPO_ROOT/gpurun_out/lds_ubench
mkdir -p $OUT && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/b -- tools/ubench/lds_conflict > $OUT/b.log 2>&1
python3 - <<PY
import csv, glob, collections, re
fs = sorted(glob.glob("$OUT/b/**/*_counter_collection.csv", recursive=True))
d = collections.OrderedDict()
for r in csv.DictReader(open(fs[-1])):
    d.setdefault(r["Kernel_Name"], {})[r["Counter_Name"]] = float(r["Counter_Value"])
n = 128 * 4 * 4096
kinds = ("b128 cols 0-3", "b128 cols 4-7", "b64  cols 4-5")
out = open("$OUT/summary_b.txt", "w")
for k, m in d.items():
    g2 = re.search(r"kseq<(\d+), (\d+), (\d+), (\d+)>", k)
    if g2:
        n2 = 128 * 4 * (4096 // 8) * 10
        line = f"WI {g2.group(1)} WP {g2.group(2)} RS {g2.group(3)} S {g2.group(4)} kernel's sequence (10 b128 back to back, 4 waves): cycles/instr {m['SQ_LDS_IDX_ACTIVE']/n2:6.2f}  conflict cycles/instr {m['SQ_LDS_BANK_CONFLICT']/n2:6.2f}"
        print(line); print(line, file=out)
        continue
    g = re.search(r"k<(\d+), (\d+), (\d+), (\d+), (\d+), (\d+)>", k)
    if not g: continue
    wi, wp, rs, s, kind, perm = map(int, g.groups())
    line = f"WI {wi:2d} WP {wp:2d} RS {rs:4d} S {s:5d} groups {perm:04d} {kinds[kind]}: cycles/instr {m['SQ_LDS_IDX_ACTIVE']/n:6.2f}  conflict cycles/instr {m['SQ_LDS_BANK_CONFLICT']/n:6.2f}"
    print(line); print(line, file=out)
PY
