# A/B of the default launch form against LG_FINAL_PERSIST=0 through bench.py (headline launch and the dense leg), two rounds on one
# box; prints frames/s, the plane kernel's ms and roofline fraction, the dense launch's ms and fraction.  usage (GPU box): bash tools/dense_ab.sh
mkdir -p gpurun_out/r3e
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "not launch_forms" 2>&1 | tail -2
for i in 1 2; do
  for p in default 0; do
    if [ $p = 0 ]; then export LG_FINAL_PERSIST=0; else unset LG_FINAL_PERSIST; fi
    timeout -k 10 200 python bench.py --cpu-frames 0 --h2d-steps 0 --node-steps 0 --train-steps 0 > gpurun_out/r3e/b_${p}_$i.json 2>/dev/null
    python3 -c "
import json,sys;d=json.loads(open(sys.argv[1]).read().strip().split(chr(10))[-1]);print(sys.argv[1],d['value'],d['kernels_ms']['final'],d['roofline']['frac'],d['roofline_dense']['avg_ms'],d['roofline_dense']['frac'])" gpurun_out/r3e/b_${p}_$i.json
  done
done
