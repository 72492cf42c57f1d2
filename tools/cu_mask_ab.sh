F="--cpu-frames 0 --node-steps 0 --train-steps 0 --dense-steps 0 --h2d-steps 0 --config-steps 0"
for cus in 0 224 192 160; do for n in 1 2 3; do
  LG_CNN_CUS=$cus timeout -k 10 200 python bench.py --inflight $n $F > gpurun_out/r3h_${cus}_$n.json 2>/dev/null
  python - <<PY
import json
d=json.load(open("gpurun_out/r3h_${cus}_$n.json"))
k=d['kernels_ms']
print("cus $cus inflight $n fps", d['value'], "ms", d['ms_per_step'], "final", k['final'], "cnn", k['cnn'], "dt", k['dt_fwd'], k['dt_bwd'], "topk", k['topk'], flush=True)
PY
done; done
