# A/B of the plane kernel on one box: default library vs a variant (LG_LIB_PATH), alternating, headline masks (256 frames) and
# the dense launch (128 frames, every tile on the stencil path).  usage (GPU box): bash tools/final_ab.sh <variant.so> [rounds]
V=$1; R=${2:-3}
F="--steps 8 --warmup 2 --cpu-frames 0 --train-steps 0 --node-steps 0 --h2d-steps 0 --config-steps 0 --pipelined 0 --dense-steps 4"
for r in $(seq $R); do for lib in "" $V; do
  LG_LIB_PATH=$lib python3 bench.py $F 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('${lib:-default}'.split('/')[-1], 'final ms', d['kernels_ms']['final'], 'frac', d['roofline']['frac'], '| dense ms', d['roofline_dense']['avg_ms'], 'frac', d['roofline_dense']['frac'], '| fps', d['value'])"
done; done
