# How much does lg_final_kernel's time vary from PROCESS to process on one box (same code, same inputs)?  usage: bash tools/final_variance.sh [runs] [ENV=VAL ...]
R=${1:-6}; shift || true
for kv in "$@"; do export "$kv"; done
F="--steps 6 --warmup 2 --cpu-frames 0 --train-steps 0 --node-steps 0 --h2d-steps 0 --config-steps 0 --pipelined 0 --dense-steps 0"
for r in $(seq $R); do
  python3 bench.py $F 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels_ms']; print('$*', 'final', k['final'], 'frac', d['roofline']['frac'], 'cnn', k['cnn'], 'fps', d['value'])"
done
