#!/bin/bash
# A/B of the d_in row search (lg_hrun_kernel + lg_dtanchor/lg_dtband or the one-level lg_dtsearch_kernel) against the two sweeps:
# headline (256 frames), configs (32 frames, 4K, 720p) and the single-frame latency, one box.
# usage: tools/dt_search_ab.sh <out-prefix> ["<mode>:<algo> ..."]
out=${1:-gpurun_out/dt_ab}
for ma in ${2:-0:2 1:1 1:2 2:2}; do
  mode=${ma%%:*}; algo=${ma##*:}
  LG_DT_SEARCH=$mode LG_DT_SEARCH_ALGO=$algo python bench.py --steps 10 --warmup 3 --node-steps 0 --train-steps 0 --dense-steps 0 --h2d-steps 0 \
      --pipelined 0 --cpu-frames 0 --config-steps 5 > ${out}_mode${mode}_algo$algo.json 2> ${out}_mode${mode}_algo$algo.err || exit 1
  python - <<PY
import json
d = json.loads(open("${out}_mode${mode}_algo$algo.json").read().strip().splitlines()[-1])
print("mode $mode algo $algo value", d["value"], "ms/step", d["ms_per_step"], {k: v for k, v in d.get("kernels_ms", {}).items() if k.startswith("dt") or k in ("bbox", "final")})
for k, v in d.get("configs", {}).items():
    if "kernels_ms" in v: print("   ", k, v["value"], v["ms_per_step"], {a: b for a, b in v["kernels_ms"].items() if a.startswith("dt")})
    else: print("   ", k, {a: b for a, b in v.items() if a != "what"})
PY
done
