# A/B of lg_final_kernel on the dense (every tile on the stencil path) workload: resident workgroups walking the tiles
# (LG_FINAL_PERSIST = workgroups per CU, default 8) against one workgroup per tile (0); library variants from
# tools/build_variants.sh, two rounds (boxes drift).  usage (GPU box): bash tools/final_persist_ab.sh [variant.so ...]
for round in 1 2; do
for lib in "" "$@"; do
for p in 0 8; do
  echo "== round=$round lib=${lib:-default} persist=$p"
  LG_LIB_PATH=$lib LG_FINAL_PERSIST=$p python3 tools/final_dense.py 128 6 1
done
done
done
