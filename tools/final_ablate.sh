# Where lg_final_kernel's dense-path time goes: LG_NO_SKIP bits -- 1 every tile on the stencil path, 8 arithmetic without the
# plane stores (only `traditional` is written), 16 the memory traffic without the arithmetic (planes written from the loaded
# values right after staging).  usage (GPU box): bash tools/final_ablate.sh
for round in 1 2; do
for ns in 1 9 17; do
for p in 0 8; do
  echo "== round=$round no_skip=$ns persist=$p"
  LG_FINAL_PERSIST=$p python3 tools/final_dense.py 128 6 $ns
done
done
done
