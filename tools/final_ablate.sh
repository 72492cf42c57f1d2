# Where lg_final_kernel's dense-path time goes.  The two ablations that change the RESULTS are compile-time variants of the
# library (LG_FINAL_ABLATE, never in liblgrasp.so itself): 8 = arithmetic without the plane stores (only `traditional` is
# written), 16 = the memory traffic without the arithmetic (planes written from the loaded values right after staging).
# LG_NO_SKIP=1 (every tile on the stencil path; same results) stays a runtime switch.  usage (GPU box): bash tools/final_ablate.sh
set -e
LG_VARIANT_SRC=lg_kernels.hip bash tools/build_variants.sh "fin_nostore:-DLG_FINAL_ABLATE=8" "fin_noarith:-DLG_FINAL_ABLATE=16"
V=leaf-grasping-vision-ml_amd/csrc/variants
for round in 1 2; do
for lib in "" $V/liblgrasp_fin_nostore.so $V/liblgrasp_fin_noarith.so; do
for p in 0 8; do
  echo "== round=$round lib=${lib:-default} persist=$p"
  LG_LIB_PATH=$lib LG_FINAL_PERSIST=$p python3 tools/final_dense.py 128 6 1
done
done
done
