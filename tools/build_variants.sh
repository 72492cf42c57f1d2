#!/bin/bash
# Builds timing-ablation variants of liblgrasp.so: tools/build_variants.sh "<name>:<extra hipcc flags>" ...
# Output: leaf-grasping-vision-ml_amd/csrc/variants/liblgrasp_<name>.so (select with LG_LIB_PATH).
set -e
cd "$(dirname "$0")/../leaf-grasping-vision-ml_amd/csrc"
make -s
mkdir -p variants
FLAGS="-std=c++17 -O3 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-unused-result -Wno-unused-value"
for spec in "$@"; do
  name="${spec%%:*}"; extra="${spec#*:}"
  src="${LG_VARIANT_SRC:-lg_cnn.hip}"; obj="${src%.hip}.o"
  /opt/rocm/bin/hipcc $FLAGS $extra -c "$src" -o "variants/${name}_$obj" 2>/dev/null
  objs=""; for o in $(make -s print-objs); do
    if [ "$o" = "$obj" ]; then objs="$objs variants/${name}_$obj"; else objs="$objs $o"; fi; done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "variants/liblgrasp_${name}.so" $objs -lpthread
  echo "built variants/liblgrasp_${name}.so"
done
