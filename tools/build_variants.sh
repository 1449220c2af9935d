#!/bin/bash
# engine variants for A/B runs on the GPU box: star_amd/lib/variants/libstaramd_<tag>.so, each = the production objects with k_stitch.hip
# recompiled with extra flags.   tools/build_variants.sh w4:"-DSTITCH_WAVES=4" w5:"-DSTITCH_WAVES=5"
set -e
cd "$(dirname "$0")/.."
make -s engine
mkdir -p star_amd/lib/variants
for spec in "$@"; do
  tag=${spec%%:*}; flags=${spec#*:}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-result -fno-unroll-loops $flags -c star_amd/csrc/engine/k_stitch.hip -o star_amd/lib/variants/k_stitch_$tag.o
  objs=$(ls star_amd/lib/obj/prod/*.o | grep -v k_stitch.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs star_amd/lib/variants/k_stitch_$tag.o -o star_amd/lib/variants/libstaramd_$tag.so
  rm star_amd/lib/variants/k_stitch_$tag.o
  echo "built variant $tag ($flags)"
done
