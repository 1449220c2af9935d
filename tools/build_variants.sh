#!/bin/bash
# engine variants for A/B runs on the GPU box: star_amd/lib/variants/libstaramd_<tag>.so, each = the production objects with ONE kernel file
# recompiled with extra flags.   tools/build_variants.sh w4:k_stitch:"-DSTITCH_WAVES=4" s6:k_seed:"-DSEED_WAVES=6"
set -e
cd "$(dirname "$0")/.."
make -s engine
mkdir -p star_amd/lib/variants
for spec in "$@"; do
  tag=${spec%%:*}; rest=${spec#*:}; file=${rest%%:*}; flags=${rest#*:}
  extra=""; if [ "$file" = k_stitch ]; then extra="-fno-unroll-loops"; fi
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-result $extra $flags -c star_amd/csrc/engine/$file.hip -o star_amd/lib/variants/${file}_$tag.o
  objs=$(ls star_amd/lib/obj/prod/*.o | grep -v "/$file.o")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs star_amd/lib/variants/${file}_$tag.o -o star_amd/lib/variants/libstaramd_$tag.so
  rm star_amd/lib/variants/${file}_$tag.o
  echo "built variant $tag ($file $flags)"
done
