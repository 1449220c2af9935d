#!/bin/bash
# engine variants for A/B runs on the GPU box: star_amd/lib/variants/libstaramd_<tag>.so, each = the production objects with ONE kernel file
# (or several: k_window+engine) recompiled with extra flags.   tools/build_variants.sh w4:k_stitch:"-DSTITCH_WAVES=4" s6:k_seed:"-DSEED_WAVES=6" pool:k_window+engine:"-DWIN_POOL_ROWS=64"
set -e
cd "$(dirname "$0")/.."
make -s engine
mkdir -p star_amd/lib/variants
for spec in "$@"; do
  tag=${spec%%:*}; rest=${spec#*:}; file=${rest%%:*}; flags=${rest#*:}
  objs=$(ls star_amd/lib/obj/prod/*.o); mine=""
  for f in ${file//+/ }; do
    extra=""; if [ "$f" = k_stitch ]; then extra="-fno-unroll-loops"; fi
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-result $extra $flags -c star_amd/csrc/engine/$f.hip -o star_amd/lib/variants/${f}_$tag.o
    objs=$(echo "$objs" | grep -v "/$f.o"); mine="$mine star_amd/lib/variants/${f}_$tag.o"
  done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs $mine -o star_amd/lib/variants/libstaramd_$tag.so
  rm $mine
  echo "built variant $tag ($file $flags)"
done
