#!/bin/bash
# register / scratch / LDS / occupancy figures of the kernels of one engine source file, as the compiler reports them
#   tools/kres.sh k_window [extra hipcc flags]
f=$1; shift
extra=""; if [ "$f" = k_stitch ]; then extra="-fno-unroll-loops -DSTITCH_WAVES=${STITCH_WAVES:-3}"; fi
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-result $extra "$@" -Rpass-analysis=kernel-resource-usage -c star_amd/csrc/engine/$f.hip -o /tmp/kres_$$.o 2>&1 \
  | grep -E "Function Name|VGPRs:|SGPRs:|ScratchSize|Occupancy|LDS Size|VGPRs Spill" | sed 's/.*remark: //' | paste - - - - - - - | sed 's/\[-Rpass-analysis=kernel-resource-usage\]//g; s/  */ /g'
rm -f /tmp/kres_$$.o
