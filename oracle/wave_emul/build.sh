#!/bin/bash
# oracle/_build/libstaramd_emul.so: the engine's kernel sources compiled for the host by the wavefront emulator (emu.h). Test infrastructure.
#   build.sh asan   builds oracle/_build/libstaramd_emul_asan.so with AddressSanitizer instead (every access of the kernels to the "device" buffers
#                   is bounds-checked; run with LD_PRELOAD=$(clang++ -print-file-name=libclang_rt.asan-x86_64.so), see tests/tools/emul_asan.sh)
set -e
cd "$(dirname "$0")/../.."
CL=${EMUL_CXX:-/opt/rocm/lib/llvm/bin/clang++}
FL="-x c++ -std=c++17 -O1 -g -fPIC -ffp-contract=off -Wno-unused-result -Wno-unknown-attributes -I oracle/wave_emul"
O=oracle/_build/emul; OUT=oracle/_build/libstaramd_emul.so; SAN=""
if [ "$1" = asan ]; then SAN="-fsanitize=address -fno-omit-frame-pointer"; FL="$FL $SAN"; O=oracle/_build/emul_asan; OUT=oracle/_build/libstaramd_emul_asan.so; fi
#   build.sh variant <tag> <defines>   e.g. build.sh variant w5 -DWIN_WAVES=5  ->  oracle/_build/libstaramd_emul_w5.so  (kernel variants behind compile-time switches)
if [ "$1" = variant ]; then FL="$FL $3"; O=oracle/_build/emul_$2; OUT=oracle/_build/libstaramd_emul_$2.so; SAN=" "; fi
mkdir -p $O
for f in k_window k_seed k_gather k_stitch_lane engine; do $CL $FL -c star_amd/csrc/engine/$f.hip -o $O/$f.o & done
$CL $FL -fno-unroll-loops -c star_amd/csrc/engine/k_stitch.hip -o $O/k_stitch.o &
$CL $FL -c star_amd/csrc/index/index_gpu.hip -o $O/index_gpu.o &
$CL -std=c++17 -O1 -g -fPIC $SAN -D_GNU_SOURCE -c oracle/wave_emul/emu.cpp -o $O/emu.o &
$CL -std=c++17 -O1 -g -fPIC $SAN -c oracle/wave_emul/emu_lds.cpp -o $O/emu_lds.o &
wait
for f in k_window k_seed k_gather k_stitch_lane engine k_stitch index_gpu emu emu_lds; do test -s $O/$f.o; done
$CL -shared -fPIC $SAN -shared-libsan $O/*.o -o $OUT -ldl
[ -n "$SAN" ] && exit 0
$CL -x c++ -std=c++17 -O1 -g -Wno-unknown-attributes -D_GNU_SOURCE -I oracle/wave_emul -I star_amd/csrc/engine -I include oracle/wave_emul/selftest.cpp oracle/wave_emul/emu.cpp oracle/wave_emul/emu_lds.cpp -o oracle/_build/wave_emul_selftest -ldl
