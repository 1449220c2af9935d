#!/bin/bash
# round 5, GPU session 1: the kernels with the wave index declared uniform (dev.h WAVE_INDEX) against the kernels of round 4 (-DWAVE_INDEX_PLAIN), and what the registers it frees
# are worth in resident wavefronts.  Build the variants BEFORE the call (they travel with the snapshot):
#   tools/build_variants.sh r4:k_stitch+k_window:"-DWAVE_INDEX_PLAIN" st5:k_stitch:"-DSTITCH_WAVES=5" st4:k_stitch:"-DSTITCH_WAVES=4" \
#        st6:k_stitch:"-DSTITCH_WAVES=6" st8:k_stitch:"-DSTITCH_WAVES=8" w7:k_window:"-DWIN_WAVES=7" w8:k_window:"-DWIN_WAVES=8" st5w8:k_stitch+k_window:"-DSTITCH_WAVES=5 -DWIN_WAVES=8"
# (k_stitch_win: 106 VGPRs as it is, 96 + 4 spilled at 5 wavefronts per SIMD, 80 + 20 at 6, 64 + 35 at 8; the lean launch's LDS slice decides how many of them fit:
#  depth 20 / arena 3072 = 7.6 KB -> 5 (the default), depth 16 / 2048 = 6.0 KB -> 6, depth 12 / 1024 = 4.4 KB -> 8; STARAMD_VERBOSE=1 prints how many items a launch hands on:
#  at 2x101 a pair has ~17 seeds, most of them in its best window -- a lean launch of 9 frames hands half of its items on)
# every variant's resource figures: tools/isa_stats.sh k_stitch -DSTITCH_WAVES=5 ...   (a variant that does not show ~106 / 96 VGPRs for k_stitch_win has fallen back into the
# other regime of the register allocator: it flips on small edits -- check before spending GPU time)
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r05s1; mkdir -p $O
V=star_amd/lib/variants
timeout 600 python tools/ab_kernels.py --genome-mb 3100 --batches 3 --repeat 2 --rounds 2 --out $O/ab.json \
  "r4_kernels|$V/libstaramd_r4.so|STARAMD_LEAN_DEPTH=0" \
  "uniform_same_occupancy|-|STARAMD_LEAN_DEPTH=0" \
  "uniform_lean20|-|" \
  "uniform_lean20_stitch5|$V/libstaramd_st5.so|" \
  "uniform_lean16_stitch5|$V/libstaramd_st5.so|STARAMD_LEAN_DEPTH=16" \
  "uniform_lean20_stitch4|$V/libstaramd_st4.so|" \
  "uniform_lean16_stitch6|$V/libstaramd_st6.so|STARAMD_LEAN_DEPTH=16 STARAMD_LEAN_ARENA=2048" \
  "uniform_lean20_arena2048_stitch6|$V/libstaramd_st6.so|STARAMD_LEAN_ARENA=2048" \
  "uniform_lean12_stitch8|$V/libstaramd_st8.so|STARAMD_LEAN_DEPTH=12 STARAMD_LEAN_ARENA=1024" \
  "uniform_win7_rows112|$V/libstaramd_w7.so|STARAMD_CAP_WINDOWS=112 STARAMD_LEAN_DEPTH=0" \
  "uniform_win8_rows96|$V/libstaramd_w8.so|STARAMD_CAP_WINDOWS=96 STARAMD_LEAN_DEPTH=0" \
  "uniform_all|$V/libstaramd_st5w8.so|STARAMD_CAP_WINDOWS=96" > $O/ab.txt 2> $O/ab.err
echo "ab rc $?"
grep -v "counts per pair" $O/ab.txt | tail -20
tail -3 $O/ab.err
