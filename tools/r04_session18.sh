#!/bin/bash
# round 4, GPU session 18 (the last 5 GPU-minutes): k_windows with the seed lists in LDS (win_pool.h) against the product kernel
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/s18; mkdir -p $O
V=star_amd/lib/variants
timeout 255 python tools/ab_kernels.py --genome-mb 3100 --batches 3 --repeat 2 --rounds 2 --out $O/ab.json \
  "base|-|" "pool|$V/libstaramd_pool.so|" "pool_list12|$V/libstaramd_pool12.so|" "pool_lib_switch_off|$V/libstaramd_pool.so|STARAMD_WIN_POOL=0" > $O/ab.txt 2> $O/ab.err
echo "ab rc $?"
grep -v "counts per pair" $O/ab.txt | tail -10
tail -3 $O/ab.err
