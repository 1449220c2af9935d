#!/bin/bash
# round 4, GPU session 19 (the last GPU-minutes of the round): section clocks of the FINAL kernels (profile build), and how each kernel's time answers to fewer resident
# wavefronts (blocks per CU): a kernel that waits on memory slows down in proportion, one that is short of issue slots does not
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/s19; mkdir -p $O
timeout 185 python tools/ab_kernels.py --genome-mb 3100 --batches 3 --repeat 2 --rounds 1 --out $O/ab.json \
  "base|-|" "profile|star_amd/lib/libstaramd_profile.so|" "win_5_blocks_per_cu|-|STARAMD_WIN_BLOCKS_PER_CU=5" "win_4_blocks_per_cu|-|STARAMD_WIN_BLOCKS_PER_CU=4" "win_3_blocks_per_cu|-|STARAMD_WIN_BLOCKS_PER_CU=3" \
  "stitch_2_blocks_per_cu|-|STARAMD_STITCH_BLOCKS_PER_CU=2" "stitch_1_block_per_cu|-|STARAMD_STITCH_BLOCKS_PER_CU=1" "seed_6_blocks_per_cu|-|STARAMD_SEED_LANES=393216" "seed_4_blocks_per_cu|-|STARAMD_SEED_LANES=262144" "base_again|-|" > $O/ab.txt 2> $O/ab.err
echo "ab rc $?"
grep -v "counts per pair" $O/ab.txt | tail -14 | cut -c1-1200
tail -2 $O/ab.err
