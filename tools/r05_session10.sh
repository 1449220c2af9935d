#!/bin/bash
# round 5, GPU session 10: does k_stitch_win answer to a FOURTH block per CU now that its atomics do not queue?  (session 9: 2 blocks 26.4 ms, 3 blocks 19.3 -> T = 5.1 + 42.6 / blocks.)
# The LDS slice of a wavefront is 12.5 KB (3 blocks per CU); a record arena of 512 bytes instead of 3584 makes it 9.5 KB = 4 blocks (windows that record more re-walk with the arena in HBM).
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r05s10; mkdir -p $O
STARAMD_VERBOSE=1 timeout 600 python tools/ab_kernels.py --genome-mb 3100 --batches 3 --repeat 2 --rounds 2 --out $O/ab.json \
  "base|-|" \
  "arena_512|-|STARAMD_STITCH_ARENA=512" \
  "arena_512_3_blocks|-|STARAMD_STITCH_ARENA=512 STARAMD_STITCH_BLOCKS_PER_CU=3" \
  "arena_1024|-|STARAMD_STITCH_ARENA=1024" \
  "arena_256|-|STARAMD_STITCH_ARENA=256" > $O/ab.txt 2> $O/ab.err
echo "ab rc $?"
grep -v "counts per pair" $O/ab.txt | tail -10
grep "k_stitch_win .* blocks/CU" $O/ab.err | sort | uniq -c | head
tail -2 $O/ab.err
