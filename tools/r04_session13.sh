#!/bin/bash
# round 4, GPU session 13: are the result buffers of two runs / of the pruning levels identical apart from maxScoreMate (a lower bound under resultSelect = 1)?
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/s13; mkdir -p $O
timeout 600 python tools/ab_kernels.py --genome-mb 3100 --batches 2 --repeat 2 --rounds 2 --out $O/ab.json "base|-|" "prune3|-|STARAMD_PRUNE=3" "prune0|-|STARAMD_PRUNE=0" "nolane|-|STARAMD_LANE=0" > $O/ab.txt 2> $O/ab.err
grep -v "counts per pair" $O/ab.txt | tail -8; tail -3 $O/ab.err
