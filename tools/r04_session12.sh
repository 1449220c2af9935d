#!/bin/bash
# round 4, GPU session 12: register budget of k_stitch_win (2 waves per SIMD without spills / 4 with more), two bisection probes in flight in the seed search
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/s12; mkdir -p $O
V=star_amd/lib/variants
timeout 900 python tools/ab_kernels.py --genome-mb 3100 --batches 3 --repeat 2 --rounds 2 --out $O/ab.json \
  "base|-|" "st2|$V/libstaramd_st2.so|" "st4|$V/libstaramd_st4.so|" "seed2|$V/libstaramd_seed2.so|" > $O/ab.txt 2> $O/ab.err
grep -v "counts per pair" $O/ab.txt | tail -8
