#!/bin/bash
# round 5, GPU session 11: the cooperative stitch walk at FOUR blocks per CU (main launch: stack of 33 frames = LDS slice of 10 KB, arena as before; windows of more seeds to a
# full-depth launch), and at five (stack of 20 / 16 frames, kernel held to 96 registers)
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r05s11; mkdir -p $O
V=star_amd/lib/variants
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > $O/pytest_parity.log 2>&1; echo "parity rc $?"; tail -2 $O/pytest_parity.log
STARAMD_VERBOSE=1 timeout 600 python tools/ab_kernels.py --genome-mb 3100 --batches 3 --repeat 2 --rounds 2 --out $O/ab.json \
  "main_33|-|" \
  "one_launch_full_depth|-|STARAMD_MAIN_DEPTH=0" \
  "main_26|-|STARAMD_MAIN_DEPTH=26" \
  "main_20|-|STARAMD_MAIN_DEPTH=20" \
  "main_20_regs96|$V/libstaramd_st5.so|STARAMD_MAIN_DEPTH=20" \
  "main_16_regs96|$V/libstaramd_st5.so|STARAMD_MAIN_DEPTH=16" \
  "main_12_regs80|$V/libstaramd_st6.so|STARAMD_MAIN_DEPTH=12 STARAMD_STITCH_ARENA=2048" > $O/ab.txt 2> $O/ab.err
echo "ab rc $?"
grep -v "counts per pair" $O/ab.txt | tail -14
grep "main launch" $O/ab.err | sort | uniq -c | head; grep "stitch work items" $O/ab.err | sort | uniq -c | cut -c1-250 | head -8
tail -2 $O/ab.err
