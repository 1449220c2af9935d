#!/bin/bash
# round 5, GPU session 9: the wave-parallel seed merge (default now) and cheap knob sweeps on the tree with the counters spread: lane-kernel class cap, light-read estimate
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r05s9; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > $O/pytest_parity.log 2>&1; echo "parity rc $?"; tail -2 $O/pytest_parity.log
timeout 300 python tests/tools/fuzz_engine.py 60 1234 > $O/fuzz_engine_hardware.log 2>&1; echo "fuzz rc $?"; tail -1 $O/fuzz_engine_hardware.log
timeout 600 python tools/ab_kernels.py --genome-mb 3100 --batches 3 --repeat 2 --rounds 2 --out $O/ab.json \
  "base|-|" \
  "seed_lane_per_read|-|STARAMD_SEED_UNITS=0" \
  "lane_class_4|-|STARAMD_LANE_CLASS=4" \
  "lane_class_5|-|STARAMD_LANE_CLASS=5" \
  "lane_class_2|-|STARAMD_LANE_CLASS=2" \
  "light_est_16k|-|STARAMD_LIGHT_EST=16384" \
  "light_est_256k|-|STARAMD_LIGHT_EST=262144" \
  "stitch_2_blocks|-|STARAMD_STITCH_BLOCKS_PER_CU=2" \
  "win_5_blocks|-|STARAMD_WIN_BLOCKS_PER_CU=5" > $O/ab.txt 2> $O/ab.err
echo "ab rc $?"
grep -v "counts per pair" $O/ab.txt | tail -18
tail -2 $O/ab.err
