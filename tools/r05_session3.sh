#!/bin/bash
# round 5, GPU session 3: the seed stage with a lane per UNIT of the search schedule (k_seed_plan / k_seed_units / k_seed_merge) against a lane per read (k_seed_search over every read =
# rounds 1-4), same library, alternating; then how the unit kernel answers to fewer resident lanes.  Result buffers of every variant must equal those of the first.
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r05s3; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > $O/pytest_parity.log 2>&1; echo "parity rc $?"; tail -2 $O/pytest_parity.log
STARAMD_VERBOSE=1 timeout 600 python tools/ab_kernels.py --genome-mb 3100 --batches 3 --repeat 2 --rounds 2 --out $O/ab.json \
  "lane_per_read|-|STARAMD_SEED_UNITS=0" \
  "lane_per_unit|-|" \
  "lane_per_unit_6_blocks|-|STARAMD_SEED_UNIT_LANES=393216" \
  "lane_per_unit_4_blocks|-|STARAMD_SEED_UNIT_LANES=262144" \
  "lane_per_unit_slots4|-|STARAMD_SEED_SLOT_LIMIT=4" > $O/ab.txt 2> $O/ab.err
echo "ab rc $?"
grep -v "counts per pair" $O/ab.txt | tail -12
grep "seed units" $O/ab.err | sort | uniq -c | head -5
tail -2 $O/ab.err
