#!/bin/bash
# round 5, GPU session 4: the seed stage in ROUNDS (k_seed_lookup / k_seed_bisect: one search of every unit per round, the bisections of a round sorted by interval length) against
# the units with their searches back to back (k_seed_units) and a lane per read (k_seed_search); number of rounds before the tail kernel.
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r05s4; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > $O/pytest_parity.log 2>&1; echo "parity rc $?"; tail -2 $O/pytest_parity.log
STARAMD_VERBOSE=1 timeout 600 python tools/ab_kernels.py --genome-mb 3100 --batches 3 --repeat 2 --rounds 2 --out $O/ab.json \
  "lane_per_read|-|STARAMD_SEED_UNITS=0" \
  "units_back_to_back|-|STARAMD_SEED_UNITS=1" \
  "rounds_6|-|" \
  "rounds_3|-|STARAMD_SEED_ROUNDS=3" \
  "rounds_4|-|STARAMD_SEED_ROUNDS=4" \
  "rounds_8|-|STARAMD_SEED_ROUNDS=8" \
  "rounds_12|-|STARAMD_SEED_ROUNDS=12" \
  "rounds_6_half_lanes|-|STARAMD_SEED_UNIT_LANES=262144" > $O/ab.txt 2> $O/ab.err
echo "ab rc $?"
grep -v "counts per pair" $O/ab.txt | tail -16
grep "seed units" $O/ab.err | sort | uniq -c | head -5
tail -2 $O/ab.err
