#!/bin/bash
# round 4, GPU session 4: single-mate leaves skipped (STARAMD_PRUNE bit 2), sjdb hash, lane class with fewer leaves; then one complete default bench run (line + 10 M-pair parity)
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/s4; mkdir -p $O
timeout 900 python tools/ab_kernels.py --genome-mb 3100 --batches 3 --repeat 2 --rounds 2 --out $O/ab.json \
  "prune3|-|STARAMD_PRUNE=3" "prune7|-|" "prune3_nohash|-|STARAMD_PRUNE=3 STARAMD_NO_SJDB_HASH=1" "p7c6|-|STARAMD_LANE_CLASS=6" "p7c7|-|STARAMD_LANE_CLASS=7" "p7c4|-|STARAMD_LANE_CLASS=4" "p7nolane|-|STARAMD_LANE=0" > $O/ab.txt 2> $O/ab.err
grep -v "counts per pair" $O/ab.txt | tail -14; grep "counts per pair" $O/ab.txt | head -2 | cut -c1-700
timeout 1500 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 3000 $O/bench_default.json; grep "^bench:" $O/bench_default.err | tail -12
cp /dev/shm/star_amd_bench/bench_extra.json $O/ 2>/dev/null
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "forced or buffers" > $O/pytest_subset.log 2>&1; tail -4 $O/pytest_subset.log
