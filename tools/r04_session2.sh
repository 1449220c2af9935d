#!/bin/bash
# round 4, GPU session 2: extension profiles A/B, k_windows occupancy variants, section profile of the stitch walk, host timing at 1M batches
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/s2; mkdir -p $O
V=star_amd/lib/variants
timeout 900 python tools/ab_kernels.py --genome-mb 3100 --batches 3 --repeat 2 --rounds 2 --out $O/ab.json \
  "oldstitch|$V/libstaramd_oldstitch.so|" "new|-|" "prof0|-|STARAMD_EXT_PROFILES=0" \
  "w6c128|$V/libstaramd_w6.so|STARAMD_CAP_WINDOWS=128" "w6c160|$V/libstaramd_w6.so|STARAMD_CAP_WINDOWS=160" "w6c192|$V/libstaramd_w6.so|STARAMD_CAP_WINDOWS=192" "w6c96|$V/libstaramd_w6.so|STARAMD_CAP_WINDOWS=96" \
  "w7c128|$V/libstaramd_w7.so|STARAMD_CAP_WINDOWS=128" "nomid128|$V/libstaramd_w6.so|STARAMD_CAP_WINDOWS=128 STARAMD_CAP_WINDOWS_MID=0" > $O/ab.txt 2> $O/ab.err
tail -22 $O/ab.txt
timeout 600 python tools/ab_kernels.py --genome-mb 3100 --batches 2 --repeat 2 --rounds 1 --out $O/prof.json \
  "profile|star_amd/lib/libstaramd_profile.so|" "profile_nolane|star_amd/lib/libstaramd_profile.so|STARAMD_LANE=0" "profile_noext|star_amd/lib/libstaramd_profile.so|STARAMD_LANE=0 STARAMD_EXT_PROFILES=0" > $O/prof.txt 2> $O/prof.err
cat $O/prof.txt
for r in 400000 1000000; do
  STARAMD_HOST_TIMING=1 timeout 400 python bench.py --steps 16 --warmup 4 --reads $r --no-cpu-baseline --no-extra-legs --no-exclusive > $O/bs_$r.json 2> $O/bs_$r.err
  python - <<PY
import json
try:
    d = json.loads(open("$O/bs_$r.json").read().strip().splitlines()[-1]); print("batch $r: value %.3f  ms/step %.1f" % (d["value"], d["ms_per_step"]))
except Exception as e:
    print("batch $r FAILED", e); print(open("$O/bs_$r.err").read()[-800:])
PY
done
cp /dev/shm/star_amd_bench/bench_extra.json $O/bench_extra_1M.json 2>/dev/null
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "shadow or buffers" > $O/pytest_subset.log 2>&1; tail -5 $O/pytest_subset.log
