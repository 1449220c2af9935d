#!/bin/bash
# round 4, GPU session 1: (1) k_windows A/B at 3.1 Gb in one process, (2) end-to-end batch-size A/B, (3) the new GPU tests
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/s1; mkdir -p $O
V=star_amd/lib/variants
timeout 900 python tools/ab_kernels.py --genome-mb 3100 --batches 3 --repeat 2 --rounds 2 --out $O/ab_win.json \
  "old|$V/libstaramd_oldwin.so|" "new|-|" "w5|$V/libstaramd_w5.so|" "w6c128|$V/libstaramd_w6.so|STARAMD_CAP_WINDOWS=128" "w8c128|$V/libstaramd_w8.so|STARAMD_CAP_WINDOWS=128" \
  "w5c192|$V/libstaramd_w5.so|STARAMD_CAP_WINDOWS=192" "newc128|-|STARAMD_CAP_WINDOWS=128" "lane6|-|STARAMD_LANE_CLASS=6" > $O/ab_win.txt 2> $O/ab_win.err
tail -20 $O/ab_win.txt
for r in 400000 1000000 600000 800000; do
  timeout 400 python bench.py --steps 10 --warmup 2 --reads $r --no-cpu-baseline --no-extra-legs > $O/bs_$r.json 2> $O/bs_$r.err
  python - <<PY
import json
try:
    d = json.loads(open("$O/bs_$r.json").read().strip().splitlines()[-1]); print("batch $r: value %.3f  %s" % (d["value"], d["roofline"]["per_kernel_ms"]))
except Exception as e:
    print("batch $r FAILED", e); print(open("$O/bs_$r.err").read()[-800:])
PY
done
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -q -x -k "pe300 or window_overflow or lane_off or 73" > $O/pytest_subset.log 2>&1; tail -5 $O/pytest_subset.log
