#!/bin/bash
# round 6, GPU session 6: (a) keys beside the suffix array (one gather per bisection probe) against the packed array + genome, at 8 / 7 / 6 wavefronts per SIMD;
# (b) the 100 Mb index of the sweep (stitch 112 ms per batch): counters + section profile; (c) config 1 (1x50 single-end): which reads the lane kernel should take; (d) parity with the keys on
cd ${GRAFT_REPO_ROOT:-.}
V=star_amd/lib/variants
bash tools/session.sh ab r06s6 "keys|-|" "nokeys|-|STARAMD_SA_KEYS=0" "keys_s7|$V/libstaramd_s7.so|" "keys_s6|$V/libstaramd_s6.so|" "nokeys_s7|$V/libstaramd_s7.so|STARAMD_SA_KEYS=0"
grep "keys beside" gpurun_out/r06s6/ab.err | sort | uniq -c | head -3
O=gpurun_out/r06s6
STARAMD_VERBOSE=1 timeout 600 python tools/ab_kernels.py --genome-mb 100 --batches 3 --repeat 2 --rounds 1 --out $O/ab100.json "mb100|-|" "mb100_prof|$V/libstaramd_prof.so|" "mb100_noprune|-|STARAMD_PRUNE=0" > $O/ab100.txt 2> $O/ab100.err; echo "ab100 rc $?"; cut -c1-1500 $O/ab100.txt | tail -12
STARAMD_VERBOSE=1 timeout 600 python tools/ab_config1.py --batches 3 --repeat 2 --rounds 2 --out $O/abse.json "se|-|" "se_nolane|-|STARAMD_LANE=0" "se_class1|-|STARAMD_LANE_CLASS=1" "se_class2|-|STARAMD_LANE_CLASS=2" "se_class5|-|STARAMD_LANE_CLASS=5" "se_prof|$V/libstaramd_prof.so|" > $O/abse.txt 2> $O/abse.err; echo "abse rc $?"; cut -c1-1500 $O/abse.txt | tail -16
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_large_index.py -m gpu -x -q > $O/pytest_parity.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest_parity.log
