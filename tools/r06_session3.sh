#!/bin/bash
# round 6, GPU session 3: junction streams in registers, preloaded extension chunks, window prefetch (A/B on one box); shadow validation + parity; result copy beside the next batch's kernels on reserved CUs
cd ${GRAFT_REPO_ROOT:-.}
V=star_amd/lib/variants
bash tools/session.sh ab r06s3 "new|-|" "base|$V/libstaramd_base.so|" "noj|$V/libstaramd_noj.so|" "nopre|$V/libstaramd_nopre.so|" "nopf|$V/libstaramd_nopf.so|" "prof|$V/libstaramd_prof.so|"
grep "coopStitch by kind\|profile (k" gpurun_out/r06s3/ab.txt | head -2
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/r06s3/pytest_parity.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/r06s3/pytest_parity.log
STARAMD_VERBOSE=1 bash tools/session.sh e2e r06s3 "blocking" "cu2 STARAMD_COPY_CUS=2 STARAMD_OVERLAP_COPIES=1" "cu4 STARAMD_COPY_CUS=4 STARAMD_OVERLAP_COPIES=1" "ovl_nomask STARAMD_OVERLAP_COPIES=1" "blocking_b" "cu8 STARAMD_COPY_CUS=8 STARAMD_OVERLAP_COPIES=1"
grep -h "confined\|refused" gpurun_out/r06s3/b_*.err | sort | uniq -c
