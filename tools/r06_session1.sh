#!/bin/bash
# round 6, GPU session 1: leaf early exits + prune rule, A/B on one box, then the parity suite of the engine
cd ${GRAFT_REPO_ROOT:-.}
V=star_amd/lib/variants
bash tools/session.sh ab r06s1 "new|-|" "r5|$V/libstaramd_r5.so|" "early|$V/libstaramd_early.so|" "prof|$V/libstaramd_prof.so|" "profr5|$V/libstaramd_profr5.so|"
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/r06s1/pytest_parity.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/r06s1/pytest_parity.log
