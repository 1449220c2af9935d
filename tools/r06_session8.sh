#!/bin/bash
# round 6, GPU session 8: (a) result copy beside the kernels of the next batch (begin / wait / end) against one blocking call per batch, alternating; (b) the measurement session:
# plain bench + rocprofv3 --stats + PMC passes (SQ, FETCH_SIZE, WRITE_SIZE) + dependent-gather ceiling + FETCH_SIZE calibrated on that pattern
cd ${GRAFT_REPO_ROOT:-.}
bash tools/session.sh e2e r06s8 "blocking_a" "overlap_a STARAMD_OVERLAP_COPIES=1" "blocking_b" "overlap_b STARAMD_OVERLAP_COPIES=1" "blocking_c" "overlap_c STARAMD_OVERLAP_COPIES=1"
bash tools/session.sh measure r06m
ls gpurun_out/r06m | head -30
