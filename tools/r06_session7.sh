#!/bin/bash
# round 6, GPU session 7: which reads the lane kernel should take (cost class cap), 2x101 at 3.1 Gb and 1x50 at 12 Mb
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r06s7; mkdir -p $O
bash tools/session.sh ab r06s7 "c3|-|" "c4|-|STARAMD_LANE_CLASS=4" "c5|-|STARAMD_LANE_CLASS=5" "c6|-|STARAMD_LANE_CLASS=6" "c8|-|STARAMD_LANE_CLASS=8"
STARAMD_VERBOSE=1 timeout 600 python tools/ab_config1.py --batches 3 --repeat 2 --rounds 2 --out $O/abse.json "se5|-|STARAMD_LANE_CLASS=5" "se6|-|STARAMD_LANE_CLASS=6" "se7|-|STARAMD_LANE_CLASS=7" "se8|-|STARAMD_LANE_CLASS=8" "se10|-|STARAMD_LANE_CLASS=10" "se31|-|STARAMD_LANE_CLASS=31" > $O/abse.txt 2> $O/abse.err; echo "abse rc $?"; grep -v "counts per pair" $O/abse.txt | cut -c1-200 | tail -14
