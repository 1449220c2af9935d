#!/bin/bash
# round 4, session 22 (no GPU work; the last seconds of the round's GPU-box budget): where the host stages spend their time on the box's 16 CPUs -- STARAMD_HOST_TIMING lines of
# tools/host_bench.py (mapper blocked 52 ms per batch), with 4 and with 8 read slices per mate
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/s22; mkdir -p $O
STARAMD_HOST_TIMING=1 timeout 25 python tools/host_bench.py --contexts 1 --block 400000 --repeat 20 --threads 16 --device-ms 52 > $O/slices4.out 2> $O/slices4.err
STARAMD_HOST_TIMING=1 STARAMD_READ_SLICES=8 timeout 25 python tools/host_bench.py --contexts 1 --block 400000 --repeat 20 --threads 16 --device-ms 52 > $O/slices8.out 2> $O/slices8.err
tail -1 $O/slices4.out | cut -c1-400; tail -1 $O/slices8.out | cut -c1-400
