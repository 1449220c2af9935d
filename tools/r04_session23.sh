#!/bin/bash
# round 4, session 23 (no GPU work, the last seconds of the budget): the committed defaults (slots last-in-first-out, 8 read slices at 16 threads), host stages alone, mapper blocked 52 ms per batch
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/s23
STARAMD_HOST_TIMING=1 timeout 14 python tools/host_bench.py --contexts 1 --block 400000 --repeat 20 --threads 16 --device-ms 52 > gpurun_out/s23/default.out 2> gpurun_out/s23/default.err
tail -1 gpurun_out/s23/default.out | cut -c1-420; grep -m3 "fill mate" gpurun_out/s23/default.err
