#!/bin/bash
# last GPU session of a round: the default bench run (CPU baseline, full-size parity, optional legs) and the whole GPU suite on the final tree
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/final; mkdir -p $O
timeout 1500 python bench.py > $O/bench_line.json 2> $O/bench.err; echo "bench rc $?"; cut -c1-600 $O/bench_line.json
cp /dev/shm/star_amd_bench/bench_extra.json $O/bench_extra.json 2>/dev/null
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest_gpu.log
