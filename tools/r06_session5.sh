#!/bin/bash
# round 6, GPU session 5: the whole GPU suite on the tree with the lane kernel fixed and the three no-gain latency changes taken out; then the default bench
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r06s5
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r06s5/pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -4 gpurun_out/r06s5/pytest_gpu.log
timeout 1700 python bench.py > gpurun_out/r06s5/bench_line.json 2> gpurun_out/r06s5/bench.err; echo "bench rc $?"; cut -c1-1500 gpurun_out/r06s5/bench_line.json
cp /dev/shm/star_amd_bench/bench_extra.json gpurun_out/r06s5/bench_extra.json 2>/dev/null
