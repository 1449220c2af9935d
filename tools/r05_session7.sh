#!/bin/bash
# round 5, GPU session 7: the GPU suite and the hardware fuzzer on the tree with the counters in cache lines of their own and the unit mapping of the seed stage; end to end:
# one engine context against two contexts over one index whose kernel phases take turns (copies of one beside the kernels of the other), alternating
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r05s7; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest_gpu.log
timeout 300 python tests/tools/fuzz_engine.py 80 909 > $O/fuzz_engine_hardware.log 2>&1; echo "fuzz rc $?"; tail -2 $O/fuzz_engine_hardware.log
run() { tag=$1; shift
env "$@" STARAMD_PIPELINE_LOG=$PWD/$O/plog_$tag.txt timeout 500 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs --no-exclusive > $O/b_$tag.json 2> $O/b_$tag.err
python - <<PY
import json
d = json.loads(open("$O/b_$tag.json").read().strip().splitlines()[-1]); e = json.load(open(d["extra"])); p = e["pipeline"]
print("%-14s value %.3f ms/step %.1f device ms %.1f %s map_batch ms %.1f | cpu us/pair %s | fast %s" % ("$tag", d["value"], d["ms_per_step"], d["roofline"]["per_kernel_ms"]["device_total"], d["roofline"]["per_kernel_ms"], p["map_batch_call_s"] / 20 * 1e3,
      p["cpu_us_per_pair_by_stage"], p["fast_path_batches"]))
PY
}
run one_context X=1
run two_turns STARAMD_CONTEXTS_PER_GPU=2 STARAMD_KERNEL_TURNS=1
run one_context_b X=1
run two_turns_b STARAMD_CONTEXTS_PER_GPU=2 STARAMD_KERNEL_TURNS=1
run seed_lane_per_read STARAMD_SEED_UNITS=0
