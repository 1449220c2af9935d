#!/bin/bash
# round 5, GPU session 8: staramd_map_begin / _wait / _end in the front end (kernels of batch k+1 beside the result copy of batch k) against one blocking call per batch,
# alternating; the GPU tests that changed; can two ranks share one MI355X over RCCL (tools/nccl_one_gpu_probe.py)
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r05s8; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_twins.py tests/test_gpu_large_index.py -m gpu -x -q > $O/pytest_twins_large.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest_twins_large.log
timeout 120 python tools/nccl_one_gpu_probe.py > $O/nccl_one_gpu_probe.txt 2>&1; echo "probe rc $?"; grep -E "rank [01]:" $O/nccl_one_gpu_probe.txt | cut -c1-300
run() { tag=$1; shift
env "$@" STARAMD_PIPELINE_LOG=$PWD/$O/plog_$tag.txt timeout 500 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs --no-exclusive > $O/b_$tag.json 2> $O/b_$tag.err
python - <<PY
import json
d = json.loads(open("$O/b_$tag.json").read().strip().splitlines()[-1]); e = json.load(open(d["extra"])); p = e["pipeline"]
print("%-14s value %.3f ms/step %.1f device ms %.1f (device only %.2f M/s) map calls ms %.1f | fast %s" % ("$tag", d["value"], d["ms_per_step"], d["roofline"]["per_kernel_ms"]["device_total"], d.get("device_only_value") or 0, p["map_batch_call_s"] / 20 * 1e3, p["fast_path_batches"]))
PY
}
run overlapped X=1
run blocking STARAMD_NO_OVERLAP=1
run overlapped_b X=1
run blocking_b STARAMD_NO_OVERLAP=1
