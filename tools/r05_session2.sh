#!/bin/bash
# round 5, GPU session 2: (a) the GPU suite on the merged tree, (b) the end-to-end bench with the per-stage thread-CPU clocks, the fast-path counters and the
# pipeline event log (STARAMD_PIPELINE_LOG: where a batch waits), twice: as shipped, and with every stage thread count halved (16 CPUs, ~40 threads wanting them)
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r05s2; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest_gpu.log
run() { tag=$1; shift
env "$@" STARAMD_PIPELINE_LOG=$PWD/$O/plog_$tag.txt STARAMD_HOST_TIMING=1 timeout 500 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs --no-exclusive > $O/b_$tag.json 2> $O/b_$tag.err
python - <<PY
import json
d = json.loads(open("$O/b_$tag.json").read().strip().splitlines()[-1]); e = json.load(open(d["extra"])); p = e["pipeline"]
print("%-14s value %.3f ms/step %.1f device ms %.1f map_batch ms %.1f | cpu us/pair %s | fast %s | post-map %s" % ("$tag", d["value"], d["ms_per_step"], d["roofline"]["per_kernel_ms"]["device_total"], p["map_batch_call_s"] / 20 * 1e3,
      p["cpu_us_per_pair_by_stage"], p["fast_path_batches"], {k: round(v, 2) for k, v in p["postmap_whole_run_s"].items()}))
PY
grep -v "^  parse\|^  emit\|^bench:" $O/b_$tag.err | tail -3
}
run shipped X=1
run pwrite STARAMD_WRITER_MMAP=0
run shipped_b X=1
run writer8 STARAMD_WRITER_THREADS=8
nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null
