#!/bin/bash
# round 4, GPU session 14: upload of the next batch beside the kernels of the current one (staramd_prefetch_batch): end to end on / off, CLI tests, then the default bench
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/s14; mkdir -p $O
run() { tag=$1; shift
  env "$@" timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs --no-exclusive ${EXTRA:-} > $O/b_$tag.json 2> $O/b_$tag.err
  python - <<PY
import json
try:
    d = json.loads(open("$O/b_$tag.json").read().strip().splitlines()[-1]); e = json.load(open(d["extra"]))
    print("%-14s value %.3f  ms/step %.1f  parse %.1f convert %.1f emit %.1f M/s  kernels %.1f ms" % ("$tag", d["value"], d["ms_per_step"], e["pipeline"]["parse_Mreads_s"], e["pipeline"]["convert_Mreads_s"], e["pipeline"]["postmap_write_Mreads_s"], d["roofline"]["per_kernel_ms"]["device_total"]))
except Exception as ex:
    print("$tag FAILED", ex); print(open("$O/b_$tag.err").read()[-600:])
PY
}
run prefetch_a X=1
run noprefetch_a STARAMD_PREFETCH=0
run prefetch_b X=1
run noprefetch_b STARAMD_PREFETCH=0
timeout 900 python -m pytest tests/test_gpu_twins.py tests/test_two_pass.py tests/test_chimeric.py -m gpu -q -x -n 4 > $O/pytest_cli.log 2>&1; tail -3 $O/pytest_cli.log
timeout 1500 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc $?"; tail -c 1500 $O/bench.json; echo
cp /dev/shm/star_amd_bench/bench_extra.json $O/ 2>/dev/null
