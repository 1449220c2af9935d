#!/bin/bash
# round 4, GPU session 10: do the batch copies overlap with kernels when they go through the SDMA engines instead of blit kernels?
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/s10; mkdir -p $O
run() { tag=$1; shift
  env "$@" timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs --no-exclusive ${EXTRA:-} > $O/b_$tag.json 2> $O/b_$tag.err
  python - <<PY
import json
try:
    d = json.loads(open("$O/b_$tag.json").read().strip().splitlines()[-1]); e = json.load(open(d["extra"]))
    print("%-22s value %.3f  ms/step %.1f  parse %.1f convert %.1f emit %.1f M/s  kernels %.1f ms" % ("$tag", d["value"], d["ms_per_step"], e["pipeline"]["parse_Mreads_s"], e["pipeline"]["convert_Mreads_s"], e["pipeline"]["postmap_write_Mreads_s"], d["roofline"]["per_kernel_ms"]["device_total"]))
except Exception as ex:
    print("$tag FAILED", ex); print(open("$O/b_$tag.err").read()[-600:])
PY
}
run c1 X=1
run c1_sdma HSA_ENABLE_SDMA=1
run c2turns_sdma HSA_ENABLE_SDMA=1 STARAMD_CONTEXTS_PER_GPU=2 STARAMD_KERNEL_TURNS=1
run c2_sdma HSA_ENABLE_SDMA=1 STARAMD_CONTEXTS_PER_GPU=2
run c1_nosdma HSA_ENABLE_SDMA=0
run c2turns X=1 STARAMD_CONTEXTS_PER_GPU=2 STARAMD_KERNEL_TURNS=1
cd /tmp && export TMPDIR=/tmp
HSA_ENABLE_SDMA=1 STARAMD_CONTEXTS_PER_GPU=2 STARAMD_KERNEL_TURNS=1 timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/trace -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-extra-legs --no-exclusive > $GRAFT_REPO_ROOT/$O/bench_traced.json 2> $GRAFT_REPO_ROOT/$O/bench_traced.err
cd $GRAFT_REPO_ROOT
python tools/timeline.py $O/trace 0.4 > $O/timeline.txt 2>&1; cat $O/timeline.txt
find $O/trace -name "*.csv" -size +20M -delete
