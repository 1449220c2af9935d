#!/bin/bash
# round 4, GPU session 11: deeper batch queue; one context vs two contexts taking turns
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/s11; mkdir -p $O
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
for rep in a b; do
run c1_q3_$rep X=1
run c1_q0_$rep STARAMD_EXTRA_SLOTS=0
run c2turns_q3_$rep STARAMD_CONTEXTS_PER_GPU=2 STARAMD_KERNEL_TURNS=1
run c2_q3_$rep STARAMD_CONTEXTS_PER_GPU=2
run c1_q6_$rep STARAMD_EXTRA_SLOTS=6
done
