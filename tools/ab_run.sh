#!/bin/bash
# A/B of engine variants / environment knobs on the GPU box (same cached workload for all):
#   tools/ab_run.sh <outdir> "<tag>|<variant or ->|<ENV=VAL ...>" ...
R=$PWD; O=$1; shift; mkdir -p $O
export STARAMD_BENCH_GENOME_MB=${AB_GENOME_MB:-400}
cp star_amd/lib/libstaramd.so /tmp/libstaramd_prod.so
for spec in "$@"; do
  IFS='|' read -r tag var envs <<< "$spec"
  if [ "$var" = "-" ] || [ -z "$var" ]; then cp /tmp/libstaramd_prod.so star_amd/lib/libstaramd.so; else cp star_amd/lib/variants/libstaramd_$var.so star_amd/lib/libstaramd.so; fi
  env $envs timeout 300 python bench.py --steps ${AB_STEPS:-3} --warmup 1 --no-cpu-baseline --no-sweep --no-two-pass --no-extra-legs > $O/$tag.json 2> $O/$tag.err
  python - <<PY
import json
try:
    d = json.load(open("$O/$tag.json")); k = d["roofline"]["per_kernel_ms"]; c = d.get("counters_per_pair", {})
    print("%-24s value %.3f  stitch %.1f  windows %.1f  seed %.1f  redecide %.1f  total %.1f  lane %.3f pruned %.2f" % ("$tag", d["value"], k["k_stitch_win"], k["k_windows"], k["k_seed_search"], k["k_stitch_verify+replay+finish"], k["device_total"], c.get("nLaneItems", 0), c.get("nPrunedWin", 0)))
except Exception as e:
    print("$tag FAILED", e); print(open("$O/$tag.err").read()[-600:])
PY
done
cp /tmp/libstaramd_prod.so star_amd/lib/libstaramd.so
