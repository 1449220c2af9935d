#!/bin/bash
# end-of-round measurement on the GPU box (run from the repo root):  tools/final_session.sh <tag>
#   (before: tools/measure_session.sh <tag> 3100 3 "stats sq1 fetch write" -> profiles/r03_3100mb_pmc_*.summary.json, the PMC passes of this engine build)
#   1. rocprofv3 --kernel-trace --stats of the bench command with ONE engine context (every launch has the GPU to itself)
#   2. profiles/r03_pmc_hbm_traffic.json from the PMC passes + those kernel times (read by bench.py for roofline.traffic / roofline.issue)
#   3. the default bench line (cpu_baseline, full-size parity, extra legs, sweep, two-pass) on the cached genome
R=$PWD; TAG=${1:-final}; O=$R/gpurun_out/$TAG; mkdir -p $O
STARAMD_CONTEXTS_PER_GPU=1 bash tools/kstats.sh gpurun_out/$TAG/k1 3100 STARAMD_CONTEXTS_PER_GPU=1 2>&1 | tail -16
mkdir -p $O/pmc; for f in fetch write sq1; do cp profiles/r03_3100mb_pmc_$f.summary.json $O/pmc/pmc_$f.summary.json; done; cp $O/k1/kernel_stats.csv $O/pmc/kernel_stats.csv
python tools/make_traffic_json.py $O/pmc 3100 400000 $R/profiles/r03_pmc_hbm_traffic.json && cp $R/profiles/r03_pmc_hbm_traffic.json $O/
cd $R
timeout 1500 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
tail -c 1200 $O/bench.json; grep "bench:" $O/bench.err | tail -40
