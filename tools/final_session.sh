#!/bin/bash
# end-of-round measurement on the GPU box (run from the repo root):  tools/final_session.sh <tag>
#   1. tools/measure_session.sh: short bench line, rocprofv3 kernel stats, FETCH_SIZE and WRITE_SIZE passes at the bench's own size
#   2. profiles/r03_pmc_hbm_traffic.json from the PMC passes (read by bench.py for roofline.traffic / roofline.issue when the engine sources match)
#   3. the default bench line (cpu_baseline, full-size parity, sweep, two-pass) on the cached genome
R=$PWD; TAG=${1:-final}; O=$R/gpurun_out/$TAG; mkdir -p $O
bash tools/measure_session.sh $TAG 3100 3 "stats sq1 fetch write" 2>&1 | tail -12
python tools/make_traffic_json.py $O 3100 400000 $R/profiles/r03_pmc_hbm_traffic.json && cp $R/profiles/r03_pmc_hbm_traffic.json $O/
cd $R
timeout 1500 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
tail -c 1500 $O/bench.json
