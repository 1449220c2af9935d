#!/bin/bash
# round 4, final GPU session: rocprofv3 kernel stats + PMC passes (SQ, FETCH_SIZE, WRITE_SIZE: separate runs) on the headline workload -> profiles/r04_pmc_hbm_traffic.json,
# then the default bench line (reads that file for roofline.traffic / issue), then the whole GPU suite
cd ${GRAFT_REPO_ROOT:-.}
R=$PWD; O=$R/gpurun_out/r04f; mkdir -p $O
bash tools/measure_session.sh r04f 3100 3 "stats sq1 fetch write" 2>&1 | tail -12
python tools/make_traffic_json.py $O 3100 400000 $R/profiles/r04_pmc_hbm_traffic.json && cp $R/profiles/r04_pmc_hbm_traffic.json $O/
cd $R
timeout 1500 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
tail -c 2600 $O/bench.json; echo; grep "bench:" $O/bench.err | tail -16
cp /dev/shm/star_amd_bench/bench_extra.json $O/ 2>/dev/null
timeout 1500 python -m pytest tests -m gpu -q -n 4 > $O/pytest_gpu.log 2>&1; tail -4 $O/pytest_gpu.log
