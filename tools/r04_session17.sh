#!/bin/bash
# round 4, last GPU session: CLI / BAM tests on the final host code, then the default bench for the record
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/s17; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_twins.py tests/test_bam.py tests/test_two_pass.py tests/test_config1.py -m gpu -q -x -n 4 > $O/pytest_cli.log 2>&1; tail -3 $O/pytest_cli.log
timeout 1500 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc $?"; python - <<'PY'
import json
d = json.loads(open("gpurun_out/s17/bench.json").read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d["full_size_parity"], d["cpu_baseline"]["value"], d["cpu_baseline"]["best_value"], len(json.dumps(d)))
PY
cp /dev/shm/star_amd_bench/bench_extra.json $O/ 2>/dev/null
