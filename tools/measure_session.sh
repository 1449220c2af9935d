#!/bin/bash
# final measurement session of the round (run from the repo root on the GPU box)
R=$PWD
mkdir -p gpurun_out/final
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -15 > gpurun_out/final/pytest_gpu.log
tail -2 gpurun_out/final/pytest_gpu.log
timeout 900 python bench.py > gpurun_out/final/bench.json 2> gpurun_out/final/bench.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/final/bench.json"))
print("value", d["value"], "ms", d["ms_per_step"], d["roofline"]["per_kernel_ms"])
for k in ("cpu_baseline", "cli_end_to_end", "full_size_parity", "two_pass_end_to_end"):
    v = d.get(k); print(k, {kk: vv for kk, vv in v.items() if kk not in ("sample", "what")} if isinstance(v, dict) else v)
PY
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/final/prof -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-cli-e2e --no-two-pass-e2e > $R/gpurun_out/final/bench_prof.json 2> $R/gpurun_out/final/prof.err
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/final/pmc_fetch -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-cli-e2e --no-two-pass-e2e > /dev/null 2> $R/gpurun_out/final/pmc_fetch.err
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/final/pmc_write -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-cli-e2e --no-two-pass-e2e > /dev/null 2> $R/gpurun_out/final/pmc_write.err
cd $R
# keep the merge small: drop everything but the csv summaries
find gpurun_out/final -type f ! -name "*.csv" ! -name "*.json" ! -name "*.log" ! -name "*.err" -delete
find gpurun_out/final -name "*.csv" -size +20M -delete
du -sh gpurun_out/final; find gpurun_out/final -name "*.csv" | head -20
