#!/bin/bash
# round 4, session 20 (the last GPU-box seconds of the round; no GPU work): the host stages ALONE on the box's 16 CPUs -- the shipped front end over played-back
# result arrays (tools/host_bench.py), with the mapper blocked 0 ms per batch (= what the host could feed and drain) and 52 ms (= the device time of a batch)
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/s20; mkdir -p $O
run() { timeout 40 python tools/host_bench.py --contexts 1 --block 400000 --repeat 12 "$@" 2>/dev/null | tail -1 >> $O/host_bench.jsonl; }
run --threads 16 --device-ms 0
run --threads 16 --device-ms 52
run --threads 2 --device-ms 0
run --threads 2 --device-ms 52
run --threads 4 --device-ms 52
python - <<'PY'
import json
for l in open("gpurun_out/s20/host_bench.jsonl"):
    try: d = json.loads(l)
    except Exception: continue
    print("threads %2d device_ms %4.0f: %.2f M pairs/s  parse %.1f ms/batch emit %.1f ms/batch  cpu %.2f us/pair" % (d["threads"], d["device_ms"], d["pairs_per_s"] / 1e6, d["parse_ms_per_batch"], d["emit_ms_per_batch"], d["cpu_us_per_pair"]))
PY
