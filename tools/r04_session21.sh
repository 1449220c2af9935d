#!/bin/bash
# round 4, session 21 (no GPU work): batch slots handed out last-in-first-out against round-robin, host stages alone on the box's 16 CPUs (tools/host_bench.py)
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/s21; mkdir -p $O
run() { tag=$1; shift; env "$@" timeout 40 python tools/host_bench.py --contexts 1 --block 400000 --repeat 20 --threads 16 --device-ms 52 2>/dev/null | tail -1 | sed "s/^{/{\"slots\": \"$tag\", /" >> $O/host_bench.jsonl; }
run lifo X=1
run fifo STARAMD_SLOTS_FIFO=1
run lifo X=1
run fifo STARAMD_SLOTS_FIFO=1
python - <<'PY'
import json
for l in open("gpurun_out/s21/host_bench.jsonl"):
    try: d = json.loads(l)
    except Exception: continue
    print("%s threads %2d device_ms %4.0f: %.2f M pairs/s  parse %.1f ms/batch emit %.1f ms/batch  cpu %.2f us/pair" % (d["slots"], d["threads"], d["device_ms"], d["pairs_per_s"] / 1e6, d["parse_ms_per_batch"], d["emit_ms_per_batch"], d["cpu_us_per_pair"]))
PY
