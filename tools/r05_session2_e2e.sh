#!/bin/bash
# round 5, GPU session 2: end to end (bench.py without its CPU baseline and extra legs), alternating runs on one box:
#   one engine context per GPU (the default since round 4) against two contexts over one resident index whose kernel phases take turns -- the second context's copies (results of batch
#   k-1 down, batch k+1 up) then run beside the kernels of batch k.  Measured in round 4 while the SAM writer was the slowest stage (no gain); the writer, the slot order and the reader
#   have changed since (profiles/r04_host_stages_on_the_gpu_box.txt).
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r05s2; mkdir -p $O
run() { tag=$1; shift
env "$@" timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs --no-exclusive > $O/b_$tag.json 2> $O/b_$tag.err
python - <<PY
import json
d = json.loads(open("$O/b_$tag.json").read().strip().splitlines()[-1]); e = json.load(open(d["extra"]))
print("%-22s value %.3f ms/step %.1f  device ms %.1f  map_batch s %.3f  parse %.1f M/s  post-map %s" % ("$tag", d["value"], d["ms_per_step"], d["roofline"]["per_kernel_ms"]["device_total"], e["pipeline"]["map_batch_call_s"], e["pipeline"]["parse_Mreads_s"], {k: round(v, 2) for k, v in e["pipeline"]["postmap_whole_run_s"].items()}))
PY
}
run one_context_a X=1
run two_contexts_turns_a STARAMD_CONTEXTS_PER_GPU=2 STARAMD_KERNEL_TURNS=1
run one_context_b X=1
run two_contexts_turns_b STARAMD_CONTEXTS_PER_GPU=2 STARAMD_KERNEL_TURNS=1
run two_contexts_free STARAMD_CONTEXTS_PER_GPU=2
run one_context_fifo_slots STARAMD_SLOTS_FIFO=1
# the SAM writer is busy 53 ms per batch with 4 copy threads (round 4): with kernels below that it is the next stage to wait for
run writer_8_threads STARAMD_WRITER_THREADS=8
run writer_12_threads STARAMD_WRITER_THREADS=12
run writer_one_pwrite_stream STARAMD_WRITER_MMAP=0 STARAMD_WRITER_PWRITE_THREADS=1      # (development box: 4.3 M pairs/s and 1.33 us of a core per pair against 3.8 and 1.5 for the mapped writer)
run copied_input STARAMD_NO_INPUT_MMAP=1
