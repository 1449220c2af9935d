#!/bin/bash
# round 4, GPU session 16: the SAM writer through a mapping (4 threads) vs positional writes (2 threads)
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/s16; mkdir -p $O
run() { tag=$1; shift
env "$@" timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs --no-exclusive > $O/b_$tag.json 2> $O/b_$tag.err
python - <<PY
import json
d = json.loads(open("$O/b_$tag.json").read().strip().splitlines()[-1]); e = json.load(open(d["extra"]))
print("%-12s value %.3f ms/step %.1f emit %.1f M/s parse %.1f M/s; post-map: %s" % ("$tag", d["value"], d["ms_per_step"], e["pipeline"]["postmap_write_Mreads_s"], e["pipeline"]["parse_Mreads_s"], {k: round(v, 2) for k, v in e["pipeline"]["postmap_whole_run_s"].items()}))
PY
}
run mmap4_a X=1
run pwrite_a STARAMD_WRITER_MMAP=0
run mmap4_b X=1
run pwrite_b STARAMD_WRITER_MMAP=0
run mmap8 STARAMD_WRITER_THREADS=8
run mmap2 STARAMD_WRITER_THREADS=2
