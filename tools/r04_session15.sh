#!/bin/bash
# round 4, GPU session 15: what the post-map stage waits for on this box (writer into tmpfs vs formatting threads)
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/s15; mkdir -p $O
for rep in a b; do
timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs --no-exclusive > $O/b_$rep.json 2> $O/b_$rep.err
python - <<PY
import json
d = json.loads(open("$O/b_$rep.json").read().strip().splitlines()[-1]); e = json.load(open(d["extra"]))
print("run $rep: value %.3f ms/step %.1f emit %.1f M/s parse %.1f M/s; post-map whole run (25 batches): %s" % (d["value"], d["ms_per_step"], e["pipeline"]["postmap_write_Mreads_s"], e["pipeline"]["parse_Mreads_s"], e["pipeline"]["postmap_whole_run_s"]))
PY
done
# tmpfs write speed of this box: 2 GB in 1 / 2 / 4 streams
python - <<'PY'
import os, time, threading
buf = b"x" * (64 << 20)
for nt in (1, 2, 4):
    def w(i):
        with open("/dev/shm/_wtest_%d" % i, "wb", buffering=0) as f:
            for _ in range(2048 // 64 // nt): f.write(buf)
    t = time.perf_counter(); th = [threading.Thread(target=w, args=(i,)) for i in range(nt)]; [x.start() for x in th]; [x.join() for x in th]; dt = time.perf_counter() - t
    print("tmpfs write, %d stream(s): %.2f GB/s" % (nt, 2.0 / dt)); [os.remove("/dev/shm/_wtest_%d" % i) for i in range(nt)]
PY
