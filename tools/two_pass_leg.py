#!/usr/bin/env python3
"""bench.py's 2-pass end-to-end leg alone, with the engine's notes (STARAMD_VERBOSE) -- where the seconds between the passes go.
  python tools/two_pass_leg.py [ENV=V ...]     (GPU box)"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench

def main():
    for a in sys.argv[1:]:
        k, v = a.split("=", 1); os.environ[k] = v
    sys.argv = [sys.argv[0]]
    args = bench.parse()
    log = lambda s: print("two_pass_leg: " + s, file=sys.stderr, flush=True)
    g, ginfo = bench.build_genome(args, args.genome_mb, log)
    idx = os.path.join(g, "idx")
    n_total = 10 * args.reads
    rd = os.path.join(g, "tp_n%d" % n_total)
    os.makedirs(rd, exist_ok=True)
    fq = bench.make_reads(args, g, rd, "reads_r0", n_total, 7000)
    os.environ["STARAMD_VERBOSE"] = "1"
    t = time.perf_counter()
    d = bench.two_pass(args, idx, fq, rd, max(4, min(64, bench.effective_cpus())))
    d["whole_call_s"] = time.perf_counter() - t
    print(json.dumps(d))

if __name__ == "__main__":
    main()
