#!/usr/bin/env python3
"""bench.py's config-5 leg with chimeric detection, alone, with the pipeline event log (where a batch waits and for what) and the engine's messages.
  python tools/chim_leg.py [--batches 4] [--out gpurun_out/<tag>]   (GPU box; the 3.1 Gb index comes from the bench cache or is generated)"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench

def main():
    out = "gpurun_out/chim"
    nb = 4
    av = sys.argv[1:]
    while av:
        a = av.pop(0)
        if a == "--out": out = av.pop(0)
        elif a == "--batches": nb = int(av.pop(0))
    os.makedirs(out, exist_ok=True)
    sys.argv = [sys.argv[0]]
    args = bench.parse()
    notes = []
    log = lambda s: (notes.append(s), print("chim_leg: " + s, file=sys.stderr, flush=True))
    g, ginfo = bench.build_genome(args, args.genome_mb, log)
    idx = os.path.join(g, "idx")
    L = 150; w = 1
    n_total = (nb + w) * args.reads
    rd = os.path.join(g, "chim_n%d" % n_total)
    fq = bench.make_reads(args, g, rd, "chim", n_total, 8100, read_len=L, chim_rate=0.05)
    for name, flags in (("chim", ["--chimSegmentMin", "12", "--chimOutType", "Junctions"]),):
        argv = ["--runMode", "alignReads", "--genomeDir", idx, "--readFilesIn"] + fq + ["--outFileNamePrefix", os.path.join(rd, "cl_" + name + "_"), "--runThreadN", str(max(4, min(64, bench.effective_cpus()))),
                "--gpuBatchReads", str(args.reads), "--benchWarmupReads", str(w * args.reads), "--readMapNumber", str(n_total)] + flags
        t = time.perf_counter()
        rep, d = bench._cli_leg(argv, 2 * L + 1, {"STARAMD_PIPELINE_LOG": os.path.abspath(os.path.join(out, "plog_%s.txt" % name)), "STARAMD_VERBOSE": "1", "STARAMD_HOST_TIMING": "1"})
        d["wall_s_whole_run"] = time.perf_counter() - t
        d["pipeline"] = {"timed_wall_s": float(rep.timedWall), "map_batch_call_s": sum(float(rep.deviceBusy[k]) for k in range(max(1, int(rep.nContexts)))), "parse_busy_s": float(rep.parseBusy),
                         "convert_busy_s": float(rep.convertBusy), "postmap_write_busy_s": float(rep.emitBusy), "finish_s": float(rep.finishSeconds),
                         "cpu_us_per_pair_by_stage": dict(zip(["input_line_table", "text_to_numeric", "mapper_threads", "postmap_format", "file_writes", "other"], [round(float(x) * 1e6 / max(1, int(rep.timedReads)), 4) for x in list(rep.cpuSeconds)[:6]]))}
        json.dump(d, open(os.path.join(out, "leg_%s.json" % name), "w"), indent=1)
        print(name, "Mreads/s %.3f" % d["Mreads_s"], "device", d["per_kernel_ms"].get("device_total"), json.dumps(d["pipeline"]))

if __name__ == "__main__":
    main()
