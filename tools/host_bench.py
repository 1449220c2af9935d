#!/usr/bin/env python3
"""Host stages of the pipeline (FASTQ reader / parser, post-map, SAM writer) timed on a box WITHOUT a GPU.

MEASUREMENT INFRASTRUCTURE, not a product path.  The shipped front end (star_amd/csrc/host/cli_run.cpp + libstaramd_host.so) is linked against
oracle/replay_shim.cpp: the first block of reads is mapped once by the CPU oracle (the warm-up), every later batch is the same block of
sequences again and gets a copy of the recorded result arrays after STARAMD_REPLAY_DEVICE_MS ms (the time a mapper thread would be blocked on
the MI355X).  What is timed is therefore exactly the code that runs beside the GPU in production: sah_parse_slot, sah_emit_slot, the writer.

  python tools/host_bench.py [--block 100000] [--repeat 20] [--threads 8] [--device-ms 0] [--contexts 2] [--workdir /dev/shm/hostbench]

Prints one JSON line: pairs/s end to end in the timed region, busy seconds of the parse / emit stages per batch, thread count.
"""
import argparse
import json
import os
import shutil
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--block", type=int, default=100000, help="read pairs per batch (= --gpuBatchReads); the input is this block repeated")
    ap.add_argument("--repeat", type=int, default=20, help="timed batches")
    ap.add_argument("--threads", type=int, default=os.cpu_count() or 8, help="--runThreadN")
    ap.add_argument("--device-ms", type=float, default=0.0, help="time a played batch blocks its mapper thread")
    ap.add_argument("--contexts", type=int, default=2)
    ap.add_argument("--read-len", type=int, default=101)
    ap.add_argument("--genome-mb", type=float, default=4.0)
    ap.add_argument("--workdir", default="/dev/shm/hostbench")
    ap.add_argument("--out", default="/dev/shm/hostbench/out_")
    ap.add_argument("--null-out", action="store_true", help="Aligned.out.sam is a link to /dev/null (tmpfs page allocation is slow in some containers and is not what is being measured)")
    ap.add_argument("--extra", nargs=argparse.REMAINDER, default=[], help="further alignReads flags")
    args = ap.parse_args()

    from star_amd import synth, capi
    from oracle import refstar
    lib = os.path.join(ROOT, "oracle", "_build", "libstaramd_cli_replay.so")
    if not os.path.isfile(lib):
        sys.exit("build it first: make oracle/_build/libstaramd_cli_replay.so")
    d = os.path.join(args.workdir, "b%d_l%d_g%g" % (args.block, args.read_len, args.genome_mb))
    tag = os.path.join(d, "ready")
    if not os.path.isfile(tag):
        if not refstar.have_ref():
            sys.exit("oracle/_ref/STAR is needed once, to build the small index (make ref)")
        shutil.rmtree(d, ignore_errors=True)
        n_chr = 4
        info = synth.make_dataset(d, seed=11, chr_lengths=(int(args.genome_mb * 1e6 / n_chr),) * n_chr, n_tr=int(60 * args.genome_mb), n_reads=args.block,
                                  read_len=args.read_len, paired=True)
        refstar.genome_generate(info["fasta"], os.path.join(d, "idx"), gtf=info["gtf"], sjdb_overhang=args.read_len - 1, sa_index_nbases=10)
        json.dump(info, open(os.path.join(d, "info.json"), "w"))
        open(tag, "w").write("ok")
    info = json.load(open(os.path.join(d, "info.json")))
    # the block, repeat+1 times (warm-up = the recorded one)
    fq = []
    for m, src in enumerate(info["fastq"]):
        dst = os.path.join(d, "rep%d_%d.fq" % (args.repeat + 1, m + 1))
        if not os.path.isfile(dst):
            blob = open(src, "rb").read()
            with open(dst + ".tmp", "wb") as f:
                for _ in range(args.repeat + 1):
                    f.write(blob)
            os.rename(dst + ".tmp", dst)
        fq.append(dst)
    os.environ["STARAMD_REPLAY_DEVICE_MS"] = str(args.device_ms)
    os.environ["STARAMD_CONTEXTS_PER_GPU"] = str(args.contexts)
    os.environ.setdefault("STARAMD_SJDB_HOST", "1")
    argv = ["--runMode", "alignReads", "--genomeDir", os.path.join(d, "idx"), "--readFilesIn"] + fq + ["--outFileNamePrefix", args.out, "--runThreadN", str(args.threads),
            "--gpuBatchReads", str(args.block), "--benchWarmupReads", str(args.block)] + list(args.extra)
    sam = args.out + "Aligned.out.sam"
    if os.path.lexists(sam):
        os.remove(sam)
    if args.null_out:
        os.symlink("/dev/null", sam)
    t0 = time.time()
    cpu = {}
    rc, rep = capi.run_cli(argv, lib_path=lib, warmup_done=lambda: cpu.__setitem__("t0", time.process_time()))
    cpu["t1"] = time.process_time()
    if rc:
        sys.exit("front end failed: rc %d" % rc)
    nb = max(1, rep.batches)
    line = {
        "what": "host stages over played-back result arrays (no GPU): tools/host_bench.py",
        "pairs_per_s": rep.timedReads / rep.timedWall if rep.timedWall > 0 else None,
        "timed_pairs": rep.timedReads, "timed_wall_s": round(rep.timedWall, 4), "batches": rep.batches, "batch_pairs": args.block,
        "parse_ms_per_batch": round(1e3 * rep.parseBusy / nb, 2), "emit_ms_per_batch": round(1e3 * rep.emitBusy / nb, 2), "finish_s": round(rep.finishSeconds, 3),
        "parse_pairs_per_s": args.block * nb / rep.parseBusy if rep.parseBusy > 0 else None, "emit_pairs_per_s": args.block * nb / rep.emitBusy if rep.emitBusy > 0 else None,
        "cpu_us_per_pair_by_stage": dict(zip(["input_line_table", "text_to_numeric", "mapper_threads", "postmap_format", "file_writes", "other"], [round(float(x) * 1e6 / max(1, rep.timedReads), 4) for x in list(rep.cpuSeconds)[:6]])),
        "fast_path_batches": [int(x) for x in list(rep.fastPaths)[:3]],
        "cpu_us_per_pair": round(1e6 * (cpu["t1"] - cpu["t0"]) / rep.timedReads, 3) if "t0" in cpu and rep.timedReads else None,
        "threads": args.threads, "device_ms": args.device_ms, "contexts": rep.nContexts, "read_len": args.read_len, "total_wall_s": round(time.time() - t0, 2),
        "sam_bytes": os.path.getsize(sam) if os.path.isfile(sam) and not args.null_out else None,
    }
    print(json.dumps(line))


if __name__ == "__main__":
    main()
