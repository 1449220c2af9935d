#!/usr/bin/env python3
"""bench.py -- throughput of the seed-search-and-stitch hot path on MI355X, measured END TO END:
FASTQ text in -> Aligned.out.sam + SJ.out.tab out, index load excluded, on a human-scale index.

Contract (see the task statement): `python bench.py --gpus N --steps K --warmup W` prints ONE JSON line on rank 0.
  step      = one batch of `--reads` DISTINCT read pairs streamed through the product pipeline (star_amd's front end run in-process:
              FASTQ parse -> staramd_map_batch on the GPU -> multMapSelect ... SAM text -> file), batches overlapping as they do in the CLI
  timing    = W warm-up batches are mapped AND written, the pipeline is drained, all ranks meet at a barrier, then the clock runs from
              the submission of the first of the K timed batches to the last SAM byte of the last one (barrier again, max over ranks)
  metric    = BASELINE.json's: million reads (pairs) aligned per second, whole job
  workload  = BASELINE config 2 stand-in: synthetic genome of `--genome-mb` megabases (default 3100 = human size, ~half of it repeat
              families) + ~350 k annotated junctions (sjdbOverhang 100, 14-base SAindex), 2x101 bp pairs, 85 % spliced, 1 % substitutions;
              the index is generated INSIDE the run on the GPU (star_amd --runMode genomeGenerate: the reference needs ~20 min for it)
  roofline  = dominant kernel by HIP-event time on the engine's stream, summed over the timed batches: algorithmic bytes / duration
              vs the 8 TB/s HBM peak
  cpu_baseline = the reference itself (oracle/_ref/STAR) on the same FASTQ and the same index, --runThreadN = all host cores, with an
              input chunk small enough that every thread has work (>= 8 chunks per thread)
Multi-GPU: one process per GPU (torch.distributed / RCCL for the barriers, the max-over-ranks reduction and the junction-table
all_gather at the end of the run); reads are sharded, every rank holds a full index replica; no data-path collective (weak scaling).
"""
import argparse
import ctypes as C
import hashlib
import json
import math
import os
import pickle
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
T_START = time.time()


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--genome-mb", type=int, default=int(os.environ.get("STARAMD_BENCH_GENOME_MB", "3100")))
    ap.add_argument("--reads", type=int, default=int(os.environ.get("STARAMD_BENCH_READS", "400000")), help="read pairs per GPU per step (= --gpuBatchReads)")
    ap.add_argument("--read-len", type=int, default=101)
    ap.add_argument("--host-threads", type=int, default=int(os.environ.get("STARAMD_BENCH_HOST_THREADS", "0")), help="post-map threads per rank (0: min(64, cores/ranks))")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sweep", action="store_true", help="skip the index-size sweep (100 / 400 / 1000 Mb)")
    ap.add_argument("--no-two-pass", action="store_true")
    ap.add_argument("--no-extra-legs", action="store_true", help="skip kernel_ms_exclusive / variants / host_budget / all_transcripts / config1")
    ap.add_argument("--budget-s", type=float, default=float(os.environ.get("STARAMD_BENCH_BUDGET_S", "1500")), help="optional legs are skipped once this much wall time is used")
    ap.add_argument("--workdir", default=os.environ.get("STARAMD_BENCH_DIR", "/dev/shm/star_amd_bench" if os.path.isdir("/dev/shm") else "/tmp/star_amd_bench"))
    return ap.parse_args()


# ---------------------------------------------------------------------------------------------------------------------
# data

def genome_dir(args, mb):
    key = hashlib.md5(("v5g|%d|%d" % (mb, args.read_len)).encode()).hexdigest()[:10]
    return os.path.join(args.workdir, "genome_%dmb_%s" % (mb, key))


def build_genome(args, mb, log):
    """Synthetic genome + annotation + index (rank 0).  The index is built by the product itself on the GPU
    (star_amd --runMode genomeGenerate, byte-identical to the reference's genomeGenerate: tests/test_index_build.py)."""
    import numpy as np
    from star_amd import synth
    g = genome_dir(args, mb)
    if os.path.isfile(os.path.join(g, "DONE")):
        return g, json.load(open(os.path.join(g, "build.json")))
    os.makedirs(g, exist_ok=True)
    info = {"genome_mb": mb}
    nchr = max(1, min(24, mb // 40))
    t = time.time()
    seqs, frac = synth.make_genome_large(20260922, mb, nchr)
    info["genome_synth_s"] = time.time() - t; info["repeat_bases_fraction"] = frac
    rng = np.random.default_rng(20260923)
    t = time.time()
    trs = synth.make_transcripts(rng, seqs, 65 * mb)
    annotated = rng.random(len(trs)) < 0.7
    info["transcripts"] = len(trs); info["transcripts_s"] = time.time() - t
    names = ["chr%d" % (i + 1) for i in range(nchr)]
    t = time.time()
    synth._write_fasta(os.path.join(g, "genome.fa"), names, seqs)
    synth.write_gtf(os.path.join(g, "annot.gtf"), names, trs, annotated)
    np.save(os.path.join(g, "genome.npy"), seqs[0].base if seqs[0].base is not None else np.concatenate(seqs))
    pickle.dump({"trs": trs, "chr_len": [len(s) for s in seqs]}, open(os.path.join(g, "transcripts.pkl"), "wb"))
    info["write_fasta_gtf_s"] = time.time() - t
    del seqs
    nb = max(4, min(14, int(math.log2(mb * 1e6) / 2 - 1)))
    exe = os.path.join(ROOT, "star_amd", "bin", "star_amd")
    idx = os.path.join(g, "idx")
    os.makedirs(idx, exist_ok=True)
    cmd = [exe, "--runMode", "genomeGenerate", "--genomeDir", idx, "--genomeFastaFiles", os.path.join(g, "genome.fa"), "--genomeSAindexNbases", str(nb),
           "--sjdbGTFfile", os.path.join(g, "annot.gtf"), "--sjdbOverhang", str(args.read_len - 1), "--runThreadN", str(os.cpu_count() or 8),
           "--outFileNamePrefix", idx + "/_log_"]
    t = time.time()
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=dict(os.environ, STARAMD_HOST_TIMING="1"))
    info["index_generate_s"] = time.time() - t
    if p.returncode != 0:
        raise RuntimeError("star_amd --runMode genomeGenerate failed: " + p.stderr[-2000:])
    info["index_generate_log"] = p.stderr.strip().splitlines()[-8:]
    info["SAindexNbases"] = nb
    info["junctions_in_index"] = int(open(os.path.join(idx, "sjdbInfo.txt")).readline().split()[0])
    info["index_bytes"] = sum(os.path.getsize(os.path.join(idx, f)) for f in ("Genome", "SA", "SAindex"))
    json.dump(info, open(os.path.join(g, "build.json"), "w"))
    open(os.path.join(g, "DONE"), "w").write("ok\n")
    log("genome %d Mb: synth %.1f s, transcripts %.1f s, files %.1f s, index %.1f s" % (mb, info["genome_synth_s"], info["transcripts_s"], info["write_fasta_gtf_s"], info["index_generate_s"]))
    return g, info


_SAMPLER = None


_CHIM = 0.0


def _chunk_job(job):
    seed, n, prefix, first_id = job
    from star_amd import synth
    m1, m2 = _SAMPLER.sample(seed, n, chim_rate=_CHIM)
    synth.write_fastq_ids(prefix, m1, m2, first_id)
    return prefix


def make_reads(args, g, out_dir, tag, n_pairs, seed_base, read_len=None, chim_rate=0.0):
    """n_pairs pairs as <out_dir>/<tag>_{1,2}.fq.  Runs in a child interpreter (`bench.py --make-reads ...`): the sampler forks a pool of
    workers, which must not happen in a process that has initialised the HIP runtime."""
    out = [os.path.join(out_dir, "%s_%d.fq" % (tag, m)) for m in (1, 2)]
    if not os.path.isfile(os.path.join(out_dir, tag + ".DONE")):
        subprocess.run([sys.executable, os.path.abspath(__file__), "--make-reads", json.dumps([read_len or args.read_len, g, out_dir, tag, n_pairs, seed_base, int(os.environ.get("WORLD_SIZE", "1")), chim_rate])],
                       check=True, timeout=1200, env=dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES=""))
    return out


def _make_reads_child(read_len, g, out_dir, tag, n_pairs, seed_base, world, chim_rate=0.0):
    """chunks are sampled by a pool of forked workers from the shared genome / transcriptome arrays, then joined into one file per mate
    (a comma-separated list would send the reference through a fifo + an executable script in its temp directory: /dev/shm is noexec)"""
    global _SAMPLER, _CHIM
    import numpy as np
    from star_amd import synth
    _CHIM = float(chim_rate)
    os.makedirs(out_dir, exist_ok=True)
    chunk = 500000
    jobs = []
    for c, lo in enumerate(range(0, n_pairs, chunk)):
        jobs.append((seed_base * 100003 + c, min(chunk, n_pairs - lo), os.path.join(out_dir, "%s_c%03d" % (tag, c)), lo))
    done = os.path.join(out_dir, tag + ".DONE")
    if not os.path.isfile(done):
        gseq = np.load(os.path.join(g, "genome.npy"), mmap_mode="r")
        meta = pickle.load(open(os.path.join(g, "transcripts.pkl"), "rb"))
        seqs, o = [], 0
        for ln in meta["chr_len"]:
            seqs.append(gseq[o:o + ln]); o += ln
        _SAMPLER = synth.ReadSampler(seqs, meta["trs"], read_len)
        _SAMPLER.gseq = np.asarray(gseq)
        import multiprocessing as mp
        nproc = max(1, min(len(jobs), (os.cpu_count() or 8) // max(1, world), 32))
        if nproc > 1:
            with mp.get_context("fork").Pool(nproc) as pool:
                pool.map(_chunk_job, jobs)
        else:
            for j in jobs:
                _chunk_job(j)
        _SAMPLER = None
        for m in (1, 2):
            with open(os.path.join(out_dir, "%s_%d.fq" % (tag, m)), "wb") as fo:
                for j in jobs:
                    part = j[2] + "_%d.fq" % m
                    with open(part, "rb") as fi:
                        while True:
                            b = fi.read(1 << 26)
                            if not b:
                                break
                            fo.write(b)
                    os.remove(part)
        open(done, "w").write("ok\n")


# ---------------------------------------------------------------------------------------------------------------------
# the product pipeline in-process (libstaramd_cli.so, include/star_amd_cli.h)

from star_amd.capi import CliHooks, CliReport, run_cli      # the product pipeline in-process (libstaramd_cli.so, include/star_amd_cli.h)


COUNTER_NAMES = ["nSAi", "nSAprobe", "nGcmp", "nSAenum", "nGstitch", "nSeeds", "nWindows", "nWA", "nNodes", "nLeaves", "nStitchCalls", "nExtendCalls", "nTrOut",
                 "nOvfWin", "nOvfStitch", "nRedoWin", "nReplayWin"]
STAGE_NAMES = ["k_seed_search", "k_windows", "k_order", "k_stitch_win", "k_stitch_verify+replay+finish", "k_scan+k_gather", "device_total"]


def engine_src_sha():
    """sha256 over the kernel sources of the engine (every .hip / .h file of star_amd/csrc/engine)"""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "star_amd", "csrc", "engine")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h")):
            h.update(f.encode()); h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


# what a PMC-derived figure of a kernel (profiles/*_pmc_hbm_traffic.json) must have been measured on: the file(s) the kernel is written in and the
# headers they share.  engine.hip (launch code of all kernels) is not part of it: a change there that moves a kernel's launch parameters needs a new
# PMC pass all the same (tools/measure_session.sh), the hash cannot know.
KERNEL_SOURCES = {
    "k_seed_search": ["k_seed.hip"], "k_windows": ["k_window.hip"], "k_gather": ["k_gather.hip"],
    "k_stitch_win": ["k_stitch.hip", "k_stitch_lane.hip", "stitch_common.h", "stitch_scalar.h"],
    "k_stitch_replay": ["k_stitch.hip", "stitch_common.h"], "k_stitch_finish": ["k_stitch.hip", "stitch_common.h"],
}


def kernel_src_sha(kernel):
    h = hashlib.sha256()
    d = os.path.join(ROOT, "star_amd", "csrc", "engine")
    for f in KERNEL_SOURCES[kernel] + ["dev.h", os.path.join("..", "..", "..", "include", "star_amd.h")]:
        h.update(os.path.basename(f).encode()); h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def report_dict(rep, lread):
    n = max(int(rep.timedReads), 1); nb = max(int(rep.batches), 1)
    c = dict(zip(COUNTER_NAMES, [int(x) for x in rep.counters]))
    c["nPrunedWin"], c["nRewalkRead"], c["nLaneItems"] = int(rep.counters[37]), int(rep.counters[38]), int(rep.counters[39])     # dev.h: DC_nPrunedWin, DC_nRewalkRead, DC_nLaneItems
    for k in ("nNodes", "nLeaves", "nStitchCalls", "nExtendCalls"):      # kept by the profile / shadow builds only
        c.pop(k, None)
    ms = dict(zip(STAGE_NAMES, [float(x) / nb for x in rep.stageMs]))       # per launch = per batch
    # Algorithmic bytes: what the algorithm must fetch / write, from the engine's own counters
    bytes_seed = 8 * c["nSAi"] + 8 * c["nSAprobe"] + c["nGcmp"] + n * lread + 24 * c["nSeeds"]
    bytes_win = 8 * c["nSAenum"] + 24 * c["nSeeds"] + 24 * c["nWA"]
    bytes_stitch = c["nGstitch"] + 24 * c["nWA"] + (lread // 2) * (c["nWindows"] if c["nWindows"] else n) + 96 * c["nTrOut"] + 32 * 2 * c["nTrOut"]
    kern = {"k_seed_search": (ms["k_seed_search"], bytes_seed / nb), "k_windows": (ms["k_windows"], bytes_win / nb), "k_stitch_win": (ms["k_stitch_win"], bytes_stitch / nb)}
    if os.environ.get("STARAMD_PROFILE_BUILD"):          # libstaramd.so built with -DSTARAMD_PROFILE: shader-clock cycles per section, summed over waves
        pn = ["walk(all)", "coopStitch", "coopExtend", "finalize(all)", "recordCandidate", "-", "-", "wave_lifetime", "windows:passA", "windows:flanks", "windows:passB_enumerate+owner",
              "windows:passB_assign", "windows:emission", "finalize:extends", "finalize:filters+score", "finalize:candidate+log"]
        raw = [int(x) for x in rep.counters]
        c["profile_cycles_per_pair"] = dict(zip(pn, [v / n for v in raw[21:37]]))
        c["profile_counts_per_pair"] = dict(zip(["nNodes", "nLeaves", "nStitchCalls", "nExtendCalls"], [v / n for v in raw[8:12]]))
    return c, ms, kern, (bytes_seed + bytes_win + bytes_stitch) / n


# ---------------------------------------------------------------------------------------------------------------------
# CPU baseline + parity

def cpu_baseline(idx, fq, out_prefix, n_pairs, log=lambda s: None):
    """The reference itself (oracle/_ref/STAR, built from /root/reference by oracle/Makefile.ref), same index, same FASTQ, default parameters.
    STAR deals input to its threads in chunks of limitIObufferSize[0]/nMates bytes (ReadAlignChunk_processChunks.cpp:14-30, Parameters.cpp:1160);
    the default (30 MB -> ~67 k pairs) would leave most of 256 threads without a chunk on a 10 M-pair sample, so the input buffer is set to
    2 MB (~4.4 k pairs per chunk: >= 8 chunks per thread).  Mapping time = wall(full run) - wall(run that only loads the index)."""
    from oracle import refstar
    ncpu = os.cpu_count() or 1
    small = ["--limitIObufferSize", "2000000", "50000000"]
    chunks = n_pairs * 225 // 1000000

    def run(nmap, threads, extra=()):
        t = time.perf_counter()
        refstar.align(idx, fq, out_prefix, threads=threads, extra=["--readMapNumber", str(nmap)] + small + list(extra), timeout=900)
        log("reference STAR: %d reads, %d threads: %.1f s" % (nmap, threads, time.perf_counter() - t))
        return time.perf_counter() - t
    run(1, ncpu)                                    # page cache
    t_load = min(run(1, ncpu), run(1, ncpu))
    tried = {}
    extras = {}
    best_th = max(1, ncpu // 4)
    try:        # what a user would run on this box instead of one 256-thread process
        t = time.perf_counter()
        refstar.align(idx, fq, out_prefix + "dflt_", threads=best_th, extra=["--readMapNumber", str(n_pairs)], timeout=900)       # default --limitIObufferSize
        extras["default_io_buffer"] = {"threads": best_th, "Mreads_s": n_pairs / max(time.perf_counter() - t - t_load, 1e-3) / 1e6}
        log("reference STAR, default input buffer, %d threads: %.1f s" % (best_th, time.perf_counter() - t))
    except Exception as e:
        extras["default_io_buffer"] = {"error": repr(e)[:200]}
    try:
        extras["multi_process"] = multi_process_baseline(idx, fq, out_prefix, n_pairs, ncpu, small, log)
    except Exception as e:
        extras["multi_process"] = {"error": repr(e)[:300]}
    for th in sorted(set([best_th, ncpu])):
        t_full = run(n_pairs, th)
        tried[th] = n_pairs / max(t_full - t_load, 1e-3) / 1e6
    # the all-core run is last: its outputs stay for the parity check
    n1 = min(n_pairs, 60000)
    one = None
    try:
        p1 = out_prefix + "t1_"
        t = time.perf_counter(); refstar.align(idx, fq, p1, threads=1, extra=["--readMapNumber", str(n1)], timeout=600); t_one = time.perf_counter() - t
        t = time.perf_counter(); refstar.align(idx, fq, p1, threads=1, extra=["--readMapNumber", "1"], timeout=600); t_one_load = time.perf_counter() - t
        one = n1 / max(t_one - t_one_load, 1e-3) / 1e6
    except Exception:
        one = None
    return {"value": tried[ncpu], "unit": "Mreads/s", "cores": ncpu, "kind": "reference",
            "sample": "%d pairs (the same FASTQ the GPU run maps), STAR 2.7.11b --runThreadN %d, --limitIObufferSize 2000000 (~%d input chunks = %.1f per thread); "
                      "mapping time = wall(full) - wall(index load only, %.1f s); by thread count (Mreads/s): %s; one thread on %d pairs: %s Mreads/s"
                      % (n_pairs, ncpu, chunks, chunks / ncpu, t_load, ", ".join("%d: %.4f" % (k, v) for k, v in sorted(tried.items())), n1,
                         ("%.5f" % one) if one else "n/a"),
            "by_threads": {str(k): v for k, v in sorted(tried.items())}, "one_thread": one, "index_load_s": t_load,
            "default_io_buffer": extras.get("default_io_buffer"), "multi_process": extras.get("multi_process")}


def memory_headroom_gb():
    """what this container may still allocate: MemAvailable, capped by the cgroup limit minus its current usage (tmpfs pages included)"""
    avail = None
    for l in open("/proc/meminfo"):
        if l.startswith("MemAvailable:"):
            avail = int(l.split()[1]) * 1024 / 1e9
    for lim, cur in (("/sys/fs/cgroup/memory.max", "/sys/fs/cgroup/memory.current"), ("/sys/fs/cgroup/memory/memory.limit_in_bytes", "/sys/fs/cgroup/memory/memory.usage_in_bytes")):
        try:
            a, b = open(lim).read().strip(), open(cur).read().strip()
            if a != "max" and int(a) < (1 << 60):
                head = (int(a) - int(b)) / 1e9
                avail = head if avail is None else min(avail, head)
        except Exception:
            pass
    return avail if avail is not None else 0.0


def multi_process_baseline(idx, fq, out_prefix, n_pairs, ncpu, small, log):
    """8 reference processes side by side, cores/8 threads each, every one on its own eighth of the FASTQ (each holds its own copy of the index, as
    8 independent STAR runs do): what the box can do when the one input mutex of a single process is taken out of the picture."""
    from oracle import refstar
    nproc = 8
    idx_gb = sum(os.path.getsize(os.path.join(idx, f)) for f in ("Genome", "SA", "SAindex")) / 1e9
    free_gb = memory_headroom_gb()
    # every process holds its own copy of the index (+ ~0.1 GB per thread); the copies must fit into what the CONTAINER may still use (its cgroup limit
    # counts the tmpfs the workload lives in), with a wide margin: a box that runs out of memory is lost, not slowed down
    while nproc > 1 and nproc * (idx_gb * 1.1 + 2 + 0.1 * (ncpu // nproc)) > 0.7 * free_gb:
        nproc //= 2
    if nproc < 2:
        return {"skipped": "%.0f GB of memory headroom for index copies of %.0f GB" % (free_gb, idx_gb)}
    per = n_pairs // nproc
    # the FASTQ records have a fixed size (synth.write_fastq_ids): slices by byte offset
    slices = []
    for k in range(nproc):
        fs = []
        for m, f in enumerate(fq):
            rec = os.path.getsize(f) // n_pairs
            o = "%smp%d_%d.fq" % (out_prefix, k, m + 1)
            with open(f, "rb") as fi, open(o, "wb") as fo:
                fi.seek(k * per * rec); left = per * rec
                while left > 0:
                    b = fi.read(min(left, 1 << 26)); fo.write(b); left -= len(b)
            fs.append(o)
        slices.append(fs)
    th = max(1, ncpu // nproc)

    def wave(nmap):
        t = time.perf_counter()
        ps = [subprocess.Popen([refstar.REF_BIN, "--runMode", "alignReads", "--genomeDir", idx, "--readFilesIn"] + slices[k] + ["--runThreadN", str(th), "--outFileNamePrefix", "%smp%d_" % (out_prefix, k),
                                "--readMapNumber", str(nmap)] + small, stdout=subprocess.DEVNULL) for k in range(nproc)]
        rcs = [p.wait(timeout=1200) for p in ps]
        if any(rcs):
            raise RuntimeError("exit codes %r" % rcs)
        return time.perf_counter() - t
    t_load = wave(1)
    t_full = wave(per)
    for fs in slices:
        for f in fs:
            os.remove(f)
    log("reference STAR x %d processes x %d threads: load %.1f s, full %.1f s" % (nproc, th, t_load, t_full))
    return {"processes": nproc, "threads_each": th, "Mreads_s": nproc * per / max(t_full - t_load, 1e-3) / 1e6, "load_only_s": t_load, "full_s": t_full, "pairs": nproc * per,
            "memory_headroom_gb": free_gb, "what": "independent reference processes side by side, each on its own slice of the FASTQ with its own index copy; as many (8, 4 or 2) as fit "
                                                   "into the container's memory limit with a wide margin"}


def _digest_range(job):
    path, lo, hi = job
    n, acc = 0, 0
    with open(path, "rb") as f:
        if lo:
            f.seek(lo - 1)
            if f.read(1) != b"\n":
                f.readline()
        while f.tell() < hi:
            l = f.readline()
            if not l:
                break
            if l[:1] == b"@":
                continue
            acc = (acc + int.from_bytes(hashlib.blake2b(l, digest_size=8).digest(), "little")) & 0xFFFFFFFFFFFFFFFF
            n += 1
    return n, acc


def sam_digest(path):
    """(record count, order-independent sum of 64-bit record hashes) of a SAM file; in a child interpreter (it forks a pool)."""
    p = subprocess.run([sys.executable, os.path.abspath(__file__), "--sam-digest", path], check=True, timeout=1200, stdout=subprocess.PIPE, text=True,
                       env=dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES=""))
    n, acc = p.stdout.split()
    return int(n), int(acc)


def _sam_digest_child(path):
    import multiprocessing as mp
    size = os.path.getsize(path)
    nproc = max(1, min(64, (os.cpu_count() or 8) // 2, size // (1 << 24) + 1))
    step = size // nproc + 1
    jobs = [(path, i * step, min(size, (i + 1) * step)) for i in range(nproc)]
    with mp.get_context("fork").Pool(nproc) as pool:
        parts = pool.map(_digest_range, jobs)
    return sum(p[0] for p in parts), sum(p[1] for p in parts) & 0xFFFFFFFFFFFFFFFF


def full_size_parity(ref, new):
    """Parity at the size of the bench run (size-independent properties): SJ.out.tab and the Log.final.out counters identical, the SAM
    bodies the same multiset of records (thread interleaving reorders them): record count + order-independent sum of record hashes."""
    from oracle import refstar
    if not all(os.path.isfile(p + f) for p in (ref, new) for f in ("Aligned.out.sam", "SJ.out.tab", "Log.final.out")):
        return None
    (na, ha), (nb, hb) = sam_digest(ref + "Aligned.out.sam"), sam_digest(new + "Aligned.out.sam")
    return {"sam_records_reference": na, "sam_records_star_amd": nb, "sam_multiset_identical": na == nb and ha == hb,
            "sj_out_tab_identical": open(ref + "SJ.out.tab", "rb").read() == open(new + "SJ.out.tab", "rb").read(),
            "sj_out_tab_lines": sum(1 for _ in open(new + "SJ.out.tab")),
            "log_final_counters_identical": refstar.final_log_counters(ref + "Log.final.out") == refstar.final_log_counters(new + "Log.final.out")}


# ---------------------------------------------------------------------------------------------------------------------

def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    if not torch.cuda.is_available():
        raise RuntimeError("bench.py needs an MI355X: the hot path has no CPU fallback")
    dev = torch.device("cuda", local_rank)
    notes = []

    def log(s):
        notes.append("[%.0f s] %s" % (time.time() - T_START, s))
        print("bench: " + notes[-1], file=sys.stderr, flush=True)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    if not os.path.isfile(os.path.join(ROOT, "star_amd", "lib", "libstaramd_cli.so")):
        raise RuntimeError("star_amd/lib/libstaramd_cli.so is missing (python -c 'import __graft_entry__ as g; g.build()')")
    mb = args.genome_mb
    n_total = (args.steps + args.warmup) * args.reads
    if rank == 0:
        g, ginfo = build_genome(args, mb, log)
    barrier()
    if rank != 0:
        g, ginfo = build_genome(args, mb, log)          # cached by rank 0
    idx = os.path.join(g, "idx")
    run_dir = os.path.join(g, "run_w%d_n%d" % (world, n_total))
    t = time.time()
    fq = make_reads(args, g, run_dir, "reads_r%d" % rank, n_total, 7000 + rank)
    log("rank %d: %d pairs of reads in %.1f s" % (rank, n_total, time.time() - t))
    outp = os.path.join(run_dir, "gpu_r%d_" % rank)
    threads = args.host_threads or max(4, min(64, (os.cpu_count() or 8) // world))
    argv = ["--runMode", "alignReads", "--genomeDir", idx, "--readFilesIn"] + fq + ["--outFileNamePrefix", outp, "--runThreadN", str(threads),
            "--gpuBatchReads", str(args.reads), "--benchWarmupReads", str(args.warmup * args.reads), "--readMapNumber", str(n_total)]
    # (the front end runs STARAMD_CONTEXTS_PER_GPU engine contexts per GPU, default 2: two mapper threads over ONE resident index, so that the copies
    # and the low-occupancy tails of one batch overlap with the kernels of the next; rep.nContexts says how many ran)
    argv += ["--gpuDevice", str(local_rank)]
    if world > 1:          # the ranks load the 31 GB index into host memory one after the other (each keeps ~5 GB of it after the upload)
        os.environ["STARAMD_INDEX_LOAD_LOCK"] = os.path.join(args.workdir, "index_load.lock")
    t_clock = {}

    def warmup_done():
        barrier()
        t_clock["t0"] = time.perf_counter()

    sj_ms = {}

    def exchange(h, last):
        if dist is None:
            return 0
        from star_amd import multi_gpu, capi
        t1 = time.perf_counter()
        if last:
            multi_gpu.merge_handle_outputs(capi.host_lib(), h, dist, dev, rank, world)
            sj_ms["ms"] = (time.perf_counter() - t1) * 1e3
        else:
            multi_gpu.exchange_before_phase(capi.host_lib(), h, dist, dev, rank, world)
        return 0

    barrier()
    rc, rep = run_cli(argv, warmup_done, exchange)
    if rc != 0:
        raise RuntimeError("the star_amd pipeline failed with exit code %d" % rc)
    elapsed = float(rep.timedWall)
    barrier()
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        nn = torch.tensor([int(rep.timedReads)], dtype=torch.int64, device=dev)
        dist.all_reduce(nn, op=dist.ReduceOp.SUM)
        timed_reads_all = int(nn.item())
    else:
        timed_reads_all = int(rep.timedReads)
    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return
    steps = max(args.steps, 1)
    lread = 2 * args.read_len + 1
    c, ms, kern, bytes_per_pair = report_dict(rep, lread)
    n = max(int(rep.timedReads), 1)
    dom = max(kern, key=lambda k: kern[k][0])
    dms, dbytes = kern[dom]
    achieved = dbytes / (dms * 1e-3) / 1e9 if dms > 0 else 0.0
    value = timed_reads_all / elapsed / 1e6
    traffic = None; traffic_all = None
    issue = None
    try:        # HBM traffic / issue utilisation from rocprofv3 --pmc passes (profiles/README.md): only when taken on THIS engine source and workload size
        tj = json.load(open(os.path.join(ROOT, "profiles", "r03_pmc_hbm_traffic.json")))
        if tj.get("genome_mb") == mb and tj.get("reads_per_launch") == args.reads:
            # a kernel's figures are reported only while the sources it is written in hash to what they were when the counters were taken
            ok = {k for k, v in tj.items() if isinstance(v, dict) and k in KERNEL_SOURCES and v.get("kernel_src_sha") == kernel_src_sha(k)}
            traffic = tj[dom].get("hbm_bytes_per_launch") if dom in ok else None
            traffic_all = {k: (tj[k]["hbm_bytes_per_launch"] if k in ok else None) for k in KERNEL_SOURCES if k in tj} or None
            issue = {k: (tj[k].get("valu_busy_frac") if k in ok else None) for k in KERNEL_SOURCES if k in tj} or None
    except Exception:
        traffic = None
    n_ctx = max(1, int(rep.nContexts))
    device_s = sum(float(rep.deviceMs[k]) for k in range(n_ctx)) / 1e3
    out = {
        "metric": "million reads aligned/sec (whole node), 2x101 bp PE human-scale index, FASTQ in -> SAM out",
        "value": value, "unit": "Mreads/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u8/u64 integer", "data": "synthetic",
        "config": {"workload": "BASELINE config 2 stand-in: synthetic %d Mb genome (%d chromosomes, %.0f %% of the bases written by repeat families), %d junctions in the index "
                               "(sjdbOverhang %d, SAindex %d bases, index %.1f GB in HBM), %d DISTINCT pairs 2x%d bp per GPU streamed as %d + %d batches of %d "
                               "(85%% spliced, 1%% substitutions, 0.1%% N); FASTQ text in -> Aligned.out.sam + SJ.out.tab out, index load excluded"
                               % (mb, max(1, min(24, mb // 40)), 100 * ginfo.get("repeat_bases_fraction", 0), ginfo.get("junctions_in_index", 0), args.read_len - 1,
                                  ginfo.get("SAindexNbases", 0), ginfo.get("index_bytes", 0) / 1e9, n_total, args.read_len, args.warmup, args.steps, args.reads),
                   "reads_per_gpu_per_step": args.reads, "genome_mb": mb, "host_threads_per_rank": threads,
                   "parallelism": "reads sharded over %d GPU(s), one process per GPU, full index replica each" % world},
        "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "traffic": traffic, "traffic_per_kernel": traffic_all, "issue": issue, "engine_src_sha": engine_src_sha(), "algorithmic_bytes_per_launch": dbytes, "kernel_ms": dms, "per_kernel_ms": ms,
                     "algorithmic_bytes_per_pair_whole_path": bytes_per_pair,
                     "note": "per launch = per batch of %d pairs, averaged over the %d timed batches of rank 0 (HIP events on the engine's stream)" % (args.reads, int(rep.batches))},
        "counters_per_pair": {k: (v / n if not isinstance(v, dict) else v) for k, v in c.items()},
        "pipeline": {"timed_wall_s": float(rep.timedWall), "device_s_sum_over_contexts": device_s,
                     "device_s_note": "HIP-event time of the batches summed over the engine contexts of the GPU: with two contexts their launches overlap, so the sum exceeds the wall time; "
                                      "kernel_ms_exclusive has the one-context figures",
                     "map_batch_call_s": sum(float(rep.deviceBusy[k]) for k in range(n_ctx)), "engine_contexts_per_gpu": n_ctx // max(1, int(rep.nDevices)), "parse_busy_s": float(rep.parseBusy), "convert_busy_s": float(rep.convertBusy), "postmap_write_busy_s": float(rep.emitBusy),
                     "reader_note": "the reader is two pipeline stages on two threads: parse_* = input + line table (sah_fill_slot), convert_* = text -> numeric batch (sah_convert_slot)",
                     "convert_Mreads_s": n / float(rep.convertBusy) / 1e6 if rep.convertBusy > 0 else None,
                     "parse_Mreads_s": n / float(rep.parseBusy) / 1e6 if rep.parseBusy > 0 else None,
                     "postmap_write_Mreads_s": n / float(rep.emitBusy) / 1e6 if rep.emitBusy > 0 else None,
                     "finish_s": float(rep.finishSeconds),
                     "finish_what": "inside timed_wall_s, after the last batch: last SAM writes, junction collapse + filter + SJ.out.tab, Log.final.out",
                     "genome_load_s": float(rep.genomeLoadSeconds), "index_upload_s": float(rep.indexUploadSeconds)},
        "index_build": ginfo, "sj_merge_ms": sj_ms.get("ms"),
    }
    if dist is not None:
        dist.destroy_process_group()
    ref_prefix = os.path.join(run_dir, "cpu_")
    if not args.no_cpu_baseline and world == 1:
        try:
            out["cpu_baseline"] = cpu_baseline(idx, fq, ref_prefix, n_total, log)
        except Exception as e:
            out["cpu_baseline"] = {"error": repr(e)[:400]}
        log("cpu_baseline done")
        try:
            out["full_size_parity"] = full_size_parity(ref_prefix, outp)
            log("full_size_parity done")
        except Exception as e:
            out["full_size_parity"] = {"error": repr(e)[:300]}
    if world == 1 and not args.no_extra_legs:
        for name, fn in (("kernel_ms_exclusive", lambda: exclusive_leg(args, idx, fq, run_dir, threads)),
                         ("host_budget", lambda: host_budget_leg(args, idx, fq, run_dir)),
                         ("all_transcripts", lambda: all_transcripts_leg(args, g, idx, log)),
                         ("config1", lambda: config1_leg(args, log))):
            if time.time() - T_START > args.budget_s:
                out[name] = {"skipped": "time budget"}; continue
            try:
                out[name] = fn()
            except Exception as e:
                out[name] = {"error": repr(e)[:400]}
            log(name + " done")
        ex = out.get("kernel_ms_exclusive")
        if isinstance(ex, dict) and "per_kernel_ms" in ex and n_ctx > max(1, int(rep.nDevices)):
            # The timed region runs two engine contexts per GPU: the HIP events around a launch then also cover the time slices the other context's
            # launches get, i.e. they no longer measure the kernel.  The roofline figures are therefore taken from the one-context leg (same index,
            # same reads, same process, minutes later: every launch has the GPU to itself, which is also what the committed rocprofv3 summary
            # shows); the overlapped figures stay beside them.
            r = out["roofline"]
            r["per_kernel_ms_timed_region_two_contexts"] = r["per_kernel_ms"]; r["kernel_ms_timed_region_two_contexts"] = r["kernel_ms"]
            r["per_kernel_ms"] = ex["per_kernel_ms"]
            ems = ex["per_kernel_ms"]
            kd = max(("k_seed_search", "k_windows", "k_stitch_win"), key=lambda k: ems[k])
            r["kernel"] = kd; r["kernel_ms"] = ems[kd]
            r["algorithmic_bytes_per_launch"] = kern[kd][1]
            r["achieved"] = kern[kd][1] / (ems[kd] * 1e-3) / 1e9; r["frac"] = r["achieved"] / HBM_PEAK_GBS
            if traffic_all:
                r["traffic"] = traffic_all.get(kd)
            r["kernel_launches"] = {"k_stitch_win": "the stitch stage of pass 0 = one k_stitch_lane launch (lane per read, cheapest cost classes) + one k_stitch_win launch (wavefront per window, the rest)",
                                    "k_windows": "k_windows x 2 + k_windows_big"}
            r["note"] = ("per launch = per batch of %d pairs; kernel times by HIP events on the engine's stream in the ONE-context leg (kernel_ms_exclusive: %d batches, a launch has "
                         "the GPU to itself); the timed region runs two contexts per GPU whose launches overlap (per_kernel_ms_timed_region_two_contexts)" % (args.reads, 4))
    if not args.no_sweep and world == 1:
        out["index_size_sweep"] = sweep(args, mb, out, log)
    if not args.no_two_pass and world == 1 and time.time() - T_START < args.budget_s:
        try:
            out["two_pass_end_to_end"] = two_pass(args, idx, fq, run_dir, threads)
        except Exception as e:
            out["two_pass_end_to_end"] = {"error": repr(e)[:300]}
    if world == 1 and not args.no_extra_legs and isinstance(out.get("kernel_ms_exclusive"), dict) and "per_kernel_ms" in out["kernel_ms_exclusive"]:
        # last: kernels that have never run on hardware are timed here, each in a child process with a time limit; nothing else of the line depends on them
        try:
            out["variants"] = variants_leg(args, idx, fq, run_dir, threads, out["kernel_ms_exclusive"], log)
        except Exception as e:
            out["variants"] = {"error": repr(e)[:300]}
        log("variants done")
    out["bench_wall_s"] = time.time() - T_START
    out["notes"] = notes
    print(json.dumps(out))


def _cli_leg(argv, lread, env=None):
    """one run of the product pipeline with extra environment; returns (rep, summary dict)"""
    old = {}
    for k, v in (env or {}).items():
        old[k] = os.environ.get(k); os.environ[k] = v
    try:
        rc, rep = run_cli(argv)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    if rc:
        raise RuntimeError("exit code %d" % rc)
    c, ms, kern, bpp = report_dict(rep, lread)
    n = max(int(rep.timedReads), 1)
    return rep, {"Mreads_s": n / float(rep.timedWall) / 1e6, "per_kernel_ms": ms, "timed_reads": n,
                 "parse_Mreads_s": n / float(rep.parseBusy) / 1e6 if rep.parseBusy > 0 else None,
                 "convert_Mreads_s": n / float(rep.convertBusy) / 1e6 if rep.convertBusy > 0 else None,
                 "postmap_write_Mreads_s": n / float(rep.emitBusy) / 1e6 if rep.emitBusy > 0 else None,
                 "device_Mreads_s": n / max(1e-9, sum(float(rep.deviceMs[k]) for k in range(max(1, int(rep.nContexts)))) / 1e3) / 1e6,
                 "engine_contexts": int(rep.nContexts), "counters_per_pair": {k: v / n for k, v in c.items() if not isinstance(v, dict)}}


def exclusive_leg(args, idx, fq, run_dir, threads):
    """Per-kernel times with ONE engine context (no second batch sharing the GPU): what a launch costs when it has the device to itself.  The bench
    line's own per_kernel_ms are taken in the timed region, where the launches of two contexts overlap."""
    nb, w = 4, 1
    argv = ["--runMode", "alignReads", "--genomeDir", idx, "--readFilesIn"] + fq + ["--outFileNamePrefix", os.path.join(run_dir, "excl_"), "--runThreadN", str(threads),
            "--gpuBatchReads", str(args.reads), "--benchWarmupReads", str(w * args.reads), "--readMapNumber", str((nb + w) * args.reads)]
    rep, d = _cli_leg(argv, 2 * args.read_len + 1, {"STARAMD_CONTEXTS_PER_GPU": "1"})
    return d


# kernels / knobs that are OFF by default because they have no hardware number yet (written when no GPU minutes were left): the leg below times each of them
# exactly like kernel_ms_exclusive (one context, same index, same reads) and checks its output against that leg's, so that the bench run at the end of a round is
# also their A/B.  A variant that wins becomes the default (and leaves this list) in the next round; one that loses is deleted.
VARIANTS = [      # (most wanted first: the leg has a time budget of its own)
    ("base_repeat", {}),      # the default kernels once more, in a child like the others: what two identical runs differ by
    ("seed_read_4waves", {"STARAMD_SEED_FLAT": "4"}),
    ("seed_staged_4waves", {"STARAMD_SEED_FLAT": "6"}),
    ("seed_flat_8waves", {"STARAMD_SEED_FLAT": "1"}),
    ("lane_class_post_2", {"STARAMD_LANE_CLASS_POST": "2"}),
    ("seed_flat_4waves", {"STARAMD_SEED_FLAT": "3"}),
    ("seed_read_6waves", {"STARAMD_SEED_FLAT": "5"}),
    ("lane_class_post_1", {"STARAMD_LANE_CLASS_POST": "1"}),
    ("lane_class_post_2_cap6", {"STARAMD_LANE_CLASS_POST": "2", "STARAMD_LANE_CLASS": "6"}),
    ("seed_flat_6waves", {"STARAMD_SEED_FLAT": "2"}),
    ("seed_read_4waves_half_grid", {"STARAMD_SEED_FLAT": "4", "STARAMD_SEED_LANES": "131072"}),      # ~3 reads per lane instead of ~1.5: better balance inside a wavefront, fewer wavefronts in flight
    # not a kernel: the same 2 M pairs as 1 + 1 batches of a million (the launches of a batch end in tails of a few wavefronts; per pair they weigh less in a larger batch)
    ("batch_1M", {"_batch_reads": "1000000"}),
]


def variants_leg(args, idx, fq, run_dir, threads, base, log):
    """A/B of the experimental kernels against the defaults of the kernel_ms_exclusive leg (`base` = its summary; its output files are the reference here)."""
    nb, w = 4, 1
    res = {"what": "experimental kernels / knobs, OFF by default, timed like kernel_ms_exclusive (ONE engine context, %d batches of %d pairs) and compared with its output "
                   "(SAM multiset, SJ.out.tab); base = the default kernels in the same process" % (nb, args.reads),
           "base_per_kernel_ms": base.get("per_kernel_ms")}
    base_prefix = os.path.join(run_dir, "excl_")
    base_dig = sam_digest(base_prefix + "Aligned.out.sam")
    base_sj = open(base_prefix + "SJ.out.tab", "rb").read()
    timed_out = set()
    t_leg = time.time()
    for name, env in VARIANTS:
        if time.time() - t_leg > 600:
            res[name] = {"skipped": "the leg's own time budget (600 s)"}; continue
        if time.time() - T_START > args.budget_s:
            res[name] = {"skipped": "time budget"}; continue
        pre = os.path.join(run_dir, "var_")
        total = (nb + w) * args.reads
        breads = int(env.get("_batch_reads", args.reads))
        wreads = w * args.reads if breads == args.reads else breads            # (a different batch size: one warm-up batch, the rest timed; the same reads in all)
        argv = ["--runMode", "alignReads", "--genomeDir", idx, "--readFilesIn"] + fq + ["--outFileNamePrefix", pre, "--runThreadN", str(threads),
                "--gpuBatchReads", str(breads), "--benchWarmupReads", str(wreads), "--readMapNumber", str(total)]
        try:
            e = {k: v for k, v in env.items() if not k.startswith("_")}; e["STARAMD_CONTEXTS_PER_GPU"] = "1"
            # in a child interpreter: a kernel that has never run on hardware may fault, and a GPU memory fault aborts the process it happens in
            for f in ("Aligned.out.sam", "SJ.out.tab"):
                if os.path.exists(pre + f):
                    os.remove(pre + f)
            fam = "_".join(name.split("_")[:2])
            if fam in timed_out:
                res[name] = {"env": env, "skipped": "an earlier variant of this family ran into its time limit"}; continue
            try:
                p = subprocess.run([sys.executable, os.path.abspath(__file__), "--variant-child", json.dumps([argv, 2 * args.read_len + 1, e])], timeout=180, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
            except subprocess.TimeoutExpired:
                timed_out.add(fam)
                raise RuntimeError("time limit (180 s)")
            if p.returncode != 0:
                raise RuntimeError("child exit code %d: %s" % (p.returncode, p.stderr[-200:]))
            d = json.loads(p.stdout.strip().splitlines()[-1])
            dig = sam_digest(pre + "Aligned.out.sam")
            res[name] = {"env": env, "batch_pairs": breads, "per_kernel_ms": d["per_kernel_ms"], "Mreads_s": d["Mreads_s"], "lane_items_per_pair": d["counters_per_pair"].get("nLaneItems"),
                         "sam_multiset_identical_to_base": dig == base_dig, "sj_out_tab_identical_to_base": open(pre + "SJ.out.tab", "rb").read() == base_sj}
        except Exception as ex:
            res[name] = {"env": env, "error": repr(ex)[:300]}
        log("variant %s done" % name)
    return res


def host_budget_leg(args, idx, fq, run_dir):
    """The host at the budget of an 8-GPU node: the same pipeline with cores/8 host threads for this GPU."""
    th = max(4, (os.cpu_count() or 8) // 8)
    nb, w = 8, 2
    argv = ["--runMode", "alignReads", "--genomeDir", idx, "--readFilesIn"] + fq + ["--outFileNamePrefix", os.path.join(run_dir, "hb_"), "--runThreadN", str(th),
            "--gpuBatchReads", str(args.reads), "--benchWarmupReads", str(w * args.reads), "--readMapNumber", str((nb + w) * args.reads)]
    rep, d = _cli_leg(argv, 2 * args.read_len + 1)
    d["host_threads"] = th
    d["what"] = "one GPU, --runThreadN = cores / 8: the host share of one rank on an 8-GPU node"
    return d


def all_transcripts_leg(args, g, idx, log):
    """2x150, 1 % errors, 5 % chimeric pairs, --chimSegmentMin 12: chimeric detection wants EVERY transcript of every window
    (resultSelect 0, no window pruning: stitchWindowAligns.cpp:245-247).  Same index (sjdbOverhang 100).  Parity against the reference on a bounded
    sample: SAM multiset, SJ.out.tab, Chimeric.out.junction."""
    from oracle import refstar
    L = 150; nb, w = 4, 1
    n_total = (nb + w) * args.reads
    rd = os.path.join(g, "chim_n%d" % n_total)
    fq = make_reads(args, g, rd, "chim", n_total, 8100, read_len=L, chim_rate=0.05)
    flags = ["--chimSegmentMin", "12", "--chimOutType", "Junctions"]
    argv = ["--runMode", "alignReads", "--genomeDir", idx, "--readFilesIn"] + fq + ["--outFileNamePrefix", os.path.join(rd, "gpu_"), "--runThreadN", str(max(4, min(64, os.cpu_count() or 8))),
            "--gpuBatchReads", str(args.reads), "--benchWarmupReads", str(w * args.reads), "--readMapNumber", str(n_total)] + flags
    rep, d = _cli_leg(argv, 2 * L + 1)
    d["workload"] = "%d pairs 2x%d, 1%% substitutions, 5%% chimeric pairs, --chimSegmentMin 12 (every transcript of every window returned, no window pruning)" % (n_total, L)
    # parity on a sample both can do quickly: the first 200 k pairs
    ns = min(200000, n_total)
    p_new, p_ref = os.path.join(rd, "gpuS_"), os.path.join(rd, "ref_")
    rc, _ = run_cli(["--runMode", "alignReads", "--genomeDir", idx, "--readFilesIn"] + fq + ["--outFileNamePrefix", p_new, "--runThreadN", "32", "--gpuBatchReads", str(args.reads), "--readMapNumber", str(ns)] + flags)
    if rc:
        raise RuntimeError("sample run: exit code %d" % rc)
    t = time.perf_counter()
    refstar.align(idx, fq, p_ref, threads=min(64, os.cpu_count() or 8), extra=["--readMapNumber", str(ns)] + flags, timeout=900)
    d["reference_sample_s"] = time.perf_counter() - t
    par = full_size_parity(p_ref, p_new) or {}
    cj = [sorted(l for l in open(p + "Chimeric.out.junction", "rb") if not l.startswith(b"#")) for p in (p_ref, p_new)]
    par["chimeric_junctions"] = len(cj[1]); par["chimeric_junction_identical"] = cj[0] == cj[1]
    d["parity_vs_reference"] = par; d["parity_sample_pairs"] = ns
    return d


def config1_leg(args, log):
    """BASELINE config 1 stand-in: yeast-size genome (12 Mb), single-end 1x50 reads; the reference with --runThreadN 1 on 100 k of them beside it."""
    from oracle import refstar
    mb = 12
    sub = argparse.Namespace(**vars(args)); sub.read_len = 50
    g, ginfo = build_genome(sub, mb, log)
    nb, w = 4, 1
    n_total = (nb + w) * args.reads
    rd = os.path.join(g, "se_n%d" % n_total)
    fq = make_reads(sub, g, rd, "se", n_total, 8200, read_len=50)[:1]
    idx = os.path.join(g, "idx")
    argv = ["--runMode", "alignReads", "--genomeDir", idx, "--readFilesIn"] + fq + ["--outFileNamePrefix", os.path.join(rd, "gpu_"), "--runThreadN", str(max(4, min(64, os.cpu_count() or 8))),
            "--gpuBatchReads", str(args.reads), "--benchWarmupReads", str(w * args.reads), "--readMapNumber", str(n_total)]
    rep, d = _cli_leg(argv, 50)
    d["workload"] = "%d single-end reads 1x50, synthetic %d Mb genome (yeast size), SAindex %d bases" % (n_total, mb, ginfo.get("SAindexNbases", 0))
    ns = 100000
    p_new, p_ref = os.path.join(rd, "gpuS_"), os.path.join(rd, "ref_")
    rc, _ = run_cli(["--runMode", "alignReads", "--genomeDir", idx, "--readFilesIn"] + fq + ["--outFileNamePrefix", p_new, "--runThreadN", "16", "--gpuBatchReads", str(args.reads), "--readMapNumber", str(ns)])
    if rc:
        raise RuntimeError("sample run: exit code %d" % rc)
    t = time.perf_counter(); refstar.align(idx, fq, p_ref, threads=1, extra=["--readMapNumber", str(ns)], timeout=900); t_full = time.perf_counter() - t
    t = time.perf_counter(); refstar.align(idx, fq, p_ref + "l_", threads=1, extra=["--readMapNumber", "1"], timeout=900); t_load = time.perf_counter() - t
    d["reference_1_thread_Mreads_s"] = ns / max(t_full - t_load, 1e-3) / 1e6
    d["parity_vs_reference"] = full_size_parity(p_ref, p_new); d["parity_sample_reads"] = ns
    return d


def sweep(args, main_mb, main_out, log):
    """The same pipeline on smaller indices (100 / 400 / 1000 Mb): how the kernels behave as the index outgrows the 256 MB Infinity Cache."""
    rows = []
    lread = 2 * args.read_len + 1
    for mb in (100, 400, 1000):
        if mb >= main_mb or time.time() - T_START > args.budget_s:
            continue
        try:
            g, ginfo = build_genome(args, mb, log)
            nb, w = 5, 1
            n_total = (nb + w) * args.reads
            rd = os.path.join(g, "sweep_n%d" % n_total)
            fq = make_reads(args, g, rd, "reads", n_total, 9000 + mb)
            argv = ["--runMode", "alignReads", "--genomeDir", os.path.join(g, "idx"), "--readFilesIn"] + fq + ["--outFileNamePrefix", os.path.join(rd, "gpu_"),
                    "--runThreadN", str(max(4, min(64, os.cpu_count() or 8))), "--gpuBatchReads", str(args.reads), "--benchWarmupReads", str(w * args.reads), "--readMapNumber", str(n_total)]
            rc, rep = run_cli(argv)
            if rc:
                rows.append({"genome_mb": mb, "error": "exit code %d" % rc}); continue
            c, ms, kern, bpp = report_dict(rep, lread)
            rows.append({"genome_mb": mb, "Mreads_s": int(rep.timedReads) / float(rep.timedWall) / 1e6, "per_kernel_ms": ms, "index_generate_s": ginfo.get("index_generate_s"),
                         "junctions_in_index": ginfo.get("junctions_in_index"), "algorithmic_bytes_per_pair": bpp})
        except Exception as e:
            rows.append({"genome_mb": mb, "error": repr(e)[:300]})
    rows.append({"genome_mb": main_mb, "Mreads_s": main_out["value"], "per_kernel_ms": main_out["roofline"]["per_kernel_ms"],
                 "index_generate_s": main_out["index_build"].get("index_generate_s"), "junctions_in_index": main_out["index_build"].get("junctions_in_index"),
                 "algorithmic_bytes_per_pair": main_out["roofline"]["algorithmic_bytes_per_pair_whole_path"]})
    return rows


def two_pass(args, idx, fq, run_dir, threads):
    """--twopassMode Basic on two batches of the workload: 1st pass on the GPU without SAM, junction insertion, index
    replaced in HBM, 2nd pass."""
    n = min(10, args.steps + args.warmup) * args.reads          # (4 M pairs in the default run: the insertion is amortised as in a real run)
    argv = ["--runMode", "alignReads", "--genomeDir", idx, "--readFilesIn"] + fq + ["--outFileNamePrefix", os.path.join(run_dir, "cli2p_"), "--runThreadN", str(threads),
            "--gpuBatchReads", str(args.reads), "--twopassMode", "Basic", "--readMapNumber", str(n)]
    rc, rep = run_cli(argv)
    if rc:
        return {"error": "exit code %d" % rc}
    sjdb = sum(1 for _ in open(os.path.join(run_dir, "cli2p__STARgenome", "sjdbList.out.tab")))
    return {"value": n / float(rep.wallMapping) / 1e6, "unit": "Mreads/s (each read counted once, both passes + insertion in the wall time)", "reads": n,
            "wall_s": float(rep.wallMapping), "pass1_plus_insertion_plus_reupload_s": float(rep.pass1Seconds), "junctions_in_index_after_pass1": sjdb,
            "host_threads": threads, "what": "star_amd --twopassMode Basic end to end (index load excluded)"}


if __name__ == "__main__":
    if sys.argv[1:2] == ["--make-reads"]:
        _make_reads_child(*json.loads(sys.argv[2]))
    elif sys.argv[1:2] == ["--variant-child"]:
        _argv, _lread, _env = json.loads(sys.argv[2])
        _rep, _d = _cli_leg(_argv, _lread, _env)
        print(json.dumps(_d))
    elif sys.argv[1:2] == ["--sam-digest"]:
        print("%d %d" % _sam_digest_child(sys.argv[2]))
    else:
        main()
