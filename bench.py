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
`--gpus N` with N > 1 and no WORLD_SIZE in the environment re-executes itself under torch.distributed.run with N ranks.

Output: ONE compact JSON line (< 4 KB) on stdout of rank 0.  Everything else -- the optional legs (host at the 8-GPU thread budget, configs 1 / 4 / 5,
index-size sweep), per-leg counters, notes -- goes to <workdir>/bench_extra.json (path in the line under "extra"), written by a child process
so that a fault in an optional leg cannot take the line with it.
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
    ap.add_argument("--no-extra-legs", action="store_true", help="skip the optional legs (bench_extra.json): host_budget / config 5 / config 1 / sweep / 2-pass")
    ap.add_argument("--no-exclusive", action="store_true", help="skip the one-context leg the roofline kernel times come from (they are then the overlapped two-context figures)")
    ap.add_argument("--cpu-selftest", action="store_true", help="TEST of this script's plumbing on a box without a GPU: gloo, the oracle behind the front end, a tiny genome; the line says selftest")
    ap.add_argument("--keep-run-dirs", action="store_true", help="do not delete the run directories (FASTQ + SAM) that earlier bench.py runs of other sizes left in the genome cache")
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
    exe = SELFTEST_EXE if getattr(args, "cpu_selftest", False) else os.path.join(ROOT, "star_amd", "bin", "star_amd")
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
STAGE_NAMES = ["k_seed_search", "k_windows", "k_order", "k_stitch_win", "k_stitch_verify+replay+finish", "k_scan+k_gather", "device_total", "k_windows:middle+last launch"]


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
    c["nSkippedLeaves"], c["nRewalkWin"] = int(rep.counters[47]), int(rep.counters[48])                                            # dev.h: DC_nSkippedLeaves, DC_nRewalkWin
    for k in ("nNodes", "nLeaves", "nStitchCalls", "nExtendCalls"):      # kept by the profile / shadow builds only
        c.pop(k, None)
    ms = dict(zip(STAGE_NAMES, [float(x) / nb for x in rep.stageMs]))       # per launch = per batch
    # Algorithmic bytes: what the algorithm must fetch / write, from the engine's own counters
    bytes_seed = 8 * c["nSAi"] + 8 * c["nSAprobe"] + c["nGcmp"] + n * lread + 24 * c["nSeeds"]
    bytes_win = 8 * c["nSAenum"] + 24 * c["nSeeds"] + 24 * c["nWA"]
    walked = max(0, c["nWindows"] - c["nPrunedWin"]) if c["nWindows"] else n           # the packed read is staged once per WALKED window (pruned windows are never touched)
    bytes_stitch = c["nGstitch"] + 24 * c["nWA"] + (lread // 2) * walked + 96 * c["nTrOut"] + 32 * 2 * c["nTrOut"]
    # (rounds 1-3 counted the packed read once per window, walked or not: kept beside the figure so that the series stays comparable)
    c["_bytes_stitch_round3_definition"] = (bytes_stitch + (lread // 2) * ((c["nWindows"] if c["nWindows"] else n) - walked)) / nb
    kern = {"k_seed_search": (ms["k_seed_search"], bytes_seed / nb), "k_windows": (ms["k_windows"], bytes_win / nb), "k_stitch_win": (ms["k_stitch_win"], bytes_stitch / nb)}
    if os.environ.get("STARAMD_PROFILE_BUILD"):          # libstaramd.so built with -DSTARAMD_PROFILE: shader-clock cycles per section, summed over waves
        pn = ["walk(all)", "coopStitch", "coopExtend", "finalize(all)", "recordCandidate", "-", "-", "wave_lifetime", "windows:passA", "windows:flanks", "windows:passB_enumerate+owner",
              "windows:passB_assign", "windows:emission", "finalize:extends", "finalize:filters+score", "finalize:candidate+log"]
        raw = [int(x) for x in rep.counters]
        c["profile_cycles_per_pair"] = dict(zip(pn, [v / n for v in raw[21:37]]))
        c["profile_counts_per_pair"] = dict(zip(["nNodes", "nLeaves", "nStitchCalls", "nExtendCalls", "nLeavesBound", "nLeavesEarly"], [v / n for v in raw[8:12] + raw[49:51]]))
    return c, ms, kern, (bytes_seed + bytes_win + bytes_stitch) / n


# ---------------------------------------------------------------------------------------------------------------------
# CPU baseline + parity

def cpu_baseline(idx, fq, out_prefix, n_pairs, log=lambda s: None):
    """The reference itself (oracle/_ref/STAR, built from /root/reference by oracle/Makefile.ref), same index, same FASTQ, default parameters.
    STAR deals input to its threads in chunks of limitIObufferSize[0]/nMates bytes (ReadAlignChunk_processChunks.cpp:14-30, Parameters.cpp:1160);
    the default (30 MB -> ~67 k pairs) would leave most of 256 threads without a chunk, so the input buffer is set to 2 MB (~4.4 k pairs per
    chunk).  Mapping time = wall(run) - wall(run that only loads the index).  Two configurations: the box's best thread count (cores / 4: the
    reference reads and splits its input under one mutex, more threads only wait) on the WHOLE FASTQ -- its output is what full_size_parity
    compares with -- and all cores (north_star: --runThreadN = all host cores) on a bounded sample (the first 4 M pairs)."""
    from oracle import refstar
    ncpu = os.cpu_count() or 1
    small = ["--limitIObufferSize", "2000000", "50000000"]

    def run(nmap, threads, prefix):
        t = time.perf_counter()
        refstar.align(idx, fq, prefix, threads=threads, extra=["--readMapNumber", str(nmap)] + small, timeout=900)
        log("reference STAR: %d reads, %d threads: %.1f s" % (nmap, threads, time.perf_counter() - t))
        return time.perf_counter() - t
    run(1, ncpu, out_prefix + "ld_")                                    # page cache
    t_load = run(1, ncpu, out_prefix + "ld_")
    # the reference's best thread count on this box: the CPUs the container may use (measured on a 16-CPU quota behind 256 hardware threads: 16 threads 0.235,
    # 24 -> 0.230, 32 -> 0.198, 48 -> 0.183, 64 -> 0.231, 256 -> 0.190 M pairs/s; profiles/r04_host_diagnostics.txt)
    best_th = max(1, min(ncpu, effective_cpus()))
    n_all = min(n_pairs, 4000000)
    v_all = n_all / max(run(n_all, ncpu, out_prefix + "all_") - t_load, 1e-3) / 1e6
    v_best = n_pairs / max(run(n_pairs, best_th, out_prefix) - t_load, 1e-3) / 1e6       # last: its outputs stay for the parity check
    # value = the reference at its best thread count on this box (the comparator); all_threads_value = --runThreadN <all host cores> as north_star words it (slower: the
    # box's CPU quota is 16 of 256 hardware threads, and the reference reads its input under one mutex)
    return {"value": round(v_best, 4), "unit": "Mreads/s", "cores": best_th, "kind": "reference", "all_threads": ncpu, "all_threads_value": round(v_all, 4),
            "sample": "STAR 2.7.11b, same index and FASTQ, --limitIObufferSize 2000000; all %d cores on the first %d pairs, %d threads on all %d; index load (%.1f s) subtracted"
                      % (ncpu, n_all, best_th, n_pairs, t_load), "index_load_s": t_load}


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

SELFTEST_CLI_LIB = os.path.join(ROOT, "oracle", "_build", "libstaramd_cli_oracle.so")      # --cpu-selftest only (test of the plumbing, never a measurement)
SELFTEST_EXE = os.path.join(ROOT, "oracle", "_build", "star_amd_oracle_cli")
_CLI_LIB = None       # set by --cpu-selftest


def _run_cli(argv, *hooks):
    return run_cli(argv, *hooks, lib_path=_CLI_LIB)


def effective_cpus():
    """CPUs this process may actually use: the smallest of the online CPUs, the affinity mask and the cgroup CPU quota (cpu.max).  The GPU boxes show 256
    hardware threads to a container whose quota is 16 CPUs: 64 + 32 + 16 busy threads are then throttled by the scheduler in 100 ms periods, and the
    single-threaded sections of the host pipeline stall for whole periods (measured: profiles/r04_host_diagnostics.txt)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    for path in ("/sys/fs/cgroup/cpu.max",):
        try:
            q, per = open(path).read().split()[:2]
            if q != "max":
                n = min(n, max(1, int(int(q) / int(per))))
        except Exception:
            pass
    try:        # cgroup v1
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0 and per > 0:
            n = min(n, max(1, q // per))
    except Exception:
        pass
    return n


def cpu_throttle():
    """(periods in which the container was throttled, microseconds throttled) so far: cgroup v2 cpu.stat"""
    try:
        d = dict(l.split() for l in open("/sys/fs/cgroup/cpu.stat"))
        return int(d.get("nr_throttled", 0)), int(d.get("throttled_usec", 0))
    except Exception:
        return None


def loaded_libs():
    """shared objects of this repository mapped into the process (the line records which native code ran)"""
    out = set()
    try:
        for l in open("/proc/self/maps"):
            p = l.rstrip().split(" ")[-1]
            if p.endswith(".so") and (p.startswith(ROOT) or "staramd" in p or "oracle" in p):
                out.add(os.path.relpath(p, ROOT) if p.startswith(ROOT) else p)
    except OSError:
        pass
    return sorted(out)


def _free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def respawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: the same command line under torch.distributed.run, one rank per GPU of this node."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    sys.stdout.flush(); sys.stderr.flush()
    os.execvpe(cmd[0], cmd, env)


def main():
    global _CLI_LIB
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        respawn_ranks(args)                     # does not return
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != max(1, args.gpus):
        raise RuntimeError("bench.py --gpus %d was launched with WORLD_SIZE=%d: one rank per GPU, and the line reports n_gpus = the number of ranks that ran" % (args.gpus, world))
    selftest = args.cpu_selftest
    if selftest:
        _CLI_LIB = SELFTEST_CLI_LIB
    else:
        # the timed run maps with the product libraries and nothing else: the variables that redirect the front end / the engine to test stand-ins are refused
        for v in ("STARAMD_CLI_LIB", "STARAMD_ENGINE_LIB"):
            if os.environ.get(v):
                raise RuntimeError("%s is set (%s): bench.py measures star_amd/lib/libstaramd_cli.so + libstaramd.so only" % (v, os.environ[v]))
    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if selftest:
            dist.init_process_group(backend="gloo")
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        if dist.get_world_size() != args.gpus:
            raise RuntimeError("process group of %d ranks for --gpus %d" % (dist.get_world_size(), args.gpus))
    if not selftest and not torch.cuda.is_available():
        raise RuntimeError("bench.py needs an MI355X: the hot path has no CPU fallback")
    dev = torch.device("cpu") if selftest else torch.device("cuda", local_rank)
    notes = []

    def log(s):
        notes.append("[%.0f s] %s" % (time.time() - T_START, s))
        print("bench: " + notes[-1], file=sys.stderr, flush=True)

    def barrier():
        if dist is not None:
            dist.barrier()
        if not selftest:
            torch.cuda.synchronize(dev)

    cli_lib = _CLI_LIB or os.path.join(ROOT, "star_amd", "lib", "libstaramd_cli.so")
    if not os.path.isfile(cli_lib):
        raise RuntimeError(cli_lib + " is missing (python -c 'import __graft_entry__ as g; g.build()')")
    if selftest:          # tiny stand-in workload, index built by the oracle-backed binary
        args.genome_mb = min(args.genome_mb, 1) or 1; args.reads = min(args.reads, 2000); args.no_extra_legs = True; args.no_cpu_baseline = True; args.no_exclusive = True
    mb = args.genome_mb
    n_total = (args.steps + args.warmup) * args.reads
    if rank == 0:
        g, ginfo = build_genome(args, mb, log)
    barrier()
    if rank != 0:
        g, ginfo = build_genome(args, mb, log)          # cached by rank 0
    idx = os.path.join(g, "idx")
    run_dir = os.path.join(g, "run_w%d_n%d" % (world, n_total))
    if rank == 0:
        # runs of other world sizes on this box (the driver's 1 / 2 / 4 / 8-GPU series) leave ~11 GB of FASTQ + SAM per rank in tmpfs: dropped before this one adds its own.
        # Only directories that a bench.py run marked as its own (.bench_run_owner = pid of its rank 0) and whose owner is no longer alive: a bench running beside this
        # one on the same cache keeps its files, and so does anything a person put there.  --keep-run-dirs leaves everything.
        import shutil
        for dname in ([] if args.keep_run_dirs else os.listdir(g)):
            d = os.path.join(g, dname)
            if not dname.startswith("run_w") or d == run_dir:
                continue
            try:
                owner = int(open(os.path.join(d, ".bench_run_owner")).read().strip() or "0")
            except (OSError, ValueError):
                continue                                     # not marked: not ours to delete
            alive = False
            if owner > 0 and owner != os.getpid():
                try:
                    os.kill(owner, 0); alive = True
                except ProcessLookupError:
                    alive = False
                except PermissionError:
                    alive = True
            if not alive:
                shutil.rmtree(d, ignore_errors=True)
        os.makedirs(run_dir, exist_ok=True)
        with open(os.path.join(run_dir, ".bench_run_owner"), "w") as f:
            f.write(str(os.getpid()))
    barrier()
    t = time.time()
    fq = make_reads(args, g, run_dir, "reads_r%d" % rank, n_total, 7000 + rank)
    log("rank %d: %d pairs of reads in %.1f s" % (rank, n_total, time.time() - t))
    outp = os.path.join(run_dir, "gpu_r%d_" % rank)
    ncpu_eff = effective_cpus()
    threads = args.host_threads or max(4, min(64, ncpu_eff // world))
    argv = ["--runMode", "alignReads", "--genomeDir", idx, "--readFilesIn"] + fq + ["--outFileNamePrefix", outp, "--runThreadN", str(threads),
            "--gpuBatchReads", str(args.reads), "--benchWarmupReads", str(args.warmup * args.reads), "--readMapNumber", str(n_total)]
    # (the front end runs STARAMD_CONTEXTS_PER_GPU engine contexts per GPU, default 1; rep.nContexts says how many ran)
    if not selftest:
        argv += ["--gpuDevice", str(local_rank)]
    if world > 1:          # the ranks load the 31 GB index into host memory one after the other (each keeps ~5 GB of it after the upload)
        os.environ["STARAMD_INDEX_LOAD_LOCK"] = os.path.join(args.workdir, "index_load.lock")
    t_clock = {}

    def warmup_done():
        barrier()
        t_clock["t0"] = time.perf_counter(); t_clock["thr0"] = cpu_throttle()

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
    rc, rep = _run_cli(argv, warmup_done, exchange)
    if rc != 0:
        raise RuntimeError("the star_amd pipeline failed with exit code %d" % rc)
    elapsed = float(rep.timedWall)
    thr1 = cpu_throttle()
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
    n_ctx = max(1, int(rep.nContexts))
    value = timed_reads_all / elapsed / 1e6
    extra = {"per_kernel_ms_timed_region": ms, "counters_per_pair": {k: (v / n if not isinstance(v, dict) else v) for k, v in c.items() if not k.startswith("_")},
             "algorithmic_bytes_per_pair_whole_path": bytes_per_pair, "index_build": ginfo, "sj_merge_ms": sj_ms.get("ms"),
             "cpu_throttled_in_timed_region": ({"periods": thr1[0] - t_clock["thr0"][0], "ms": (thr1[1] - t_clock["thr0"][1]) / 1e3} if thr1 and t_clock.get("thr0") else None),
             "pipeline": {"timed_wall_s": float(rep.timedWall), "device_s_sum_over_contexts": sum(float(rep.deviceMs[k]) for k in range(n_ctx)) / 1e3,
                          "map_batch_call_s": sum(float(rep.deviceBusy[k]) for k in range(n_ctx)), "engine_contexts_per_gpu": n_ctx // max(1, int(rep.nDevices)),
                          "parse_busy_s": float(rep.parseBusy), "convert_busy_s": float(rep.convertBusy), "postmap_write_busy_s": float(rep.emitBusy),
                          "convert_Mreads_s": n / float(rep.convertBusy) / 1e6 if rep.convertBusy > 0 else None,
                          "parse_Mreads_s": n / float(rep.parseBusy) / 1e6 if rep.parseBusy > 0 else None,
                          "postmap_write_Mreads_s": n / float(rep.emitBusy) / 1e6 if rep.emitBusy > 0 else None,
                          "finish_s": float(rep.finishSeconds), "genome_load_s": float(rep.genomeLoadSeconds), "index_upload_s": float(rep.indexUploadSeconds),
                          "cpu_us_per_pair_by_stage": dict(zip(["input_line_table", "text_to_numeric", "mapper_threads", "postmap_format", "file_writes", "other"], [round(float(x) * 1e6 / n, 4) for x in list(rep.cpuSeconds)[:6]])),
                          "fast_path_batches": dict(zip(["output_through_file_mapping", "input_from_file_mapping", "upload_prefetched", "kernels_begun_beside_result_copy"], [int(x) for x in list(rep.fastPaths)[:4]])),
                          "postmap_whole_run_s": dict(zip(["wait_for_writer", "format_on_threads", "serial_tail", "writer_thread_busy"], [float(x) for x in rep.emitParts]))}}
    if dist is not None:
        dist.destroy_process_group()
    # ---- kernel times for the roofline: ONE engine context (a launch has the GPU to itself); the timed region overlaps the launches of two contexts, so
    # the HIP events around a launch there also cover the time slices of the other context's kernels
    kms, kms_src = ms, "timed region (%d engine context(s))" % n_ctx
    if world == 1 and not args.no_exclusive and n_ctx > max(1, int(rep.nDevices)):
        try:
            ex = exclusive_leg(args, idx, fq, run_dir, threads)
            extra["kernel_ms_exclusive"] = ex
            kms, kms_src = ex["per_kernel_ms"], "one-context leg of the same run (%d batches of %d pairs; the timed region overlaps two contexts)" % (4, args.reads)
            log("kernel_ms_exclusive done")
        except Exception as e:
            extra["kernel_ms_exclusive"] = {"error": repr(e)[:300]}
    dom = max(("k_seed_search", "k_windows", "k_stitch_win"), key=lambda k: kms[k])
    dms, dbytes = kms[dom], kern[dom][1]
    achieved = dbytes / (dms * 1e-3) / 1e9 if dms > 0 else 0.0
    traffic = issue = None
    per_kernel = {}
    try:        # HBM traffic / issue utilisation from rocprofv3 --pmc passes (profiles/README.md): only when taken on THIS kernel source and workload size
        tj = json.load(open(os.path.join(ROOT, "profiles", PMC_TRAFFIC_FILE)))
        ceiling = tj.get("gather_ceiling_Gsectors_s")
        for k in ("k_seed_search", "k_windows", "k_stitch_win"):
            e = {"ms": round(kms[k], 2), "alg_MB": round(kern[k][1] / 1e6, 1)}
            if tj.get("genome_mb") == mb and tj.get("reads_per_launch") == args.reads and k in tj and tj[k].get("kernel_src_sha") == kernel_src_sha(k):
                t = tj[k]
                hb = t.get("hbm_bytes_per_launch")
                if hb and kms[k] > 0:
                    # sectors = counter bytes / 64: what HBM moves for gathers of 1-8 bytes; the rate against the measured dependent-gather ceiling of the session
                    e.update({"traffic_MB": round(hb / 1e6, 1), "traffic_over_alg": round(hb / max(kern[k][1], 1.0), 2), "Msectors": round(hb / 64e6, 1),
                              "Gsectors_s": round(hb / 64.0 / (kms[k] * 1e-3) / 1e9, 2), "of_gather_ceiling": (round(hb / 64.0 / (kms[k] * 1e-3) / 1e9 / ceiling, 3) if ceiling else None)})
                for kk in ("valu_busy_frac", "valu_busy_frac_simd32", "wave_instructions_per_read", "waves_per_simd_resident", "scratch_bytes_per_lane", "vgprs"):
                    if t.get(kk) is not None:
                        e[kk] = round(t[kk], 3) if isinstance(t[kk], float) else t[kk]
                if k == dom:
                    traffic = hb; issue = t.get("valu_busy_frac")
            per_kernel[k] = e
    except Exception:
        pass
    line = {
        "metric": "million reads aligned/sec (whole node), 2x101 bp PE human-scale index, FASTQ in -> SAM out",
        "value": round(value, 4), "unit": "Mreads/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        # what the kernels alone allow (pairs of a step / HIP-event device time of a step): the distance to `value` is the host pipeline + copies
        "device_only_value": (round(args.reads * world / (kms["device_total"] * 1e-3) / 1e6, 4) if kms.get("device_total", 0) > 0 else None),
        "dtype": "u8/u64 integer", "data": "synthetic",
        "config": {"workload": "BASELINE config 2 stand-in: synthetic %d Mb genome, %d annotated junctions, %.1f GB index in HBM, %d distinct pairs 2x%d per GPU as %d+%d batches of %d; "
                               "FASTQ text in -> SAM + SJ.out.tab out, index load excluded"
                               % (mb, ginfo.get("junctions_in_index", 0), ginfo.get("index_bytes", 0) / 1e9, n_total, args.read_len, args.warmup, args.steps, args.reads),
                   "reads_per_gpu_per_step": args.reads, "genome_mb": mb, "host_threads_per_rank": threads, "cpus_online": os.cpu_count(), "cpus_usable": ncpu_eff, "engine_contexts_per_gpu": n_ctx // max(1, int(rep.nDevices)),
                   "parallelism": "reads sharded over %d GPU(s), one process per GPU, full index replica each" % world},
        # "bound": the roofline the path is priced against (integer gathers: HBM, no MFMA anywhere).  What the counters say limits the kernels is in "limited_by": none of them is
        # anywhere near the HBM roof; they issue 0.2 - 0.3 wave-instructions per SIMD cycle out of chains of dependent gathers / LDS / scalar operations at 4 - 7 wavefronts per SIMD
        "roofline": {"bound": "hbm", "limited_by": "instruction issue and dependent-access latency at 4-7 wavefronts per SIMD, not HBM bandwidth (per_kernel: valu_busy_frac, traffic_MB, of_gather_ceiling)", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
                     "traffic": traffic, "issue": issue, "kernel_ms": round(dms, 3), "algorithmic_bytes_per_launch": int(dbytes),
                     "per_kernel_ms": {k: round(kms[k], 2) for k in ("k_seed_search", "k_windows", "k_stitch_win", "device_total")},
                     "per_kernel": per_kernel,
                     "kernel_ms_source": kms_src, "algorithmic_bytes_per_pair_whole_path": round(bytes_per_pair, 1),
                     "whole_path_frac": round(bytes_per_pair * n / max(int(rep.batches), 1) / (kms["device_total"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if kms["device_total"] > 0 else None,
                     "frac_with_round3_byte_count": (round(c["_bytes_stitch_round3_definition"] / (dms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if dom == "k_stitch_win" and dms > 0 else None)},
    }
    if selftest:
        line["selftest"] = "plumbing test on CPU (oracle behind the front end, gloo): NOT a measurement"
    ref_prefix = os.path.join(run_dir, "cpu_")
    if not args.no_cpu_baseline and world == 1:
        try:
            cb = cpu_baseline(idx, fq, ref_prefix, n_total, log)
            extra["cpu_baseline"] = cb
            line["cpu_baseline"] = {k: cb[k] for k in ("value", "unit", "cores", "kind", "all_threads", "all_threads_value", "sample")}
        except Exception as e:
            line["cpu_baseline"] = {"error": repr(e)[:300]}
        log("cpu_baseline done")
        try:
            fp = full_size_parity(ref_prefix, outp)
            line["full_size_parity"] = fp
            log("full_size_parity done")
        except Exception as e:
            line["full_size_parity"] = {"error": repr(e)[:300]}
    line["loaded_libs"] = loaded_libs()
    # ---- everything else: a child process writes it to bench_extra.json (a GPU fault in an optional leg must not cost the line)
    extra_path = os.path.join(args.workdir, "bench_extra.json")
    extra["notes"] = notes; extra["line"] = line
    json.dump(extra, open(extra_path, "w"))
    if world == 1 and not args.no_extra_legs:
        left = args.budget_s - (time.time() - T_START)
        if left > 60:
            try:
                subprocess.run([sys.executable, os.path.abspath(__file__), "--extra-legs-child", json.dumps({"argv": sys.argv[1:], "g": g, "idx": idx, "fq": fq, "run_dir": run_dir, "threads": threads,
                                                                                                              "extra_path": extra_path, "t_start": T_START, "main_value": value, "main_ms": kms})],
                               timeout=left + 120, check=False)
            except Exception as e:
                log("extra legs: " + repr(e)[:200])
    line["extra"] = extra_path
    try:
        keep = os.path.join(ROOT, "gpurun_out")
        if os.path.isdir(keep):
            import shutil
            shutil.copy(extra_path, os.path.join(keep, "bench_extra.json"))
    except Exception:
        pass
    line["bench_wall_s"] = round(time.time() - T_START, 1)
    s = json.dumps(line)
    if len(s) > 3900:           # the driver reads the tail of stdout: the line must stay under 4 KB whatever a leg put into it
        for k in ("full_size_parity", "loaded_libs"):
            if len(s) > 3900 and isinstance(line.get(k), (dict, list)):
                line[k] = str(line[k])[:200]; s = json.dumps(line)
        if len(s) > 3900:
            line["config"]["workload"] = line["config"]["workload"][:300]; line.get("cpu_baseline", {}).pop("sample", None); s = json.dumps(line)
    print(s, flush=True)


PMC_TRAFFIC_FILE = "r06_pmc_hbm_traffic.json"


def _cli_leg(argv, lread, env=None):
    """one run of the product pipeline with extra environment; returns (rep, summary dict)"""
    old = {}
    for k, v in (env or {}).items():
        old[k] = os.environ.get(k); os.environ[k] = v
    try:
        rc, rep = _run_cli(argv)
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
                 "engine_contexts": int(rep.nContexts), "counters_per_pair": {k: v / n for k, v in c.items() if not isinstance(v, dict) and not k.startswith("_")}}


def exclusive_leg(args, idx, fq, run_dir, threads):
    """Per-kernel times with ONE engine context (no second batch sharing the GPU): what a launch costs when it has the device to itself.  The
    per_kernel_ms of the timed region are taken where the launches of two contexts overlap."""
    nb, w = 4, 1
    argv = ["--runMode", "alignReads", "--genomeDir", idx, "--readFilesIn"] + fq + ["--outFileNamePrefix", os.path.join(run_dir, "excl_"), "--runThreadN", str(threads),
            "--gpuBatchReads", str(args.reads), "--benchWarmupReads", str(w * args.reads), "--readMapNumber", str(min((nb + w), args.steps + args.warmup) * args.reads)]
    rep, d = _cli_leg(argv, 2 * args.read_len + 1, {"STARAMD_CONTEXTS_PER_GPU": "1"})
    return d


def extra_legs_child(spec):
    """the optional legs, in a process of their own; results merged into bench_extra.json after every leg"""
    global T_START
    sys.argv = [sys.argv[0]] + spec["argv"]
    args = parse()
    T_START = spec["t_start"]
    g, idx, fq, run_dir, threads, path = spec["g"], spec["idx"], spec["fq"], spec["run_dir"], spec["threads"], spec["extra_path"]
    extra = json.load(open(path))
    notes = extra.setdefault("notes", [])

    def log(s):
        notes.append("[%.0f s] %s" % (time.time() - T_START, s))
        print("bench: " + notes[-1], file=sys.stderr, flush=True)

    legs = [("host_budget", lambda: host_budget_leg(args, idx, fq, run_dir)),
            ("config5_default_flags", lambda: config5_leg(args, g, idx, log, chim_detection=False)),
            ("config5_chimeric_detection", lambda: config5_leg(args, g, idx, log, chim_detection=True)),
            ("config1", lambda: config1_leg(args, log)),
            ("bam_output", lambda: bam_leg(args, idx, fq, run_dir, threads))]
    if not args.no_two_pass:
        legs.append(("two_pass_end_to_end", lambda: two_pass(args, idx, fq, run_dir, threads)))
        legs.append(("two_pass_parity_400mb", lambda: two_pass_parity(args, log)))
        legs.append(("two_pass_parity_full_index", lambda: two_pass_parity(args, log, full=(g, idx, fq, run_dir))))
    if not args.no_sweep:
        legs.append(("index_size_sweep", lambda: sweep(args, args.genome_mb, spec["main_value"], spec["main_ms"], log)))
    for name, fn in legs:
        if time.time() - T_START > args.budget_s:
            extra[name] = {"skipped": "time budget"}
        else:
            try:
                extra[name] = fn()
            except Exception as e:
                extra[name] = {"error": repr(e)[:400]}
            log(name + " done")
        json.dump(extra, open(path, "w"))


def host_budget_leg(args, idx, fq, run_dir):
    """The host at the budget of an 8-GPU node: the same pipeline with cores/8 host threads for this GPU."""
    th = max(2, effective_cpus() // 8)
    nb, w = 8, 2
    argv = ["--runMode", "alignReads", "--genomeDir", idx, "--readFilesIn"] + fq + ["--outFileNamePrefix", os.path.join(run_dir, "hb_"), "--runThreadN", str(th),
            "--gpuBatchReads", str(args.reads), "--benchWarmupReads", str(w * args.reads), "--readMapNumber", str(min(nb + w, args.steps + args.warmup) * args.reads)]
    # the share of ONE rank of eight, enforced: the leg's threads are confined to `th` CPUs (the front end derives every helper thread count -- read slices, writer
    # copy threads -- from --runThreadN, and whatever it starts beyond that shares these CPUs)
    old_aff = None
    try:
        old_aff = os.sched_getaffinity(0)
        os.sched_setaffinity(0, set(sorted(old_aff)[:th]))
    except Exception:
        old_aff = None
    try:
        rep, d = _cli_leg(argv, 2 * args.read_len + 1)
    finally:
        if old_aff is not None:
            os.sched_setaffinity(0, old_aff)
    d["host_threads"] = th
    d["cpus_the_leg_was_confined_to"] = th if old_aff is not None else None
    d["what"] = "one GPU, --runThreadN = cores / 8 on that many CPUs (affinity): the host share of one rank on an 8-GPU node"
    return d


def bam_leg(args, idx, fq, run_dir, threads):
    """BAM output (ReadAlign_alignBAM.cpp, BAMoutput.cpp, bamSortByCoordinate.cpp): the headline workload's first batches written as BAM instead of SAM text,
    unsorted and sorted by coordinate.  Throughput only (parity of both forms: tests/test_bam.py -m gpu); the host compresses (bgzf / zlib level 1 as the reference) on its threads."""
    out = {}
    nb, w = 6, 2
    for name, typ in (("unsorted", ["BAM", "Unsorted"]), ("sorted_by_coordinate", ["BAM", "SortedByCoordinate"])):
        prefix = os.path.join(run_dir, "bam_%s_" % name)
        argv = ["--runMode", "alignReads", "--genomeDir", idx, "--readFilesIn"] + fq + ["--outFileNamePrefix", prefix, "--runThreadN", str(threads), "--gpuBatchReads", str(args.reads),
                "--benchWarmupReads", str(w * args.reads), "--readMapNumber", str(min(nb + w, args.steps + args.warmup) * args.reads), "--outSAMtype"] + typ
        t = time.perf_counter()
        rc, rep = _run_cli(argv)
        wall = time.perf_counter() - t
        if rc:
            out[name] = {"error": "exit code %d" % rc}
            continue
        f = prefix + ("Aligned.out.bam" if name == "unsorted" else "Aligned.sortedByCoord.out.bam")
        out[name] = {"Mreads_s_timed_region": int(rep.timedReads) / max(float(rep.timedWall), 1e-9) / 1e6, "Mreads_s_whole_run_incl_sort_and_index_load": int(rep.reads) / wall / 1e6,
                     "timed_reads": int(rep.timedReads), "bam_bytes": os.path.getsize(f) if os.path.isfile(f) else None, "host_threads": threads,
                     "device_Mreads_s": int(rep.timedReads) / max(sum(float(x) for x in rep.deviceMs) / 1e3, 1e-9) / 1e6}
        for q in (f,):
            try:
                os.remove(q)
            except OSError:
                pass
    out["what"] = "--outSAMtype BAM Unsorted / SortedByCoordinate on the first batches of the headline workload; timed region = after the warm-up batches up to the last output byte, as the main line (the final merge of the sorted form included)"
    return out


def config5_leg(args, g, idx, log, chim_detection):
    """2x150, 1 % errors, 5 % chimeric pairs.  chim_detection False: default flags (chimeric detection off,
    /root/reference/source/parametersDefault:690): window pruning on, 301-base reads, 4 starts per mate -- multi-window stitching, extendAlign soft clips and
    the "too short" path.  True: --chimSegmentMin 12, where chimeric detection wants EVERY transcript of every window (no window
    pruning: stitchWindowAligns.cpp:245-247; the partner is chosen on the device, resultSelect 2, and what comes back is what multMapSelect can pick + the partner).  Same index (sjdbOverhang 100).  Parity against the reference on the first 200 k pairs: SAM multiset,
    SJ.out.tab, Log counters (+ Chimeric.out.junction)."""
    from oracle import refstar
    L = 150; nb, w = 4, 1
    n_total = (nb + w) * args.reads
    rd = os.path.join(g, "chim_n%d" % n_total)
    fq = make_reads(args, g, rd, "chim", n_total, 8100, read_len=L, chim_rate=0.05)
    flags = ["--chimSegmentMin", "12", "--chimOutType", "Junctions"] if chim_detection else []
    tag = "cd_" if chim_detection else "df_"
    argv = ["--runMode", "alignReads", "--genomeDir", idx, "--readFilesIn"] + fq + ["--outFileNamePrefix", os.path.join(rd, tag + "gpu_"), "--runThreadN", str(max(4, min(64, effective_cpus()))),
            "--gpuBatchReads", str(args.reads), "--benchWarmupReads", str(w * args.reads), "--readMapNumber", str(n_total)] + flags
    rep, d = _cli_leg(argv, 2 * L + 1)
    d["workload"] = "%d pairs 2x%d, 1%% substitutions, 5%% chimeric pairs, %s" % (n_total, L, "--chimSegmentMin 12 (every window stitched and recorded, no window pruning; the partner loop of chimericDetectionOld on the device: staramd_params::resultSelect 2)" if chim_detection else "default flags (chimeric detection off)")
    ns = min(200000, n_total)
    p_new, p_ref = os.path.join(rd, tag + "gpuS_"), os.path.join(rd, tag + "ref_")
    rc, _ = _run_cli(["--runMode", "alignReads", "--genomeDir", idx, "--readFilesIn"] + fq + ["--outFileNamePrefix", p_new, "--runThreadN", "32", "--gpuBatchReads", str(args.reads), "--readMapNumber", str(ns)] + flags)
    if rc:
        raise RuntimeError("sample run: exit code %d" % rc)
    t = time.perf_counter()
    refstar.align(idx, fq, p_ref, threads=min(64, os.cpu_count() or 8), extra=["--readMapNumber", str(ns)] + flags, timeout=900)
    d["reference_sample_s"] = time.perf_counter() - t
    par = full_size_parity(p_ref, p_new) or {}
    if chim_detection:
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
    argv = ["--runMode", "alignReads", "--genomeDir", idx, "--readFilesIn"] + fq + ["--outFileNamePrefix", os.path.join(rd, "gpu_"), "--runThreadN", str(max(4, min(64, effective_cpus()))),
            "--gpuBatchReads", str(args.reads), "--benchWarmupReads", str(w * args.reads), "--readMapNumber", str(n_total)]
    rep, d = _cli_leg(argv, 50)
    d["workload"] = "%d single-end reads 1x50, synthetic %d Mb genome (yeast size), SAindex %d bases" % (n_total, mb, ginfo.get("SAindexNbases", 0))
    ns = 100000
    p_new, p_ref = os.path.join(rd, "gpuS_"), os.path.join(rd, "ref_")
    rc, _ = _run_cli(["--runMode", "alignReads", "--genomeDir", idx, "--readFilesIn"] + fq + ["--outFileNamePrefix", p_new, "--runThreadN", "16", "--gpuBatchReads", str(args.reads), "--readMapNumber", str(ns)])
    if rc:
        raise RuntimeError("sample run: exit code %d" % rc)
    t = time.perf_counter(); refstar.align(idx, fq, p_ref, threads=1, extra=["--readMapNumber", str(ns)], timeout=900); t_full = time.perf_counter() - t
    t = time.perf_counter(); refstar.align(idx, fq, p_ref + "l_", threads=1, extra=["--readMapNumber", "1"], timeout=900); t_load = time.perf_counter() - t
    d["reference_1_thread_Mreads_s"] = ns / max(t_full - t_load, 1e-3) / 1e6
    d["parity_vs_reference"] = full_size_parity(p_ref, p_new); d["parity_sample_reads"] = ns
    return d


def sweep(args, main_mb, main_value, main_ms, log):
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
                    "--runThreadN", str(max(4, min(64, effective_cpus()))), "--gpuBatchReads", str(args.reads), "--benchWarmupReads", str(w * args.reads), "--readMapNumber", str(n_total)]
            rc, rep = _run_cli(argv)
            if rc:
                rows.append({"genome_mb": mb, "error": "exit code %d" % rc}); continue
            c, ms, kern, bpp = report_dict(rep, lread)
            rows.append({"genome_mb": mb, "Mreads_s": int(rep.timedReads) / float(rep.timedWall) / 1e6, "per_kernel_ms": ms, "index_generate_s": ginfo.get("index_generate_s"),
                         "junctions_in_index": ginfo.get("junctions_in_index"), "algorithmic_bytes_per_pair": bpp})
        except Exception as e:
            rows.append({"genome_mb": mb, "error": repr(e)[:300]})
    rows.append({"genome_mb": main_mb, "Mreads_s": main_value, "per_kernel_ms": main_ms})
    return rows


def two_pass(args, idx, fq, run_dir, threads):
    """--twopassMode Basic on the first 10 batches of the workload (4 M pairs at the default size): 1st pass on the GPU without SAM,
    junction insertion into the resident index, 2nd pass."""
    n = min(10, args.steps + args.warmup) * args.reads
    argv = ["--runMode", "alignReads", "--genomeDir", idx, "--readFilesIn"] + fq + ["--outFileNamePrefix", os.path.join(run_dir, "cli2p_"), "--runThreadN", str(threads),
            "--gpuBatchReads", str(args.reads), "--twopassMode", "Basic", "--readMapNumber", str(n)]
    rc, rep = _run_cli(argv)
    if rc:
        return {"error": "exit code %d" % rc}
    sjdb = sum(1 for _ in open(os.path.join(run_dir, "cli2p__STARgenome", "sjdbList.out.tab")))
    return {"value": n / float(rep.wallMapping) / 1e6, "unit": "Mreads/s (each read counted once, both passes + insertion in the wall time)", "reads": n,
            "wall_s": float(rep.wallMapping), "pass1_plus_insertion_plus_reupload_s": float(rep.pass1Seconds), "junctions_in_index_after_pass1": sjdb,
            "host_threads": threads, "what": "star_amd --twopassMode Basic end to end (index load excluded)"}


def two_pass_parity(args, log, full=None):
    """SURVEY.md 8d config 4 PINNED at scale: --twopassMode Basic, star_amd against the reference's own 2-pass run (twoPassRunPass1.cpp:17-49, sjdbInsertJunctions.cpp:11-102,
    sjdbBuildIndex.cpp:141-284): the SAM records as a multiset, SJ.out.tab (column 6 = 1 for the junctions the 1st pass inserted), the inserted junction list of the 2nd-pass
    index and the Log.final.out counters.  Two sizes: the 400 Mb index of the sweep with one batch of reads, and (full = the main workload) the index of the headline run --
    3.1 Gb, 6.3e9 suffixes -- with the first three batches (1.2 M pairs) of its FASTQ."""
    from oracle import refstar
    more = []
    if full:
        g, idx_dir, fq, rd = full
        ginfo = json.load(open(os.path.join(g, "build.json"))) if os.path.isfile(os.path.join(g, "build.json")) else {}
        mb = args.genome_mb
        n = min(3, args.steps + args.warmup) * args.reads
        more = ["--readMapNumber", str(n)]
    else:
        mb = 400
        g, ginfo = build_genome(args, mb, log)
        idx_dir = os.path.join(g, "idx")
        n = args.reads
        rd = os.path.join(g, "twopass_n%d" % n)
        fq = make_reads(args, g, rd, "reads", n, 9400)
    th = max(4, min(64, effective_cpus()))
    new, ref = os.path.join(rd, "gpu2p_"), os.path.join(rd, "ref2p_")
    t = time.perf_counter()
    rc, rep = _run_cli(["--runMode", "alignReads", "--genomeDir", idx_dir, "--readFilesIn"] + fq + ["--outFileNamePrefix", new, "--runThreadN", str(th),
                        "--gpuBatchReads", str(args.reads), "--twopassMode", "Basic"] + more)
    t_new = time.perf_counter() - t
    if rc:
        return {"error": "star_amd exit code %d" % rc}
    t = time.perf_counter()
    refstar.align(idx_dir, fq, ref, threads=th, extra=["--twopassMode", "Basic"] + more, timeout=1200)
    t_ref = time.perf_counter() - t
    fp = full_size_parity(ref, new) or {}
    la, lb = (open(p + "_STARgenome/sjdbList.out.tab", "rb").read() for p in (ref, new))
    fp.update({"genome_mb": ginfo.get("genome_mb", mb), "pairs": n, "inserted_junction_list_identical": la == lb, "junctions_inserted_by_pass1": la.count(b"\n"),
               "star_amd_wall_s_with_index_load": round(t_new, 1), "reference_wall_s_with_index_load": round(t_ref, 1), "reference_threads": th,
               "star_amd_pass1_plus_insertion_s": float(rep.pass1Seconds)})
    return fp


if __name__ == "__main__":
    if sys.argv[1:2] == ["--make-reads"]:
        _make_reads_child(*json.loads(sys.argv[2]))
    elif sys.argv[1:2] == ["--sam-digest"]:
        print("%d %d" % _sam_digest_child(sys.argv[2]))
    elif sys.argv[1:2] == ["--extra-legs-child"]:
        extra_legs_child(json.loads(sys.argv[2]))
    else:
        main()
