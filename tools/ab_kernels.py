#!/usr/bin/env python3
"""Same-process A/B of engine builds / knobs on the GPU box, cheap in GPU-minutes: the index is generated (or taken from the bench cache) and
loaded ONCE, a few batches are parsed ONCE, and every variant -- an engine library + environment knobs -- gets a fresh engine context over
them (index upload ~1 s at 3.1 Gb), maps every batch `--repeat` times and reports the per-kernel HIP-event times of staramd_get_timings.
A configuration costs ~2 s instead of the ~25 s of a bench.py run at 3100 Mb, and the variants see byte-identical input.

  tools/ab_kernels.py [--genome-mb 3100] [--reads 400000] [--batches 3] [--repeat 2] [--rounds 2] "tag|lib or -|ENV=V ENV2=V" ...

`lib`: path of a libstaramd*.so (tools/build_variants.sh), `-` = star_amd/lib/libstaramd.so.  The variants run in the given order, `--rounds`
times (ABAB...: drift of the box shows up as a difference between the rounds of one variant).  The result buffers of the first variant
are the reference: every other variant must return the same bytes (a variant that changes results is reported, not timed).
One engine context, so per-kernel times are exclusive (bench.py's timed region overlaps two contexts)."""
import argparse
import ctypes as C
import hashlib
import json
import os
import shutil
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
STAGES = ["k_seed_search", "k_windows", "k_order", "k_stitch(lane+win)", "verify+replay+finish", "scan+gather", "device_total"]


def load_engine(path):
    """a private handle of an engine library: a copy under a unique name (dlopen returns the cached handle for a path it has seen)"""
    from star_amd import capi
    tmp = tempfile.mkdtemp(prefix="abk_")
    dst = os.path.join(tmp, os.path.basename(path))
    shutil.copy(path, dst)
    L = C.CDLL(dst)
    L.staramd_create.restype = C.c_int
    L.staramd_create.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.POINTER(capi.Genome), C.POINTER(capi.Params), C.c_uint32, C.c_uint64]
    L.staramd_map_batch.restype = C.c_int
    L.staramd_map_batch.argtypes = [C.c_void_p, C.POINTER(capi.Batch), C.POINTER(capi.Results)]
    L.staramd_destroy.restype = None; L.staramd_destroy.argtypes = [C.c_void_p]
    L.staramd_last_error.restype = C.c_char_p
    L.staramd_get_timings.restype = C.c_int; L.staramd_get_timings.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.c_int]
    L.staramd_get_counters.restype = C.c_int; L.staramd_get_counters.argtypes = [C.c_void_p, capi.u64p, C.c_int]
    return L, tmp


class OwnedBatch:
    """a copy of a parsed batch (the host library reuses its buffers for the next one)"""

    def __init__(self, b):
        from star_amd import capi
        n = b.nReads
        nb = b.readOffset[n]
        self.bases = (C.c_uint8 * nb).from_buffer_copy(C.string_at(b.bases, nb))
        self.off = (C.c_uint64 * (n + 1)).from_buffer_copy(C.string_at(b.readOffset, 8 * (n + 1)))
        self.m1 = (C.c_uint16 * n).from_buffer_copy(C.string_at(b.mate1Length, 2 * n))
        self.mm = (C.c_uint16 * n).from_buffer_copy(C.string_at(b.mmMaxTotal, 2 * n))
        self.b = capi.Batch()
        self.b.nReads = n
        self.b.bases = C.cast(self.bases, capi.u8p); self.b.readOffset = C.cast(self.off, capi.u64p)
        self.b.mate1Length = C.cast(self.m1, capi.u16p); self.b.mmMaxTotal = C.cast(self.mm, capi.u16p)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--genome-mb", type=int, default=3100)
    ap.add_argument("--reads", type=int, default=400000)
    ap.add_argument("--read-len", type=int, default=101)
    ap.add_argument("--batches", type=int, default=3)
    ap.add_argument("--repeat", type=int, default=2, help="times every batch is mapped per variant and round (the first pass of the first batch is a warm-up and not counted)")
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--workdir", default="/dev/shm/star_amd_bench" if os.path.isdir("/dev/shm") else "/tmp/star_amd_bench")
    ap.add_argument("--genome-dir", default=None, help="an existing genomeDir + --fastq instead of the bench workload (tests)")
    ap.add_argument("--fastq", nargs="*", default=None)
    ap.add_argument("--flags", default="", help="extra alignReads flags, space separated")
    ap.add_argument("--out", default=None, help="JSON file for the table")
    ap.add_argument("variants", nargs="+")
    a = ap.parse_args()
    from star_amd import capi
    if a.genome_dir:
        idx, fq = a.genome_dir, a.fastq
    else:
        import bench
        args = argparse.Namespace(read_len=a.read_len, reads=a.reads, workdir=a.workdir)
        g, _ = bench.build_genome(args, a.genome_mb, lambda s: print("ab_kernels:", s, file=sys.stderr))
        idx = os.path.join(g, "idx")
        fq = bench.make_reads(args, g, os.path.join(g, "abk_n%d" % (a.batches * a.reads)), "reads", a.batches * a.reads, 7700)
    t0 = time.time()
    run = capi.HostRun(["--genomeDir", idx, "--readFilesIn"] + list(fq) + ["--outFileNamePrefix", os.path.join(tempfile.mkdtemp(prefix="abk_out_"), "x_"), "--runThreadN", str(min(64, os.cpu_count() or 8))]
                       + (a.flags.split() if a.flags else []))
    batches = []
    while len(batches) < a.batches:
        b = run.next_batch(a.reads)
        if b is None:
            break
        batches.append(OwnedBatch(b))
    print("ab_kernels: index + %d batches on the host in %.1f s" % (len(batches), time.time() - t0), file=sys.stderr)
    nmax = max(x.b.nReads for x in batches)
    table, ref_digest, ref_full = {}, None, None
    saved_env = dict(os.environ)
    for rnd in range(a.rounds):
        for spec in a.variants:
            tag, lib, envs = (spec.split("|") + ["", ""])[:3]
            lib = os.path.join(ROOT, "star_amd", "lib", "libstaramd.so") if lib in ("", "-") else lib
            os.environ.clear(); os.environ.update(saved_env)
            for kv in envs.split():
                k, v = kv.split("=", 1); os.environ[k] = v
            L, tmp = load_engine(lib)
            ctx = C.c_void_p()
            rc = L.staramd_create(C.byref(ctx), 0, run.genome, run.params, nmax, 0)
            if rc:
                print("%-20s create failed: %s" % (tag, L.staramd_last_error().decode())); continue
            bufs = capi.ResultBuffers(nmax, tr_cap=nmax * 24)
            acc = [0.0] * 9; cnt = 0; dig = hashlib.sha256(); dig_full = hashlib.sha256(); ok = True
            for rep in range(a.repeat):
                for ib, ob in enumerate(batches):
                    rc = L.staramd_map_batch(ctx, C.byref(ob.b), C.byref(bufs.res))
                    if rc == -3:                                     # result arrays too small: grow once
                        bufs = capi.ResultBuffers(nmax, tr_cap=int(bufs.res.trCount * 1.3) + 4096, ex_cap=int(bufs.res.exCount * 1.3) + 4096)
                        rc = L.staramd_map_batch(ctx, C.byref(ob.b), C.byref(bufs.res))
                    if rc:
                        print("%-20s map_batch failed: %s" % (tag, L.staramd_last_error().decode())); ok = False; break
                    if rep == 0:
                        rd, tr, ex = bufs.as_bytes(ob.b.nReads)
                        dig_full.update(rd); dig_full.update(tr); dig_full.update(ex)
                        # staramd_read_result::maxScoreMate[] (bytes 20-27 of the 32-byte record) is a lower bound under resultSelect = 1 (include/star_amd.h: it covers the
                        # walked windows / finalised leaves only, and for reads whose windows are separate work items what is walked depends on timing): compared apart
                        import numpy as np
                        arr = np.frombuffer(rd, dtype=np.uint32).reshape(-1, 8).copy(); arr[:, 5:7] = 0
                        dig.update(arr.tobytes()); dig.update(tr); dig.update(ex)
                    if rep == 0 and ib == 0:
                        continue                                    # warm-up
                    ms = (C.c_float * 9)(); L.staramd_get_timings(ctx, ms, 9)
                    for i in range(9):
                        acc[i] += ms[i]
                    cnt += 1
                if not ok:
                    break
            counters = (C.c_uint64 * 64)(); L.staramd_get_counters(ctx, counters, 64)
            L.staramd_destroy(ctx)
            shutil.rmtree(tmp, ignore_errors=True)
            if not ok or cnt == 0:
                continue
            d = dig.hexdigest(); dfull = dig_full.hexdigest()
            if ref_digest is None:
                ref_digest = d; ref_full = dfull
            row = {"round": rnd, "ms": {STAGES[i]: acc[i] / cnt for i in range(7)}, "launches_timed": cnt, "results_equal_first_variant": d == ref_digest, "windows_mid_big_ms": acc[7] / cnt, "stitch_lane_ms": acc[8] / cnt,
                   "lane_fraction": counters[39] / max(1, batches[-1].b.nReads)}
            nlast = max(1, batches[-1].b.nReads)         # (the engine's counters are those of the last launch)
            row["counters_per_pair"] = {k: counters[i] / nlast for i, k in enumerate(["nSAi", "nSAprobe", "nGcmp", "nSAenum", "nGstitch", "nSeeds", "nWindows", "nWA", "nNodes", "nLeaves", "nStitchCalls", "nExtendCalls", "nTrOut"])}
            row["counters_per_pair"].update({"nPrunedWin": counters[37] / nlast, "nRewalkRead": counters[38] / nlast, "nOwnerLookups": counters[40] / nlast, "nOwnerMisses": counters[41] / nlast,
                                             "nAnchorLoci": counters[42] / nlast, "nAnchorReplayed": counters[43] / nlast, "nSkippedLeaves": counters[47] / nlast, "nRewalkWin": counters[48] / nlast,
                                             "nLeavesBound": counters[49] / nlast, "nLeavesEarly": counters[50] / nlast})       # (the last two: profile / shadow builds only)
            if sum(counters[21:37]):                     # a -DSTARAMD_PROFILE build: shader-clock cycles per section, summed over wavefronts
                pn = ["walk(all)", "coopStitch", "coopExtend", "finalize(all)", "recordCandidate", "-", "-", "wave_lifetime", "windows:passA", "windows:flanks", "windows:passB_enumerate+owner",
                      "windows:passB_assign", "windows:emission", "finalize:extends", "finalize:filters+score", "finalize:candidate+log"]
                row["profile_kcycles_per_pair"] = {pn[i]: round(counters[21 + i] / nlast / 1e3, 2) for i in range(16) if pn[i] != "-"}
                row["profile_kcycles_per_pair"]["windows:passB_owner_lookups"] = round(counters[44] / nlast / 1e3, 2)       # (then passB_enumerate+owner is the enumeration alone)
                row["profile_kcycles_per_pair"]["windows:passA_loads"] = round(counters[45] / nlast / 1e3, 2)               # (then passA is the replay alone)
                row["profile_kcycles_per_pair"]["windows:passA_prefilter"] = round(counters[46] / nlast / 1e3, 2)
                row["profile_kcycles_per_pair"]["window_setup"] = round(counters[21 + 5] / nlast / 1e3, 2)
                kinds = ["annotated", "no/equal gap", "deletion/junction", "mate join", "insertion/rejected"]      # dev.h DC_sprof0..: coopStitch by kind of join
                row["stitch_calls_by_kind"] = {kinds[i]: {"calls_per_pair": round(counters[51 + 2 * i + 1] / nlast, 2), "kcycles_per_call": round(counters[51 + 2 * i] / max(1, counters[51 + 2 * i + 1]) / 1e3, 2)} for i in range(5)}
                print("    profile (k cycles per pair):", row["profile_kcycles_per_pair"], flush=True)
                print("    coopStitch by kind:", row["stitch_calls_by_kind"], flush=True)
                pass
            if True:
                print("    counts per pair:", {k: round(v, 2) for k, v in row["counters_per_pair"].items()}, flush=True)
            table.setdefault(tag, []).append(row)
            m = row["ms"]
            print("%-20s r%d  seed %6.2f  windows %6.2f (mid+big %5.2f)  stitch %6.2f (lane %5.2f)  redecide %5.2f  total %7.2f  lane %.3f  ovfWin %.5f  %s" %
                  (tag, rnd, m[STAGES[0]], m[STAGES[1]], acc[7] / cnt, m[STAGES[3]], acc[8] / cnt, m[STAGES[4]], m[STAGES[6]], row["lane_fraction"], counters[13] / max(1, batches[-1].b.nReads), ("" if dfull == ref_full else "(maxScoreMate differs)") if row["results_equal_first_variant"] else "RESULTS DIFFER FROM THE FIRST VARIANT"), flush=True)
    os.environ.clear(); os.environ.update(saved_env)
    run.close()
    if a.out:
        json.dump(table, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
