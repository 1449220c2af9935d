#!/usr/bin/env python3
"""GPU side of the bug hunt (needs an MI355X): random data sets and random HOT-PATH flags, the HIP engine's result buffers against the oracle's, byte for
byte -- every read result, transcript and exon record, in both result-selection modes, including the second batch of merged mates and clipped / 0-length mates.
The oracle side of the same combinations is pinned against the reference by tests/tools/fuzz_flags.py on CPU.
usage (on the GPU box): python tests/tools/fuzz_engine.py [iterations] [seed]         e.g.  gpurun --timeout 900 -- 'python tests/tools/fuzz_engine.py 60 1 > gpurun_out/fuzz_engine.log 2>&1'
On a machine WITHOUT a GPU:  FUZZ_EMUL=1 python tests/tools/fuzz_engine.py [iterations] [seed]  runs the same combinations through the wavefront emulator
(oracle/_build/libstaramd_emul.so: the kernel sources compiled for the host) on the first FUZZ_EMUL_READS (40) reads of every data set, one process per
combination with a time limit (a few reads take minutes in the emulator: such a combination is reported as skipped)."""
import os
import random
import subprocess
import sys
import tempfile

EMUL = os.environ.get("FUZZ_EMUL") == "1"
EMUL_READS = int(os.environ.get("FUZZ_EMUL_READS", "40"))
if EMUL:
    os.environ["STARAMD_ENGINE_LIB"] = os.environ.get("STARAMD_EMUL_LIB") or os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "oracle", "_build", "libstaramd_emul.so")    # STARAMD_EMUL_LIB: a variant build (oracle/wave_emul/build.sh variant ...)

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
from util import capi, oracle_lib, prepare, refstar   # noqa: E402
from star_amd import synth                            # noqa: E402

HOT = [["--alignEndsType", "EndToEnd"], ["--alignEndsType", "Extend5pOfRead1"], ["--alignEndsType", "Extend5pOfReads12"], ["--alignEndsProtrude", "15", "ConcordantPair"],
       ["--alignInsertionFlush", "Right"], ["--alignSoftClipAtReferenceEnds", "No"], ["--alignIntronMax", "300"], ["--alignIntronMax", "5000", "--alignMatesGapMax", "5000"], ["--alignIntronMin", "5"],
       ["--alignIntronMin", "60"], ["--alignSJoverhangMin", "3", "--alignSJDBoverhangMin", "1"], ["--alignSJoverhangMin", "20"], ["--alignSplicedMateMapLmin", "30", "--alignSplicedMateMapLminOverLmate", "0"],
       ["--alignSplicedMateMapLminOverLmate", "0.9"], ["--alignSJstitchMismatchNmax", "0", "0", "0", "0"], ["--alignSJstitchMismatchNmax", "5", "-1", "5", "5"], ["--alignTranscriptsPerReadNmax", "50", "--alignTranscriptsPerWindowNmax", "5"],
       ["--scoreGap", "-2", "--scoreGapNoncan", "-4"], ["--scoreGapATAC", "-2", "--scoreGapGCAG", "-1"], ["--scoreGenomicLengthLog2scale", "0"], ["--scoreGenomicLengthLog2scale", "-1"],
       ["--scoreDelOpen", "-1", "--scoreInsOpen", "-1", "--scoreDelBase", "-1", "--scoreInsBase", "-1"], ["--scoreInsOpen", "-5", "--scoreDelOpen", "0"], ["--scoreStitchSJshift", "0"], ["--sjdbScore", "0"], ["--sjdbScore", "5"],
       ["--seedSearchStartLmax", "12"], ["--seedSearchStartLmax", "80"], ["--seedSearchStartLmaxOverLread", "0.3"], ["--seedSearchLmax", "30"], ["--seedMultimapNmax", "50"], ["--seedMultimapNmax", "300", "--winAnchorMultimapNmax", "200"],
       ["--seedPerWindowNmax", "10"], ["--seedPerReadNmax", "300"], ["--seedSplitMin", "8", "--seedMapMin", "3"], ["--seedMapMin", "10", "--seedSplitMin", "20"], ["--winAnchorMultimapNmax", "20"],
       ["--winBinNbits", "10", "--winAnchorDistNbins", "30"], ["--winBinNbits", "14", "--winAnchorDistNbins", "5"], ["--winBinNbits", "18"], ["--winFlankNbins", "2"],
       ["--outFilterMismatchNmax", "3"], ["--outFilterMismatchNoverLmax", "0.05"], ["--outFilterMismatchNoverLmax", "0.0"], ["--outFilterMismatchNoverReadLmax", "0.04"], ["--outFilterMultimapScoreRange", "4"],
       ["--outFilterIntronMotifs", "RemoveNoncanonical"], ["--outFilterIntronMotifs", "RemoveNoncanonicalUnannotated"], ["--outFilterIntronStrands", "None"], ["--outSAMstrandField", "intronMotif"],
       ["--chimSegmentMin", "12"], ["--peOverlapNbasesMin", "10", "--peOverlapMMp", "0.1"], ["--clip3pNbases", "CLIP"], ["--clip5pNbases", "CLIP"], ["--clip3pAdapterSeq", "ADAPT"]]


def one(it, rng):
    work = tempfile.mkdtemp(prefix="fzeng%04d_" % it)
    if rng.random() < 0.5:
        rl = rng.choice([36, 50, 75, 100, 125, 151, 250])
        pe = rng.random() < 0.6
        kw = dict(seed=rng.randrange(1 << 30), chr_lengths=tuple(rng.randrange(60000, 300000) for _ in range(rng.randrange(1, 5))), n_tr=rng.randrange(20, 120), n_reads=rng.randrange(800, 2500),
                  read_len=rl, paired=pe, sub_rate=rng.choice([0.0, 0.005, 0.02, 0.05]), n_rate=rng.choice([0.0, 0.002, 0.02]), indel_rate=rng.choice([0.0, 0.0, 0.002]),
                  chim_rate=rng.choice([0.0, 0.0, 0.1]), frag=(max(rl // 2, 40), max(rl * 3, 200)))
        name = "rand_%s%d" % ("pe" if pe else "se", rl)
        d = os.path.join(work, name)
        info = synth.make_dataset(d, **kw)
        info["idx"] = os.path.join(d, "idx")
        gtf = rng.random() < 0.7
        refstar.genome_generate(info["fasta"], info["idx"], gtf=info["gtf"] if gtf else None, sa_index_nbases=rng.choice([6, 8, 10]), **(dict(sjdb_overhang=rng.choice([rl - 1, 30, 100])) if gtf else {}),
                                **(dict(extra=("--genomeSAsparseD", str(rng.choice([2, 3])))) if rng.random() < 0.2 else {}))
        info["extra"] = []
    else:
        name = rng.choice(["pe101", "se50", "pe150_indel", "pe76_overlap", "pe150_chim", "pe101_sparse3"])
        info = dict(prepare(name, work, need_ref=False))
    paired = len(info["fastq"]) == 2
    used, flags = set(x for x in info["extra"] if x.startswith("--")), []
    for fl in rng.sample(HOT, rng.randrange(1, 6)):
        names = [x for x in fl if x.startswith("--")]
        if any(n in used for n in names) or (not paired and fl[0] in ("--peOverlapNbasesMin", "--alignEndsProtrude", "--alignSplicedMateMapLminOverLmate") ) or (not paired and "Extend5pOfReads12" in fl):
            continue
        used.update(names)
        fl = [str(rng.choice([3, 20, 400])) if v == "CLIP" else ("AGATCGGAAG" if v == "ADAPT" else v) for v in fl]   # 400: everything clipped, 0-length mates
        if fl[0].startswith("--clip") and paired:
            fl = [fl[0], fl[1], fl[1] if rng.random() < 0.5 else "0"] if fl[1] != "AGATCGGAAG" else [fl[0], fl[1], "-", "--clip3pAdapterMMp", "0.1", "0.1"]
        flags += fl
    flags += ["--gpuResultSelect", rng.choice(["All", "Selected"])]
    if EMUL:
        flags = ["--readMapNumber", str(EMUL_READS)] + flags
    tag = "%s %s" % (name, " ".join(info["extra"] + flags))
    print("run  [%d] %s" % (it, tag), flush=True)
    try:
        run = capi.HostRun(["--genomeDir", info["idx"], "--readFilesIn"] + info["fastq"] + ["--outFileNamePrefix", os.path.join(work, "x_")] + info["extra"] + flags)
    except RuntimeError as e:
        if "PARAMETERS error" in str(e) or "INPUT ERROR" in str(e):           # a combination the parameter checks refuse (as the reference does): nothing to compare
            print("ok   [%d] (refused: %s)" % (it, str(e).splitlines()[0][:120]), flush=True)
            return True
        raise
    eng = capi.Engine(run.genome, run.params, device=0, max_reads=4096)
    orc = oracle_lib.Oracle(run.genome, run.params)
    bad = None
    try:
        while bad is None:
            b = run.next_batch(rng.choice([700, 1500, 4000]))
            if b is None:
                break
            for bt in (b, run.merged_batch()):
                if bt is None:
                    continue
                n = bt.nReads
                cap = n * 400
                while True:            # (EndToEnd + every transcript of 2x250 reads: > 600 transcripts per read; the engine reports the overflow, the caller grows)
                    bg, bo = capi.ResultBuffers(n, tr_cap=cap), capi.ResultBuffers(n, tr_cap=cap)
                    try:
                        eng.map_batch(bt, bg); orc.map_batch(bt, bo)
                        break
                    except RuntimeError as e:
                        if "result arrays too small" not in str(e) or cap > n * 20000:
                            raise
                        cap *= 4
                rg, tg, eg = bg.as_bytes(n); ro, to, eo = bo.as_bytes(n)
                if rg != ro or tg != to or eg != eo or bg.res.trCount != bo.res.trCount:
                    for i in range(n):
                        a, o = bg.reads[i], bo.reads[i]
                        fa = (a.status, a.nW, a.nTr, a.trOffset, a.trBest, a.unmappedLength, a.maxScoreMate[0], a.maxScoreMate[1])
                        fo = (o.status, o.nW, o.nTr, o.trOffset, o.trBest, o.unmappedLength, o.maxScoreMate[0], o.maxScoreMate[1])
                        if fa != fo:                           # (resultSelect 1: maxScoreMate[] is 0 on both sides, include/star_amd.h)
                            bad = "read %d: engine %r oracle %r" % (i, fa, fo); break
                    if bad is None and (tg != to or eg != eo or bg.res.trCount != bo.res.trCount):
                        bad = "transcript / exon records differ"
                    if bad:
                        break
            if bad is None:
                bufs = capi.ResultBuffers(b.nReads, tr_cap=b.nReads * 4); bufs = None
    except Exception as e:
        bad = "exception: %s" % str(e)[:300]
    finally:
        eng.close(); orc.close(); run.close()
    print(("FAIL [%d] %s\n      %s\n      kept in %s" % (it, tag, bad, work)) if bad else ("ok   [%d]" % it), flush=True)
    return bad is None


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "--one":          # one combination of the emulator mode, in a process of its own
    it, seed = int(sys.argv[2]), int(sys.argv[3])
    sys.exit(0 if one(it, random.Random(seed * 100003 + it)) else 1)

if __name__ == "__main__" and EMUL:
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    bad = slow = 0
    for i in range(n):
        try:
            env = dict(os.environ, STARAMD_EMUL_ORDER="desc" if i % 2 else "asc")       # the lanes of a wavefront scheduled in either order: the results must not depend on it
            p = subprocess.run([sys.executable, os.path.abspath(__file__), "--one", str(i), str(seed)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env, timeout=int(os.environ.get("FUZZ_EMUL_TIMEOUT", "240")))
            out = p.stdout.strip().splitlines()
            print("\n".join(l for l in out if l.startswith(("run ", "ok ", "FAIL", "      "))), flush=True)
            if p.returncode != 0:
                bad += 1
                if not any(l.startswith("FAIL") for l in out):
                    print("FAIL [%d] exit code %d\n%s" % (i, p.returncode, "\n".join(out[-8:])), flush=True)
        except subprocess.TimeoutExpired:
            slow += 1; print("skip [%d] over the time limit in the emulator" % i, flush=True)
    print("%d of %d combinations differ (%d skipped as too slow for the emulator)" % (bad, n, slow))
    sys.exit(1 if bad else 0)

if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    bad = sum(0 if one(i, rng) else 1 for i in range(n))
    print("%d of %d combinations differ" % (bad, n))
    sys.exit(1 if bad else 0)
