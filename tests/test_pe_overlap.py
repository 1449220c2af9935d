"""--peOverlapNbasesMin / --peOverlapMMp: pairs whose mates overlap are merged into one read, mapped again as a second batch through the same
engine, and the merged alignments, cut back into the two mates and re-scored, replace the pair's own (ReadAlign_peOverlapMergeMap.cpp).
Merging and conversion are host code (reads.cpp MergedBatch::build, postmap.cpp mergedReadToPair).  The reference with the same flags is
the truth: SAM records, junctions, counters, and the chimeric junction table when the multimapping chimeric detection runs on the merged read."""
import os

import pytest

from util import compare_outputs, oracle_lib, prepare, refstar, run_with_engine

pytestmark = pytest.mark.skipif(not refstar.have_ref(), reason="oracle/_ref/STAR not built (no /root/reference here)")

CASES = [("pe76_overlap", ["--peOverlapNbasesMin", "10", "--peOverlapMMp", "0.1"]),
         ("pe76_overlap", ["--peOverlapNbasesMin", "5", "--peOverlapMMp", "0.01", "--outSJfilterReads", "Unique"]),
         ("pe76_overlap", ["--peOverlapNbasesMin", "12", "--peOverlapMMp", "0.1", "--twopassMode", "Basic", "--outFilterType", "BySJout", "--runThreadN", "3"]),
         ("pe101", ["--peOverlapNbasesMin", "10", "--peOverlapMMp", "0.2"]),
         ("pe150_indel", ["--peOverlapNbasesMin", "20", "--peOverlapMMp", "0.05", "--alignEndsProtrude", "10", "ConcordantPair"])]
CHIM = [("pe76_overlap", ["--peOverlapNbasesMin", "10", "--peOverlapMMp", "0.1", "--chimSegmentMin", "10", "--chimJunctionOverhangMin", "10", "--chimMultimapNmax", "20",
                          "--chimMultimapScoreRange", "3", "--chimNonchimScoreDropMin", "10", "--chimOutJunctionFormat", "1"]),
        ("pe150_chim", ["--peOverlapNbasesMin", "12", "--peOverlapMMp", "0.1", "--chimSegmentMin", "12", "--chimJunctionOverhangMin", "8", "--chimMultimapNmax", "20",
                        "--chimMultimapScoreRange", "3", "--chimScoreJunctionNonGTAG", "-4", "--chimNonchimScoreDropMin", "10", "--alignSJstitchMismatchNmax", "5", "-1", "5", "5",
                        "--alignInsertionFlush", "Right", "--alignSplicedMateMapLminOverLmate", "0", "--alignSplicedMateMapLmin", "30"])]      # STAR-Fusion's option set


def _run(name, more, tmp_path, chim):
    info = dict(prepare(name, str(tmp_path), need_ref=False))
    d = os.path.dirname(info["fastq"][0])
    info["extra"] = list(info["extra"]) + more
    rf = list(info["extra"])
    if "--runThreadN" in rf:
        k = rf.index("--runThreadN"); del rf[k:k + 2]
    ref = refstar.align(info["idx"], info["fastq"], os.path.join(d, "refO_"), threads=1, extra=rf)
    new = run_with_engine(info, os.path.join(d, "newO_"), lambda g, p: oracle_lib.Oracle(g, p), batch_reads=600)
    problems = compare_outputs(ref, new)
    if chim:
        lines = lambda p: [l for l in open(p + "Chimeric.out.junction") if not l.startswith("# 2.7.11b")]
        a, b = lines(ref), lines(new)
        if a != b:
            problems.append("Chimeric.out.junction differs: %d vs %d lines" % (len(a), len(b)))
        print("chimeric lines %d, merged %d" % (len(a), sum(1 for l in a[1:] if len(l.split("\t")) > 19 and l.rstrip("\n").split("\t")[19] == "1")))
    assert not problems, problems


@pytest.mark.parametrize("name,more", CASES)
def test_pe_overlap(name, more, tmp_path, built):
    _run(name, more, tmp_path, False)


@pytest.mark.parametrize("name,more", CHIM)
def test_pe_overlap_with_multimapping_chimeric_detection(name, more, tmp_path, built):
    _run(name, more, tmp_path, True)


def _chimeric_fragments(info, d, n=1500, L=100, seed=3):
    """pairs from short chimeric FRAGMENTS: two loci joined at a random point, fragment shorter than two reads, so the mates overlap and
    the junction sits in the overlap or in one mate -- what mate merging is for"""
    import random
    from util import _read_fasta, _rc
    rng = random.Random(seed)
    chrs = _read_fasta(info["fasta"])
    def piece(n):
        c = rng.choice(chrs); p = rng.randrange(1000, len(c) - 1000); s = c[p:p + n]
        return s if rng.random() < 0.5 else _rc(s)
    out = [open(os.path.join(d, "cf_%d.fq" % m), "w") for m in (1, 2)]
    for i in range(n):
        F = rng.randrange(L + 10, 2 * L - 10)
        cut = rng.randrange(25, F - 25)
        frag = piece(cut) + piece(F - cut)
        if i % 4 == 0:
            frag = piece(F)                       # plain short fragments among them
        m1, m2 = frag[:L], _rc(frag)[:L]
        for o, sq in zip(out, (m1, m2)):
            o.write("@cf%06d\n%s\n+\n%s\n" % (i, sq, "I" * len(sq)))
    for o in out:
        o.close()
    return [o.name for o in out]


def test_pe_overlap_chimeric_stress(tmp_path, built):
    """short chimeric fragments: the multimapping detection runs on the merged reads (PEmerged_bool = 1)"""
    info = dict(prepare("pe101", str(tmp_path), need_ref=False))
    d = os.path.dirname(info["fastq"][0])
    info["fastq"] = _chimeric_fragments(info, d)
    flags = ["--peOverlapNbasesMin", "10", "--peOverlapMMp", "0.1", "--chimSegmentMin", "12", "--chimJunctionOverhangMin", "10", "--chimMultimapNmax", "10", "--chimMultimapScoreRange", "2",
             "--chimNonchimScoreDropMin", "15", "--chimScoreDropMax", "80", "--chimSegmentReadGapMax", "5", "--outSAMunmapped", "Within"]
    ref = refstar.align(info["idx"], info["fastq"], os.path.join(d, "refO_"), threads=1, extra=flags)
    info["extra"] = flags
    new = run_with_engine(info, os.path.join(d, "newO_"), lambda g, p: oracle_lib.Oracle(g, p), batch_reads=700)
    assert not compare_outputs(ref, new)
    a, b = open(ref + "Chimeric.out.junction").readlines(), open(new + "Chimeric.out.junction").readlines()
    merged = sum(1 for l in a[1:] if l.rstrip("\n").split("\t")[19] == "1")
    print("chimeric lines %d, merged %d" % (len(a), merged))
    assert a == b
    assert merged > 100


@pytest.mark.parametrize("clip", ["HardClip", "SoftClip", "Old"])
def test_pe_overlap_chimeric_within_bam(clip, tmp_path, built):
    """Arriba's way: chimeras of merged mates written into the BAM; both segments are cut back into the two mates and the shortest
    one-mate part is dropped, so that only one mate is split"""
    from util import bam_parts
    info = dict(prepare("pe101", str(tmp_path), need_ref=False))
    d = os.path.dirname(info["fastq"][0])
    info["fastq"] = _chimeric_fragments(info, d, n=1200, seed=5)
    flags = ["--peOverlapNbasesMin", "10", "--chimSegmentMin", "10", "--chimOutType", "WithinBAM", clip, "Junctions", "--chimJunctionOverhangMin", "10", "--chimScoreMin", "1", "--chimScoreDropMax", "30",
             "--chimScoreJunctionNonGTAG", "0", "--chimScoreSeparation", "1", "--alignSJstitchMismatchNmax", "5", "-1", "5", "5", "--chimSegmentReadGapMax", "3", "--chimMultimapNmax", "50",
             "--outSAMtype", "BAM", "Unsorted", "SortedByCoordinate", "--outSAMunmapped", "Within"]
    if clip == "Old":            # the default detection algorithm on the merged reads (older Arriba recipes): BAM only
        flags = ["--peOverlapNbasesMin", "10", "--chimSegmentMin", "10", "--chimOutType", "WithinBAM", "--chimJunctionOverhangMin", "10", "--chimScoreMin", "1", "--chimScoreDropMax", "30",
                 "--chimScoreJunctionNonGTAG", "0", "--chimScoreSeparation", "1", "--chimSegmentReadGapMax", "3", "--outSAMtype", "BAM", "Unsorted", "SortedByCoordinate", "--outSAMunmapped", "Within"]
    ref = refstar.align(info["idx"], info["fastq"], os.path.join(d, "refW_"), threads=1, extra=flags)
    info["extra"] = flags
    new = run_with_engine(info, os.path.join(d, "newW_"), lambda g, p: oracle_lib.Oracle(g, p), batch_reads=500)
    for f in ("Aligned.out.bam", "Aligned.sortedByCoord.out.bam"):
        (ta, ra, rr), (tb, rb, nr) = bam_parts(ref + f), bam_parts(new + f)
        assert ra == rb and len(rr) == len(nr)
        bad = [k for k in range(len(rr)) if rr[k] != nr[k]]
        assert not bad, (f, len(bad), bad[:3])
    flag = lambda rec: int.from_bytes(rec[18:20], "little")
    print("records %d, supplementary %d" % (len(rr), sum(1 for x in rr if flag(x) & 0x800)))
    assert sum(1 for x in rr if flag(x) & 0x800) > 100
    if clip != "Old":
        assert open(ref + "Chimeric.out.junction").readlines() == open(new + "Chimeric.out.junction").readlines()
    assert refstar.final_log_counters(ref + "Log.final.out") == refstar.final_log_counters(new + "Log.final.out")
