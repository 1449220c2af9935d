"""Output-side flags of alignReads that pipelines rely on: several input files per mate (comma-separated --readFilesIn) with one read group per
file (--outSAMattrRGline ... , ...), RG attribute and @RG header lines, --outReadsUnmapped Fastx, --outSAMreadID Number, --outSAMmultNmax,
--outSAMtlen 2 (BAM only, as in the reference).  The reference run with the same flags and files is the truth."""
import os

import pytest

from util import bam_parts, compare_outputs, oracle_lib, prepare, refstar, run_with_engine

pytestmark = pytest.mark.skipif(not refstar.have_ref(), reason="oracle/_ref/STAR not built (no /root/reference here)")

CASES = {
    "rg2": ["--outSAMattrRGline", "ID:a", "SM:x", ",", "ID:b", "SM:y", "PL:z z", "--outSAMunmapped", "Within"],
    "rg1": ["--outSAMattrRGline", "ID:only", "CN:c", "--outSAMattributes", "NH", "HI", "RG", "AS"],
    "unm": ["--outReadsUnmapped", "Fastx", "--outSAMreadID", "Number", "--outFilterScoreMinOverLread", "0.9", "--outFilterMatchNminOverLread", "0.9"],
    "mult": ["--outSAMmultNmax", "1", "--outSAMtlen", "2"],
    "bam": ["--outSAMtype", "BAM", "Unsorted", "SortedByCoordinate", "--outSAMtlen", "2", "--outSAMattrRGline", "ID:q", "--outSAMunmapped", "Within", "--outSAMmultNmax", "2"],
}


def _split(paths, d):
    """each mate's FASTQ cut into two files of unequal size (the batch size of the test does not divide either)"""
    out = [[] for _ in paths]
    for im, p in enumerate(paths):
        lines = open(p).read().split("\n")
        if lines[-1] == "":
            lines.pop()
        n = len(lines) // 4
        cut = [0, n // 3, n]
        for j in range(2):
            q = os.path.join(d, "part%d_%d.fq" % (j, im + 1))
            open(q, "w").write("\n".join(lines[4 * cut[j]:4 * cut[j + 1]]) + "\n")
            out[im].append(q)
    return [",".join(x) for x in out]


@pytest.mark.parametrize("name", ["pe101", "se50"])
@pytest.mark.parametrize("tag", sorted(CASES))
def test_output_options(name, tag, tmp_path, built):
    info = dict(prepare(name, str(tmp_path), need_ref=False))
    d = os.path.dirname(info["fastq"][0])
    info["fastq"] = _split(info["fastq"], d)
    info["extra"] = list(info["extra"]) + CASES[tag]
    ref = refstar.align(info["idx"], info["fastq"], os.path.join(d, "ref_"), threads=1, extra=info["extra"])
    new = run_with_engine(info, os.path.join(d, "new_"), lambda g, p: oracle_lib.Oracle(g, p), batch_reads=450)
    rg = lambda t: [l for l in t.split(b"\n") if l.startswith(b"@RG")]
    if tag == "bam":
        for f in ("Aligned.out.bam", "Aligned.sortedByCoord.out.bam"):
            (ta, ra, rr), (tb, rb, nr) = bam_parts(ref + f), bam_parts(new + f)
            assert ra == rb and rr == nr, f
            assert rg(ta) == rg(tb)
    else:
        assert not compare_outputs(ref, new)
        assert rg(open(ref + "Aligned.out.sam", "rb").read()) == rg(open(new + "Aligned.out.sam", "rb").read())
    if tag == "unm":
        for m in range(len(info["fastq"])):
            f = "Unmapped.out.mate%d" % (m + 1)
            assert open(ref + f, "rb").read() == open(new + f, "rb").read(), f
