"""FASTA reads (> records, the sequence possibly over several lines): no qualities anywhere in the output (* in SAM, 0xFF in BAM, FASTA in
Unmapped.out.mate*), everything else as with FASTQ.  reads.cpp FastqReader::fillFasta; reference: ReadAlignChunk_processChunks.cpp:158-190, readLoad.cpp:84-88."""
import os

import pytest

from util import bam_parts, compare_outputs, oracle_lib, prepare, refstar, run_with_engine

pytestmark = pytest.mark.skipif(not refstar.have_ref(), reason="oracle/_ref/STAR not built (no /root/reference here)")


def _to_fasta(paths, d, wrap):
    out = []
    for im, p in enumerate(paths):
        lines = open(p).read().split("\n")
        q = os.path.join(d, "reads_%d.fa" % (im + 1))
        with open(q, "w") as o:
            for i in range(len(lines) // 4):
                name, s = lines[4 * i][1:], lines[4 * i + 1]
                o.write(">%s some description\n" % name)
                w = wrap if (wrap and i % 3 == 0) else len(s)
                for k in range(0, len(s), w):
                    o.write(s[k:k + w] + ("\r\n" if i % 7 == 0 else "\n"))
        out.append(q)
    return out


@pytest.mark.parametrize("name,more,wrap", [("se50", ["--outSAMunmapped", "Within", "--outReadsUnmapped", "Fastx"], 0),
                                            ("pe101", ["--outSAMunmapped", "Within", "--outReadsUnmapped", "Fastx", "--outFilterType", "BySJout", "--twopassMode", "Basic"], 60),
                                            ("pe76_overlap", ["--outSAMtype", "BAM", "Unsorted", "--outSAMunmapped", "Within", "--runThreadN", "3"], 25)])
def test_fasta_reads(name, more, wrap, tmp_path, built):
    info = dict(prepare(name, str(tmp_path), need_ref=False))
    d = os.path.dirname(info["fastq"][0])
    info["fastq"] = _to_fasta(info["fastq"], d, wrap)
    info["extra"] = [x for x in info["extra"] if x not in ("--outSAMunmapped", "Within")] + more
    rf = list(info["extra"])
    if "--runThreadN" in rf:
        k = rf.index("--runThreadN"); del rf[k:k + 2]
    ref = refstar.align(info["idx"], info["fastq"], os.path.join(d, "refF_"), threads=1, extra=rf)
    new = run_with_engine(info, os.path.join(d, "newF_"), lambda g, p: oracle_lib.Oracle(g, p), batch_reads=400)
    if "BAM" in more:
        (ta, ra, rr), (tb, rb, nr) = bam_parts(ref + "Aligned.out.bam"), bam_parts(new + "Aligned.out.bam")
        assert ra == rb and sorted(rr) == sorted(nr)
        assert open(ref + "SJ.out.tab", "rb").read() == open(new + "SJ.out.tab", "rb").read()
    else:
        assert not compare_outputs(ref, new)
        assert all(l.split(b"\t")[10] == b"*" for l in open(new + "Aligned.out.sam", "rb") if not l.startswith(b"@"))
        for m in range(len(info["fastq"])):
            f = "Unmapped.out.mate%d" % (m + 1)
            assert open(ref + f, "rb").read() == open(new + f, "rb").read(), f
