"""--readFilesType SAM SE | PE: reads from SAM text (e.g. `samtools view` of an unaligned BAM through --readFilesCommand): header lines skipped, mates on consecutive
records in either order, reverse-strand records turned back, and the attributes of the input records written out again with every alignment
(--readFilesSAMattrKeep picks the ones that go into BAM output).  reads.cpp FastqReader::fillSam; reference: ReadAlignChunk_processChunks.cpp:28-107."""
import os

import pytest

from util import _rc, bam_parts, capi, compare_outputs, oracle_lib, prepare, refstar, run_with_engine

pytestmark = pytest.mark.skipif(not refstar.have_ref(), reason="oracle/_ref/STAR not built (no /root/reference here)")


def _to_sam(paths, d):
    mates = []
    for p in paths:
        lines = open(p).read().split("\n")
        mates.append([(lines[4 * i][1:], lines[4 * i + 1], lines[4 * i + 3]) for i in range(len(lines) // 4)])
    q = os.path.join(d, "reads.sam")
    with open(q, "w") as o:
        o.write("@HD\tVN:1.6\tSO:unsorted\n@RG\tID:lane1\tSM:x\n@CO\tunaligned\n")
        for i in range(len(mates[0])):
            recs = []
            for m, mt in enumerate(mates):
                name, s, ql = mt[i]
                flag = 4 if len(mates) == 1 else (77 if m == 0 else 141)
                if i % 5 == 0:                      # stored on the reverse strand
                    flag |= 0x10; s = _rc(s); ql = ql[::-1]
                attrs = ["RG:Z:lane1", "XN:i:%d" % (i * 7 - 100)]
                if i % 3 == 0:
                    attrs += ["XC:A:%s" % "QZ"[m], "XF:f:0.25", "XB:B:c,1,2"]
                if i % 11 == 0:
                    attrs = []
                recs.append("\t".join([name, str(flag), "*", "0", "0", "*", "*", "0", "0", s, ql] + attrs))
            if i % 4 == 1:
                recs.reverse()                      # mate 2 first
            o.write("\n".join(recs) + "\n")
    return [q]


@pytest.mark.parametrize("name,more", [("se50", ["--outSAMunmapped", "Within", "--outReadsUnmapped", "Fastx"]),
                                       ("pe101", ["--outSAMunmapped", "Within", "--outFilterType", "BySJout", "--twopassMode", "Basic", "--outReadsUnmapped", "Fastx"]),
                                       ("pe76_overlap", ["--outSAMtype", "BAM", "Unsorted", "--outSAMunmapped", "Within", "--readFilesSAMattrKeep", "RG", "XC", "XF"]),
                                       ("se50", ["--outSAMtype", "BAM", "Unsorted", "--readFilesCommand", "cat", "--readFilesSAMattrKeep", "None"])])
def test_sam_reads(name, more, tmp_path, built):
    info = dict(prepare(name, str(tmp_path), need_ref=False))
    d = os.path.dirname(info["fastq"][0])
    paired = len(info["fastq"]) == 2
    info["fastq"] = _to_sam(info["fastq"], d)
    info["extra"] = [x for x in info["extra"] if x not in ("--outSAMunmapped", "Within")] + more + ["--readFilesType", "SAM", "PE" if paired else "SE"]
    ref = refstar.align(info["idx"], info["fastq"], os.path.join(d, "refM_"), threads=1, extra=info["extra"])
    new = run_with_engine(info, os.path.join(d, "newM_"), lambda g, p: oracle_lib.Oracle(g, p), batch_reads=350)
    if "BAM" in more:
        (ta, ra, rr), (tb, rb, nr) = bam_parts(ref + "Aligned.out.bam"), bam_parts(new + "Aligned.out.bam")
        assert ra == rb and rr == nr
    else:
        assert not compare_outputs(ref, new)
        for m in range(2 if paired else 1):
            f = "Unmapped.out.mate%d" % (m + 1)
            assert open(ref + f, "rb").read() == open(new + f, "rb").read(), f


def test_sam_reads_errors(tmp_path, built):
    info = prepare("pe101", str(tmp_path), need_ref=False)
    base = ["--genomeDir", info["idx"], "--readFilesIn", info["fastq"][0], "--outFileNamePrefix", str(tmp_path / "e_")]
    for extra, text in [(["--readFilesType", "SAM"], "--readFilesType SAM requires specifying SE or PE reads"), (["--readFilesType", "BAM"], "unknown/unimplemented value for --readFilesType: BAM"),
                        (["--readFilesType", "SAM", "SE", "--readFilesSAMattrKeep", "RGX"], "should contain two letters")]:
        with pytest.raises(RuntimeError) as e:
            capi.HostRun(base + extra)
        assert text in str(e.value)
