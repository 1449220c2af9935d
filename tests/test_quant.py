"""--quantMode GeneCounts (ReadsPerGene.out.tab): host post-map code over the unique alignment of each read, three strandedness
columns.  Byte-identical to the reference, also combined with a GTF given at the mapping stage, 2-pass, BySJout and threads."""
import os

import pytest

from util import bam_parts, capi, compare_outputs, oracle_lib, prepare, refstar, run_with_engine

pytestmark = pytest.mark.skipif(not refstar.have_ref(), reason="oracle/_ref/STAR not built (no /root/reference here)")

CASES = [("pe101", []), ("pe101", ["--twopassMode", "Basic", "--outFilterType", "BySJout"]), ("se50", ["GTF"]), ("pe150_chim", ["--runThreadN", "3"]),
         ("pe76_overlap", [])]


def _case(name, more, tmp_path, factory):
    info = dict(prepare(name, str(tmp_path), need_ref=False))
    d = os.path.dirname(info["fastq"][0])
    m = ["--sjdbGTFfile", info["gtf"]] if more == ["GTF"] else more
    info["extra"] = list(info["extra"]) + ["--quantMode", "GeneCounts"] + m
    rf = list(info["extra"])
    if "--runThreadN" in rf:
        k = rf.index("--runThreadN"); del rf[k:k + 2]
    ref = refstar.align(info["idx"], info["fastq"], os.path.join(d, "refQ_"), threads=1, extra=rf)
    new = run_with_engine(info, os.path.join(d, "newQ_"), factory, batch_reads=900)
    problems = compare_outputs(ref, new)
    if open(ref + "ReadsPerGene.out.tab", "rb").read() != open(new + "ReadsPerGene.out.tab", "rb").read():
        problems.append("ReadsPerGene.out.tab differs")
    assert not problems, problems


@pytest.mark.parametrize("name,more", CASES)
def test_gene_counts_oracle(name, more, tmp_path, built):
    _case(name, more, tmp_path, lambda g, p: oracle_lib.Oracle(g, p))


@pytest.mark.gpu
@pytest.mark.parametrize("name,more", CASES[:2])
def test_gene_counts_engine(name, more, tmp_path, built):
    _case(name, more, tmp_path, lambda g, p: capi.Engine(g, p, device=0, max_reads=4096))


TRSAM = [("pe101", []),
         ("pe101", ["GeneCounts", "--quantTranscriptomeSAMoutput", "BanSingleEnd"]),
         ("se50", ["GTF"]),
         ("pe150_indel", ["--quantTranscriptomeSAMoutput", "BanSingleEnd_ExtendSoftclip", "--outSAMattributes", "NH", "HI", "AS", "MC"]),
         ("pe150_chim", ["--twopassMode", "Basic", "--outFilterType", "BySJout", "--runThreadN", "3"]),
         ("pe76_overlap", ["--outSAMattrRGline", "ID:rg1", "SM:s"])]


@pytest.mark.parametrize("name,more", TRSAM)
def test_transcriptome_sam(name, more, tmp_path, built):
    """--quantMode TranscriptomeSAM: alignments projected onto the annotated transcripts (Aligned.toTranscriptome.out.bam, what RSEM reads).  Indel /
    single-end bans, soft-clip extension, the randomly chosen primary alignment (one draw of the reference's mt19937 stream per mapped read, in
    read order), RG / MC attributes: the decompressed BAM stream is byte-identical to the reference's."""
    info = dict(prepare(name, str(tmp_path), need_ref=False))
    d = os.path.dirname(info["fastq"][0])
    q, m = ["--quantMode", "TranscriptomeSAM"], list(more)
    if m and m[0] == "GeneCounts":
        q.append("GeneCounts"); m = m[1:]
    if m == ["GTF"]:
        m = ["--sjdbGTFfile", info["gtf"]]
    info["extra"] = list(info["extra"]) + q + m
    rf = list(info["extra"])
    if "--runThreadN" in rf:
        k = rf.index("--runThreadN"); del rf[k:k + 2]
    ref = refstar.align(info["idx"], info["fastq"], os.path.join(d, "refT_"), threads=1, extra=rf)
    new = run_with_engine(info, os.path.join(d, "newT_"), lambda g, p: oracle_lib.Oracle(g, p), batch_reads=700)
    assert not compare_outputs(ref, new)
    (ta, ra, rr), (tb, rb, nr) = bam_parts(ref + "Aligned.toTranscriptome.out.bam"), bam_parts(new + "Aligned.toTranscriptome.out.bam")
    assert ta == tb and ra == rb
    assert len(rr) == len(nr) and rr == nr
    if "GeneCounts" in q:
        assert open(ref + "ReadsPerGene.out.tab", "rb").read() == open(new + "ReadsPerGene.out.tab", "rb").read()
