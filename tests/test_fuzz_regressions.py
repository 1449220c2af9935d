"""Flag combinations that tests/tools/fuzz_flags.py found to differ from the reference, kept as regression tests: unmapped reads in the transcriptome BAM,
random multimapper order with merged mates, order of auto-added RG / XS attributes, read groups of reads held for the 2nd BySJout stage with several input
files, random order + chimeric alignments in the BAM + transcriptome BAM."""
import os

import pytest

from test_output_options import _split
from util import bam_parts, oracle_lib, prepare, refstar, run_with_engine

pytestmark = pytest.mark.skipif(not refstar.have_ref(), reason="oracle/_ref/STAR not built (no /root/reference here)")

CASES = [("pe76_overlap", False, ["--clip3pNbases", "4", "2", "--outFilterMismatchNoverLmax", "0.05", "--quantMode", "TranscriptomeSAM", "GeneCounts", "--quantTranscriptomeSAMoutput", "BanSingleEnd",
                                  "--outSAMtype", "BAM", "SortedByCoordinate"]),
         ("pe150_indel", False, ["--peOverlapNbasesMin", "5", "--outMultimapperOrder", "Random", "--outSAMtype", "BAM", "Unsorted", "SortedByCoordinate"]),
         ("se50", True, ["--outSAMstrandField", "intronMotif", "--outSAMtype", "BAM", "Unsorted", "--outSAMattrRGline", "ID:a", "SM:x", ",", "ID:b"]),
         ("pe150_indel", True, ["--outFilterType", "BySJout", "--outSAMtype", "BAM", "SortedByCoordinate", "--outSAMattrRGline", "ID:a", "SM:x", ",", "ID:b"]),
         ("pe150_chim", False, ["--outMultimapperOrder", "Random", "--chimSegmentMin", "12", "--chimScoreDropMax", "40", "--chimScoreSeparation", "3", "--chimSegmentReadGapMax", "3",
                                "--quantMode", "TranscriptomeSAM", "--chimOutType", "WithinBAM", "Junctions", "--outSAMtype", "BAM", "Unsorted"]),
         ("pe150_chim", False, ["--chimSegmentMin", "12", "--chimJunctionOverhangMin", "10", "--chimOutType", "WithinBAM", "Junctions", "--varVCFfile", "VCF", "--outSAMattributes", "NH", "HI", "AS", "nM", "vA", "vG",
                                "--waspOutputMode", "SAMtag", "--outSAMtype", "BAM", "Unsorted"])]


@pytest.mark.parametrize("name,split,more", CASES)
def test_fuzz_regression(name, split, more, tmp_path, built):
    info = dict(prepare(name, str(tmp_path), need_ref=False))
    d = os.path.dirname(info["fastq"][0])
    if split:
        info["fastq"] = _split(info["fastq"], d)
    if "VCF" in more:            # chimeric records carry the variants of the alignments before the junction shift, and the WASP verdict of the read before
        from test_wasp import _vcf
        more = [_vcf(info, d) if x == "VCF" else x for x in more]
        info["extra"] = [x for x in info["extra"] if False]
    info["extra"] = list(info["extra"]) + more
    ref = refstar.align(info["idx"], info["fastq"], os.path.join(d, "ref_"), threads=1, extra=info["extra"])
    new = run_with_engine(info, os.path.join(d, "new_"), lambda g, p: oracle_lib.Oracle(g, p), batch_reads=777)
    n = 0
    for f in sorted(os.listdir(d)):
        if f.startswith("ref_") and f.endswith(".bam"):
            assert bam_parts(os.path.join(d, f))[1:] == bam_parts(os.path.join(d, "new_" + f[4:]))[1:], f
            n += 1
    assert n >= 1
    assert open(ref + "SJ.out.tab", "rb").read() == open(new + "SJ.out.tab", "rb").read()
    assert refstar.final_log_counters(ref + "Log.final.out") == refstar.final_log_counters(new + "Log.final.out")
