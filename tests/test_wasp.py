"""--varVCFfile (vA / vG attributes: the sample's SNVs under every alignment) and --waspOutputMode SAMtag (vW: the WASP filter -- every other combination of
alleles of a uniquely mapped read is mapped again, as one more batch through the same engine, and must land on the same blocks).  variation.cpp; the reference
run with the same VCF is the truth, BAM records compared byte for byte."""
import os
import random

import pytest

from util import _read_fasta, bam_parts, capi, oracle_lib, prepare, refstar, run_with_engine

pytestmark = pytest.mark.skipif(not refstar.have_ref(), reason="oracle/_ref/STAR not built (no /root/reference here)")


def _vcf(info, d, seed=4):
    rng = random.Random(seed)
    names = [l[1:].split()[0] for l in open(info["fasta"]) if l.startswith(">")]
    seqs = _read_fasta(info["fasta"])
    p = os.path.join(d, "sample.vcf")
    with open(p, "w") as o:
        o.write("##fileformat=VCFv4.2\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tS1\n")
        for nm, sq in zip(names, seqs):
            pos = 50
            dense = rng.randrange(2000, len(sq) - 5000)
            while pos < len(sq) - 50:
                ref = sq[pos - 1].upper()
                if ref in "ACGT":
                    alts = [x for x in "ACGT" if x != ref]
                    k = rng.random()
                    if k < 0.70: o.write("%s\t%d\t.\t%s\t%s\t.\tPASS\t.\tGT\t%s\n" % (nm, pos, ref, rng.choice(alts), rng.choice(["0|1", "1|0", "0/1"])))
                    elif k < 0.78: o.write("%s\t%d\t.\t%s\t%s\t.\tPASS\t.\tGT\t1|1\n" % (nm, pos, ref, rng.choice(alts)))          # homozygous: not used by WASP
                    elif k < 0.86: o.write("%s\t%d\t.\t%s\t%s,%s\t.\tPASS\t.\tGT:DP\t1|2:30\n" % (nm, pos, ref, alts[0], alts[1]))
                    elif k < 0.90: o.write("%s\t%d\t.\t%s\t%sAC\t.\tPASS\t.\tGT\t0|1\n" % (nm, pos, ref, ref))                    # an insertion: skipped
                    elif k < 0.94: o.write("%s\t%d\t.\t%s\t%s\t.\tPASS\t.\tGT\t./.\n" % (nm, pos, ref, alts[0]))
                    elif k < 0.97: o.write("chrUn\t%d\t.\t%s\t%s\t.\tPASS\t.\tGT\t0|1\n" % (pos, ref, alts[0]))
                    else: o.write("%s\t%d\t.\t%s\t%s\t.\tPASS\t.\tGT\t0|0\n" % (nm, pos, ref, alts[0]))
                pos += rng.randrange(2, 7) if dense <= pos < dense + 3000 else rng.randrange(40, 260)
    return p


CASES = [("pe101", ["--waspOutputMode", "SAMtag", "--outSAMattributes", "NH", "HI", "AS", "nM", "vA", "vG", "vW", "--outSAMtype", "BAM", "Unsorted"]),
         ("se50", ["--waspOutputMode", "SAMtag", "--outSAMtype", "BAM", "Unsorted", "SortedByCoordinate", "--outSAMunmapped", "Within", "--runThreadN", "3"]),
         ("pe76_overlap", ["--outSAMattributes", "NH", "HI", "vG", "vA", "--outSAMtype", "BAM", "Unsorted", "--twopassMode", "Basic"]),
         ("pe101_sparse3", ["--waspOutputMode", "SAMtag", "--outSAMattributes", "vA", "vW", "NH", "--outSAMtype", "BAM", "SortedByCoordinate", "--outFilterType", "BySJout", "--quantMode", "TranscriptomeSAM"]),
         ("pe76_overlap", ["--waspOutputMode", "SAMtag", "--outSAMtype", "BAM", "Unsorted", "--quantMode", "TranscriptomeSAM", "GeneCounts", "--outMultimapperOrder", "Random"])]


@pytest.mark.parametrize("name,more", CASES)
def test_variants_and_wasp(name, more, tmp_path, built):
    info = dict(prepare(name, str(tmp_path), need_ref=False))
    d = os.path.dirname(info["fastq"][0])
    info["extra"] = [x for x in info["extra"]]
    if "--outSAMattributes" in more and "--outSAMattributes" in info["extra"]:
        k = info["extra"].index("--outSAMattributes"); j = k + 1
        while j < len(info["extra"]) and not info["extra"][j].startswith("--"):
            j += 1
        del info["extra"][k:j]
    info["extra"] += more + ["--varVCFfile", _vcf(info, d)]
    rf = list(info["extra"])
    if "--runThreadN" in rf:
        k = rf.index("--runThreadN"); del rf[k:k + 2]
    ref = refstar.align(info["idx"], info["fastq"], os.path.join(d, "refV_"), threads=1, extra=rf)
    new = run_with_engine(info, os.path.join(d, "newV_"), lambda g, p: oracle_lib.Oracle(g, p), batch_reads=600)
    seen = {}
    for f in sorted(x for x in os.listdir(d) if x.startswith("refV_") and x.endswith(".bam")):
        (ta, ra, rr), (tb, rb, nr) = bam_parts(os.path.join(d, f)), bam_parts(os.path.join(d, "newV_" + f[5:]))
        assert ra == rb and len(rr) == len(nr)
        bad = [k for k in range(len(rr)) if rr[k] != nr[k]]
        assert not bad, (f, len(bad), rr[bad[0]][-60:], nr[bad[0]][-60:])
        for x in rr:
            k = x.find(b"vWi")
            if k >= 0:
                seen[x[k + 3]] = seen.get(x[k + 3], 0) + 1
    print("vW values:", sorted(seen.items()))
    if "--waspOutputMode" in more:
        assert seen.get(1, 0) > 50 and len(seen) >= 3
    assert refstar.final_log_counters(ref + "Log.final.out") == refstar.final_log_counters(new + "Log.final.out")


def test_wasp_parameter_errors(tmp_path, built):
    info = prepare("se50", str(tmp_path), need_ref=False)
    base = ["--genomeDir", info["idx"], "--readFilesIn"] + info["fastq"] + ["--outFileNamePrefix", str(tmp_path / "e_")]
    for extra, text in [(["--waspOutputMode", "SAMtag"], "--waspOutputMode option requires VCF file"), (["--waspOutputMode", "SAMtag", "--varVCFfile", "x.vcf"], "--waspOutputMode requires output to BAM file"),
                        (["--outSAMattributes", "NH", "vA"], "contains vA and/or vG tag(s), but --varVCFfile is not set"), (["--outSAMattributes", "NH", "vW"], "contains vW tag, but --waspOutputMode is not set"),
                        (["--outSAMattributes", "NH", "vA", "--varVCFfile", "x.vcf"], "contains vA tag, which requires BAM output"),
                        (["--waspOutputMode", "Yes"], "unknown/unimplemented --waspOutputMode option: Yes")]:
        with pytest.raises(RuntimeError) as e:
            capi.HostRun(base + extra)
        assert text in str(e.value)


def test_large_batches_are_mapped_in_pieces(tmp_path, built):
    """capi.map_in_pieces (what Engine.map_batch does with a batch larger than its context, e.g. a WASP re-mapping batch): same result bytes as one call"""
    info = prepare("pe101", str(tmp_path), need_ref=False)
    run = capi.HostRun(["--genomeDir", info["idx"], "--readFilesIn"] + info["fastq"] + ["--outFileNamePrefix", str(tmp_path / "p_")])
    orc = oracle_lib.Oracle(run.genome, run.params)
    try:
        b = run.next_batch(2500)
        whole, parts = capi.ResultBuffers(b.nReads, tr_cap=b.nReads * 64), capi.ResultBuffers(b.nReads, tr_cap=b.nReads * 64)
        orc.map_batch(b, whole)
        capi.map_in_pieces(orc.map_batch, b, parts, 700)
        assert whole.res.trCount == parts.res.trCount and whole.as_bytes(b.nReads) == parts.as_bytes(b.nReads)
    finally:
        orc.close(); run.close()
