"""Read clipping before mapping (--clip5pNbases, --clip3pNbases, --clip3pAdapterSeq, --clip3pAdapterMMp, --clip3pAfterAdapterNbases,
--clip5pAfterAdapterNbases; Hamming adapter type): ClipMate::clip / localSearch are restated in star_amd/csrc/host/reads.cpp, the
clipped lengths come back as soft clips in the SAM / BAM CIGARs (postmap.cpp).  The engine only ever sees the shorter reads.
The reference run with the same flags on the same FASTQ is the truth; reads get adapters spliced in (with errors, Ns, at the very
start) so that the search, the mismatch budget and 0-length mates after clipping are all exercised."""
import os
import random

import pytest

from util import bam_parts, capi, compare_outputs, oracle_lib, prepare, refstar, run_with_engine

pytestmark = pytest.mark.skipif(not refstar.have_ref(), reason="oracle/_ref/STAR not built (no /root/reference here)")

AD1, AD2 = "AGATCGGAAGAGCACACGTCTGAACTCCAGTCA", "AGATCGGAAGAGCGTCGTGTAGGGAAAGAGTGT"


def _with_adapters(paths, d, adapters, seed=5, zero_len=True):
    """every 3rd read: its tail replaced by the adapter (+ random bases to keep the length), sometimes with a mismatch or an N in
    the adapter; a few reads are adapter from base 0 (0-length after clipping) when zero_len"""
    rng = random.Random(seed)
    out = []
    for im, p in enumerate(paths):
        lines = open(p).read().split("\n")
        if lines[-1] == "":
            lines.pop()
        ad = adapters[im]
        for i in range(0, len(lines) // 4):
            if i % 3:
                continue
            s = lines[4 * i + 1]
            L = len(s)
            keep = 0 if (zero_len and i % 60 == 0) else rng.randrange(L // 4, L)
            a = list(ad)
            k = rng.random()
            if k < 0.3:
                a[rng.randrange(len(a))] = rng.choice("ACGT")
            elif k < 0.4:
                a[rng.randrange(len(a))] = "N"
            tail = "".join(a) + "".join(rng.choice("ACGT") for _ in range(L))
            lines[4 * i + 1] = (s[:keep] + tail)[:L]
        q = os.path.join(d, "adapt_%d.fq" % (im + 1))
        open(q, "w").write("\n".join(lines) + "\n")
        out.append(q)
    return out


PE = {
    "n5n3": ["--clip5pNbases", "3", "7", "--clip3pNbases", "11", "2"],
    "adapter": ["--clip3pAdapterSeq", AD1, AD2, "--clip3pAdapterMMp", "0.1", "0.2"],
    "all": ["--clip5pNbases", "4", "0", "--clip3pNbases", "0", "5", "--clip3pAdapterSeq", AD1, AD2, "--clip3pAdapterMMp", "0.15", "0.05",
            "--clip3pAfterAdapterNbases", "2", "3", "--outSAMattributes", "NH", "HI", "AS", "nM", "NM", "MD", "MC", "jM", "jI"],
    "one_mate": ["--clip3pAdapterSeq", "-", AD2, "--clip3pAdapterMMp", "0.1", "0.1", "--clip5pNbases", "6", "0", "--clip5pAfterAdapterNbases", "2", "9"],
    "polyA": ["--clip3pAdapterSeq", "polyA", "-", "--clip3pAdapterMMp", "0.1", "0.1", "--outSAMunmapped", "Within"],
}
SE = {
    "n5n3": ["--clip5pNbases", "5", "--clip3pNbases", "9"],
    "adapter": ["--clip3pAdapterSeq", AD1, "--outSAMunmapped", "Within", "--clip3pAfterAdapterNbases", "1"],
    "huge": ["--clip3pNbases", "45", "--clip5pNbases", "4", "--outSAMunmapped", "Within"],
}


def _run(info, tag, flags, tmp_path, adapters=None, zero_len=True, **kw):
    d = os.path.dirname(info["fastq"][0])
    if adapters:
        info["fastq"] = _with_adapters(info["fastq"], d, adapters, zero_len=zero_len)
    info["extra"] = list(info["extra"]) + flags
    ref = refstar.align(info["idx"], info["fastq"], os.path.join(d, "ref_%s_" % tag), threads=1, extra=info["extra"])
    new = run_with_engine(info, os.path.join(d, "new_%s_" % tag), lambda g, p: oracle_lib.Oracle(g, p), **kw)
    return ref, new


@pytest.mark.parametrize("tag", sorted(PE))
def test_clipping_paired(tag, tmp_path, built):
    info = dict(prepare("pe101", str(tmp_path), need_ref=False))
    ref, new = _run(info, tag, PE[tag], tmp_path, adapters=(AD1, AD2))
    assert not compare_outputs(ref, new)


@pytest.mark.parametrize("tag", sorted(SE))
def test_clipping_single(tag, tmp_path, built):
    info = dict(prepare("se50", str(tmp_path), need_ref=False))
    ref, new = _run(info, tag, SE[tag], tmp_path, adapters=(AD1,))
    assert not compare_outputs(ref, new)


def test_clipping_in_bam_and_chimeric_output(tmp_path, built):
    """soft clips of the clipped bases in the packed CIGARs; Chimeric.out.junction CIGARp is written against the unclipped lengths"""
    info = dict(prepare("pe150_chim", str(tmp_path), need_ref=False))
    flags = ["--clip5pNbases", "2", "5", "--clip3pNbases", "6", "1", "--outSAMtype", "BAM", "Unsorted", "--outSAMunmapped", "Within",
             "--chimSegmentMin", "15", "--chimJunctionOverhangMin", "15"]
    ref, new = _run(info, "bam", flags, tmp_path)
    (ta, ra, rr), (tb, rb, nr) = bam_parts(ref + "Aligned.out.bam"), bam_parts(new + "Aligned.out.bam")
    assert ra == rb and rr == nr
    chim = lambda p: [l for l in open(p + "Chimeric.out.junction") if not l.startswith("# 2.7.11b")]
    assert chim(ref) == chim(new) and len(chim(ref)) >= 5
    assert open(ref + "SJ.out.tab", "rb").read() == open(new + "SJ.out.tab", "rb").read()


def test_clip_and_block_attributes(tmp_path, built):
    """cN (bases clipped at the two ends) and rB (read and genome coordinates of every block), BAM only"""
    info = dict(prepare("pe101", str(tmp_path), need_ref=False))
    flags = ["--outSAMattributes", "NH", "HI", "rB", "cN", "AS", "--clip3pNbases", "5", "2", "--clip5pNbases", "1", "3", "--outSAMunmapped", "Within", "--outSAMtype", "BAM", "Unsorted"]
    ref, new = _run(info, "cn", flags, tmp_path)
    (ta, ra, rr), (tb, rb, nr) = bam_parts(ref + "Aligned.out.bam"), bam_parts(new + "Aligned.out.bam")
    assert ra == rb and rr == nr and sum(1 for x in rr if b"rBBi" in x) > 1000


def test_clipping_parameter_errors(tmp_path, built):
    info = prepare("pe101", str(tmp_path), need_ref=False)
    base = ["--genomeDir", info["idx"], "--readFilesIn"] + info["fastq"] + ["--outFileNamePrefix", str(tmp_path / "e_")]
    for extra, text in [(["--clip3pNbases", "5"], "--clip3pNbases has to contain 2 values to match the number of mates"),
                        (["--clip3pAdapterSeq", AD1, AD2], "--clip3pAdapterMMp has to contain 2 values"),
                        (["--clip3pAdapterSeq", AD1], "--clip3pAdapterSeq has to contain 2 values"),
                        (["--clip5pAdapterSeq", AD1, AD2], "--clip5pAdapterSeq is not supported yet"),
                        (["--clipAdapterType", "CellRanger4"], "--clipAdapterType")]:
        with pytest.raises(RuntimeError) as e:
            capi.HostRun(base + extra)
        assert text in str(e.value)
