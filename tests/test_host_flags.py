"""Run-level flags of alignReads that live entirely on the host side (star_amd/csrc/host/params.cpp, reads.cpp, postmap.cpp, runner.cpp):
--parametersFiles and --name=value, --readFilesPrefix, --readFilesManifest, --outSAMheaderHD / PG / CommentFile, --outSJtype None,
--outQSconversionAdd, --outMultimapperOrder Random (+ --runRNGseed), --outSAMunmapped Within KeepPairs, --outStd, --runDirPerm and the
resource knobs of the reference that have nothing to size here.  The reference run with the same flags is the truth; with one thread
its output order is deterministic, so where the order is the point the SAM bodies are compared line by line, not sorted."""
import os
import stat
import subprocess
import sys

import pytest

from util import bam_parts, capi, compare_outputs, oracle_lib, prepare, refstar, run_with_engine

pytestmark = pytest.mark.skipif(not refstar.have_ref(), reason="oracle/_ref/STAR not built (no /root/reference here)")

_oracle = lambda g, p: oracle_lib.Oracle(g, p)


def _body(path):
    return [l for l in open(path, "rb") if not l.startswith(b"@")]


def _header(path, keep_cl=False):
    """header lines; the two that quote the command line (paths, binary name) are dropped"""
    h = [l for l in open(path, "rb") if l.startswith(b"@")]
    return [l for l in h if keep_cl or not (l.startswith(b"@PG\tID:STAR") or l.startswith(b"@CO\tuser command line"))]


def _both(info, tag, flags, **kw):
    d = os.path.dirname(info["fastq"][0].split(",")[0]) or info.get("outdir", "")
    assert d, "outputs must not land in the current directory"
    info = dict(info)
    info["extra"] = list(info["extra"]) + flags
    ref = refstar.align(info["idx"], info["fastq"], os.path.join(d, "ref_%s_" % tag), threads=1, extra=info["extra"])
    new = run_with_engine(info, os.path.join(d, "new_%s_" % tag), _oracle, **kw)
    return ref, new


def test_parameters_file_and_equals_syntax(tmp_path, built):
    """values come from the file, the command line wins where both define a name; --name=value is one definition"""
    info = dict(prepare("pe101", str(tmp_path), need_ref=False))
    pf = str(tmp_path / "pars.txt")
    open(pf, "w").write("# comment\n// another\n\noutSAMattributes NH HI AS nM NM MD\noutSAMunmapped   Within\noutFilterMultimapNmax 3\n"
                        "outSAMattrRGline ID:x \"DS:two words\"\nscoreGap -1\n")
    ref, new = _both(info, "pf", ["--parametersFiles", pf, "--outFilterMultimapNmax=7", "--scoreGap=0"])
    assert not compare_outputs(ref, new)
    assert _header(ref + "Aligned.out.sam") == _header(new + "Aligned.out.sam")
    base = ["--genomeDir", info["idx"], "--readFilesIn"] + info["fastq"] + ["--outFileNamePrefix", str(tmp_path / "e_")]
    bad = str(tmp_path / "bad.txt")
    for text, msg in [("scoreGap 0\nscoreGap 1\n", "duplicate parameter \"scoreGap\" in input \"%s\"" % bad), ("scoreGap\n", "empty value for parameter \"scoreGap\""),
                      ("outFileNamePrefix x_\n", "cannot be defined at the input level")]:
        open(bad, "w").write(text)
        with pytest.raises(RuntimeError) as e:
            capi.HostRun(base + ["--parametersFiles", bad])
        assert msg in str(e.value)
    with pytest.raises(RuntimeError) as e:
        capi.HostRun(base + ["--parametersFiles", str(tmp_path / "missing.txt")])
    assert "could not open user-defined parameters file" in str(e.value)


def _split(paths, d):
    out = [[] for _ in paths]
    for im, p in enumerate(paths):
        lines = open(p).read().split("\n")
        if lines[-1] == "":
            lines.pop()
        n = len(lines) // 4
        cut = [0, n // 4, n]
        for j in range(2):
            q = os.path.join(d, "part%d_%d.fq" % (j, im + 1))
            open(q, "w").write("\n".join(lines[4 * cut[j]:4 * cut[j + 1]]) + "\n")
            out[im].append(q)
    return out


@pytest.mark.parametrize("name", ["pe101", "se50"])
def test_manifest_and_prefix(name, tmp_path, built):
    """--readFilesManifest: file names and read groups from a table (ID: added where missing), --readFilesPrefix in front of every name"""
    info = dict(prepare(name, str(tmp_path), need_ref=False))
    d = os.path.dirname(info["fastq"][0])
    parts = _split(info["fastq"], d)
    man = os.path.join(d, "manifest.tsv")
    with open(man, "w") as o:
        for j, rg in enumerate(["ID:first\tSM:a b", "second"]):
            m2 = os.path.basename(parts[1][j]) if len(parts) == 2 else "-"
            o.write("%s\t%s\t%s\n" % (os.path.basename(parts[0][j]), m2, rg))
        o.write("   \n")
    ref, new = _both(info, "man", ["--readFilesManifest", man, "--readFilesPrefix", d + "/", "--outSAMunmapped", "Within"], batch_reads=450)
    assert not compare_outputs(ref, new)
    assert _header(ref + "Aligned.out.sam") == _header(new + "Aligned.out.sam")
    # the same through --readFilesIn with a trailing comma
    info["fastq"] = [",".join(os.path.basename(x) for x in p) + "," for p in parts]
    info["outdir"] = d
    ref, new = _both(info, "pre", ["--readFilesPrefix", d + "/", "--outSAMattrRGline", "ID:first", "SM:a b", ",", "ID:second"])
    assert not compare_outputs(ref, new)


def test_sam_header_options(tmp_path, built):
    info = dict(prepare("se50", str(tmp_path), need_ref=False))
    co = str(tmp_path / "co.txt")
    open(co, "w").write("@CO\tLIBID:xyz\n\n   \n@CO\tanother line\n")
    flags = ["--outSAMheaderHD", "@HD", "VN:1.4", "SO:unsorted", "--outSAMheaderPG", "@PG", "ID:upstream", "PN:tool x", "--outSAMheaderCommentFile", co]
    ref, new = _both(info, "hd", flags)
    assert _header(ref + "Aligned.out.sam") == _header(new + "Aligned.out.sam")
    assert not compare_outputs(ref, new)
    ref, new = _both(info, "hdb", flags + ["--outSAMtype", "BAM", "Unsorted", "SortedByCoordinate"])
    for f in ("Aligned.out.bam", "Aligned.sortedByCoord.out.bam"):
        (ta, ra, rr), (tb, rb, nr) = bam_parts(ref + f), bam_parts(new + f)
        strip = lambda t: [l for l in t.split(b"\n") if not (l.startswith(b"@PG\tID:STAR") or l.startswith(b"@CO\tuser command line"))]
        assert strip(ta) == strip(tb) and ra == rb and rr == nr, f


def test_no_junction_output_and_quality_conversion(tmp_path, built):
    info = dict(prepare("pe101", str(tmp_path), need_ref=False))
    ref, new = _both(info, "sjn", ["--outSJtype", "None", "--outQSconversionAdd", "-20", "--outReadsUnmapped", "Fastx", "--outSAMunmapped", "Within"])
    assert not os.path.exists(ref + "SJ.out.tab") and not os.path.exists(new + "SJ.out.tab")
    assert refstar.sam_body_sorted(ref + "Aligned.out.sam") == refstar.sam_body_sorted(new + "Aligned.out.sam")
    assert refstar.final_log_counters(ref + "Log.final.out") == refstar.final_log_counters(new + "Log.final.out")
    for m in (1, 2):
        assert open(ref + "Unmapped.out.mate%d" % m, "rb").read() == open(new + "Unmapped.out.mate%d" % m, "rb").read()
    # held reads of BySJout carry the converted qualities into the 2nd stage, where they are converted again (as in the reference)
    ref, new = _both(info, "qs2", ["--outQSconversionAdd", "31", "--outFilterType", "BySJout"])
    assert not compare_outputs(ref, new)
    with pytest.raises(RuntimeError) as e:
        capi.HostRun(["--genomeDir", info["idx"], "--readFilesIn"] + info["fastq"] + ["--outFileNamePrefix", str(tmp_path / "e_"), "--outSJtype", "None", "--outFilterType", "BySJout"])
    assert "--outFilterType BySJout requires --outSJtype Standard" in str(e.value)


RANDOM = {
    "plain": ["--outMultimapperOrder", "Random", "--outFilterMultimapScoreRange", "3"],
    "seed": ["--outMultimapperOrder", "Random", "--runRNGseed", "12345", "--outSAMmultNmax", "2", "--outFilterMultimapNmax", "30", "--outFilterMultimapScoreRange", "4",
             "--outSAMattributes", "NH", "HI", "AS", "nM", "NM"],
    "allbest": ["--outMultimapperOrder", "Random", "--outSAMprimaryFlag", "AllBestScore", "--outFilterMultimapScoreRange", "3"],
    "twopass": ["--outMultimapperOrder", "Random", "--twopassMode", "Basic", "--outFilterMultimapScoreRange", "2"],
    "bysjout": ["--outMultimapperOrder", "Random", "--outFilterType", "BySJout", "--outSAMmultNmax", "1", "--outFilterMultimapScoreRange", "3"],
}


def _multicopy_build(d, paired, seed):
    """a genome where every chromosome has two diverged copies (0.7 % and 1.5 % substitutions): most reads are multimappers whose
    alignments tie or differ by a few points"""
    import numpy as np
    from star_amd import synth
    info = synth.make_dataset(d, seed=seed, chr_lengths=(120000, 90000), n_tr=60, n_reads=2000, read_len=101 if paired else 75, paired=paired, sub_rate=0.004)
    rng = np.random.default_rng(seed)
    names, seqs, cur = [], [], None
    for l in open(info["fasta"]):
        if l.startswith(">"):
            names.append(l[1:].strip()); seqs.append([])
        else:
            seqs[-1].append(l.strip())
    seqs = ["".join(x) for x in seqs]
    with open(info["fasta"], "a") as o:
        for k, rate in enumerate((0.007, 0.015)):
            for nm, sq in zip(names, seqs):
                a = np.frombuffer(sq.encode(), dtype=np.uint8).copy()
                hit = np.nonzero(rng.random(a.size) < rate)[0]
                a[hit] = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, hit.size)]
                o.write(">%s_copy%d\n" % (nm, k + 1))
                t = a.tobytes().decode()
                for i in range(0, len(t), 70):
                    o.write(t[i:i + 70] + "\n")
    info["idx"] = os.path.join(d, "idx")
    refstar.genome_generate(info["fasta"], info["idx"], gtf=info["gtf"], sa_index_nbases=8, sjdb_overhang=100)
    info["extra"] = []
    return info


_MULTICOPY = {}


def _multicopy(tmp_path, paired=True, seed=21):
    """built once per (paired, seed) and test process, linked into the test's own directory"""
    import atexit
    import shutil
    import tempfile
    key = (paired, seed)
    if key not in _MULTICOPY:
        root = tempfile.mkdtemp(prefix="staramd_multicopy_")
        atexit.register(shutil.rmtree, root, True)
        _MULTICOPY[key] = _multicopy_build(os.path.join(root, "multicopy"), paired, seed)
    src = _MULTICOPY[key]
    d = str(tmp_path / "multicopy")
    os.makedirs(d, exist_ok=True)
    info = dict(src)
    link = lambda p: (os.path.lexists(os.path.join(d, os.path.basename(p))) or os.symlink(p, os.path.join(d, os.path.basename(p)))) and None or os.path.join(d, os.path.basename(p))
    info["fasta"], info["gtf"], info["idx"] = link(src["fasta"]), link(src["gtf"]), link(src["idx"])
    info["fastq"] = [link(f) for f in src["fastq"]]
    info["extra"] = []
    return info


@pytest.mark.parametrize("paired", [True, False])
@pytest.mark.parametrize("tag", sorted(RANDOM))
def test_random_multimapper_order(paired, tag, tmp_path, built):
    """the two Fisher-Yates shuffles per multimapper, drawn from mt19937(runRNGseed) in read order: same order of records, HI, primary flags"""
    info = _multicopy(tmp_path, paired)
    ref, new = _both(info, "rnd", RANDOM[tag], batch_reads=333)
    a, b = _body(ref + "Aligned.out.sam"), _body(new + "Aligned.out.sam")
    assert sum(1 for l in a if b"NH:i:1\t" not in l) > 1000
    if tag == "bysjout":        # held reads come after all others, there and here; within the two groups the order is the input order
        a, b = sorted(a), sorted(b)
    assert a == b
    assert not compare_outputs(ref, new)


def test_random_order_in_bam(tmp_path, built):
    info = _multicopy(tmp_path)
    ref, new = _both(info, "rndb", ["--outMultimapperOrder", "Random", "--outSAMtype", "BAM", "Unsorted", "SortedByCoordinate", "--quantMode", "GeneCounts", "--outFilterMultimapScoreRange", "3"])
    for f in ("Aligned.out.bam", "Aligned.sortedByCoord.out.bam"):
        (ta, ra, rr), (tb, rb, nr) = bam_parts(ref + f), bam_parts(new + f)
        assert ra == rb and rr == nr, f
    assert open(ref + "ReadsPerGene.out.tab", "rb").read() == open(new + "ReadsPerGene.out.tab", "rb").read()


@pytest.mark.parametrize("paired,more", [(True, []), (False, ["--outFilterType", "BySJout"]), (True, ["--twopassMode", "Basic", "--runThreadN", "3", "--outSAMmultNmax", "2"])])
def test_random_order_with_transcriptome_bam(paired, more, tmp_path, built):
    """Random order + --quantMode TranscriptomeSAM: the shuffles of a read and its draw of the primary transcriptomic alignment alternate in
    one random stream; Aligned.toTranscriptome.out.bam and the SAM must still equal the reference's 1-thread run"""
    info = _multicopy(tmp_path, paired)
    d = os.path.dirname(info["fastq"][0])
    info["extra"] = ["--outMultimapperOrder", "Random", "--quantMode", "TranscriptomeSAM", "GeneCounts", "--outFilterMultimapScoreRange", "3"] + more
    rf = list(info["extra"])
    if "--runThreadN" in rf:
        k = rf.index("--runThreadN"); del rf[k:k + 2]
    ref = refstar.align(info["idx"], info["fastq"], os.path.join(d, "refRT_"), threads=1, extra=rf)
    new = run_with_engine(info, os.path.join(d, "newRT_"), _oracle, batch_reads=1500)
    a, b = _body(ref + "Aligned.out.sam"), _body(new + "Aligned.out.sam")
    if "BySJout" in more:
        a, b = sorted(a), sorted(b)
    assert a == b
    (ta, ra, rr), (tb, rb, nr) = bam_parts(ref + "Aligned.toTranscriptome.out.bam"), bam_parts(new + "Aligned.toTranscriptome.out.bam")
    assert ta == tb and ra == rb and rr == nr and len(rr) > 500
    assert open(ref + "ReadsPerGene.out.tab", "rb").read() == open(new + "ReadsPerGene.out.tab", "rb").read()


@pytest.mark.parametrize("mode", ["sam", "bam_unsorted", "bam_sorted", "bam_both", "sam_random"])
def test_keep_pairs(mode, tmp_path, built):
    """--outSAMunmapped Within KeepPairs: the unmapped mate follows every one-mate alignment of a multimapper (secondary where the alignment is);
    the sorted BAM keeps one unmapped record per read"""
    info = _multicopy(tmp_path, seed=22)
    d = os.path.dirname(info["fastq"][0])
    # make one-mate alignments common: the 2nd mate of every 3rd pair becomes junk
    lines = open(info["fastq"][1]).read().split("\n")
    for i in range(0, len(lines) // 4, 3):
        L = len(lines[4 * i + 1])
        lines[4 * i + 1] = ("ACGTTGCATGCCGATATCGGCTAGCTAGGATCCGATTTAGGCTCTAGAGCTCGATCGGGATATCCGCGATATTAGCAGCTACGACTAGCATCGACTAGC" * 3)[i % 7:i % 7 + L]
    junk = os.path.join(d, "junk_2.fq")
    open(junk, "w").write("\n".join(lines))
    info["fastq"] = [info["fastq"][0], junk]
    info["extra"] = []
    flags = ["--outSAMunmapped", "Within", "KeepPairs", "--outFilterMultimapNmax", "50", "--outFilterMultimapScoreRange", "4",
             "--outFilterScoreMinOverLread", "0.3", "--outFilterMatchNminOverLread", "0.3"]
    if mode == "sam" or mode == "sam_random":
        if mode == "sam_random":
            flags += ["--outMultimapperOrder", "Random", "--outSAMmultNmax", "3"]
        ref, new = _both(info, "kp", flags)
        a, b = _body(ref + "Aligned.out.sam"), _body(new + "Aligned.out.sam")
        assert a == b
        flag = lambda l: int(l.split(b"\t")[1])
        assert sum(1 for l in a if flag(l) & 0x4 and flag(l) & 0x100) > 0
    else:
        kinds = {"bam_unsorted": ["Unsorted"], "bam_sorted": ["SortedByCoordinate"], "bam_both": ["Unsorted", "SortedByCoordinate"]}[mode]
        ref, new = _both(info, "kpb", flags + ["--outSAMtype", "BAM"] + kinds)
        for kind in kinds:
            f = "Aligned.out.bam" if kind == "Unsorted" else "Aligned.sortedByCoord.out.bam"
            if mode == "bam_both" and kind == "SortedByCoordinate":
                # with both files the reference overwrites the record buffers of a one-mate alignment with its unmapped mate before the sorted copy is taken
                # (ReadAlign_outputAlignments.cpp:185-199), so its sorted BAM loses those alignments; ours is the sorted file of the sorted-only run
                ref = refstar.align(info["idx"], info["fastq"], ref + "only_", threads=1, extra=list(info["extra"]) + flags + ["--outSAMtype", "BAM", "SortedByCoordinate"])
            (ta, ra, rr), (tb, rb, nr) = bam_parts(ref + f), bam_parts(new + f)
            assert ra == rb and rr == nr, f


def test_alignments_to_stdout(tmp_path, built):
    """--outStd SAM / BAM_Unsorted / BAM_SortedByCoordinate: the alignments go to stdout instead of the file, nothing else does"""
    info = dict(prepare("se50", str(tmp_path), need_ref=False))
    d = os.path.dirname(info["fastq"][0])
    script = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\nfrom util import run_with_engine, oracle_lib\n"
              "run_with_engine(dict(idx=%r, fastq=%r, extra=sys.argv[2:]), sys.argv[1], lambda g, p: oracle_lib.Oracle(g, p))\n"
              % (os.path.dirname(os.path.abspath(__file__)), os.path.dirname(os.path.dirname(os.path.abspath(__file__))), info["idx"], info["fastq"]))
    for k, (std, more, fname) in enumerate([("SAM", [], "Aligned.out.sam"), ("BAM_Unsorted", ["--outSAMtype", "BAM", "Unsorted"], "Aligned.out.bam"),
                                            ("BAM_SortedByCoordinate", ["--outSAMtype", "BAM", "Unsorted", "SortedByCoordinate"], "Aligned.sortedByCoord.out.bam")]):
        flags = ["--outStd", std] + more
        refp, newp = os.path.join(d, "refstd%d_" % k), os.path.join(d, "newstd%d_" % k)
        with open(refp + "stdout", "wb") as o:
            subprocess.check_call([refstar.REF_BIN, "--runMode", "alignReads", "--genomeDir", info["idx"], "--readFilesIn"] + info["fastq"] +
                                  ["--outFileNamePrefix", refp] + flags, stdout=o)
        with open(newp + "stdout", "wb") as o:
            subprocess.check_call([sys.executable, "-c", script, newp] + flags, stdout=o)
        assert not os.path.exists(refp + fname) and not os.path.exists(newp + fname)
        if std == "SAM":
            assert sorted(_body(refp + "stdout")) == sorted(_body(newp + "stdout"))
            assert _header(refp + "stdout") == _header(newp + "stdout")
        else:
            (ta, ra, rr), (tb, rb, nr) = bam_parts(refp + "stdout"), bam_parts(newp + "stdout")
            assert ra == rb and rr == nr
            if std == "BAM_SortedByCoordinate":     # the other BAM still goes to its file
                assert bam_parts(refp + "Aligned.out.bam")[1:] == bam_parts(newp + "Aligned.out.bam")[1:]


def test_accepted_knobs_and_directory_permissions(tmp_path, built):
    """limits / sorting / tmp-dir knobs of the reference are accepted (nothing is sized by them here); --runDirPerm sets the mode of the run directories"""
    info = dict(prepare("se50", str(tmp_path), need_ref=False))
    flags = ["--limitBAMsortRAM", "1000000000", "--limitIObufferSize", "30000000", "50000000", "--limitOutSJcollapsed", "500000", "--limitOutSJoneRead", "500",
             "--limitOutSAMoneReadBytes", "50000", "--outBAMsortingThreadN", "2", "--outBAMsortingBinsN", "20", "--outSAMorder", "PairedKeepInputOrder",
             "--readMatesLengthsIn", "Equal", "--readQualityScoreBase", "33", "--runDirPerm", "All_RWX", "--twopassMode", "Basic", "--outMultimapperOrder", "Old_2.4"]
    ref, new = _both(info, "knobs", flags)
    assert not compare_outputs(ref, new)
    for sub in ("_STARgenome", "_STARpass1"):
        assert stat.S_IMODE(os.stat(ref + sub).st_mode) == stat.S_IMODE(os.stat(new + sub).st_mode), sub
    run = capi.HostRun(["--genomeDir", info["idx"], "--readFilesIn"] + info["fastq"] + ["--outFileNamePrefix", str(tmp_path / "k_"), "--genomeLoad", "LoadAndKeep"])
    run.close()
    base = ["--genomeDir", info["idx"], "--readFilesIn"] + info["fastq"] + ["--outFileNamePrefix", str(tmp_path / "e_")]
    for extra, text in [(["--genomeLoad", "LoadAndKeep", "--twopassMode", "Basic"], "2-pass method is not compatible with genomeLoad shared memory options"),
                        (["--genomeLoad", "Remove"], "--genomeLoad Remove"), (["--outStd", "Nowhere"], "outStd=Nowhere is not a valid value"),
                        (["--outSAMunmapped", "Within", "KeepAll"], "unrecognized option for --outSAMunmapped= Within KeepAll"),
                        (["--outMultimapperOrder", "Sorted"], "unknown/unimplemented value for --outMultimapperOrder: Sorted"),
                        (["--runDirPerm", "World"], "unrecognized option in --runDirPerm")]:
        with pytest.raises(RuntimeError) as e:
            capi.HostRun(base + extra)
        assert text in str(e.value), (extra, str(e.value))


def test_output_directory_is_created(tmp_path, built):
    """the directory part of --outFileNamePrefix is created with its parents (streamFuns.cpp createDirectory)"""
    info = prepare("se50", str(tmp_path), need_ref=False)
    prefix = str(tmp_path / "a" / "b" / "run_")
    run = capi.HostRun(["--genomeDir", info["idx"], "--readFilesIn"] + info["fastq"] + ["--outFileNamePrefix", prefix])
    run.close()
    assert os.path.isdir(str(tmp_path / "a" / "b")) and os.path.exists(prefix + "Log.out")


def test_numeric_flags_are_validated(tmp_path, built):
    """a value that is not a number is an input error (the reference's operator>> would carry on with 0); ADVICE round 1"""
    info = prepare("se50", str(tmp_path), need_ref=False)
    base = ["--genomeDir", info["idx"], "--readFilesIn"] + info["fastq"] + ["--outFileNamePrefix", str(tmp_path / "v_")]
    for flags, word in [(["--runThreadN", "abc"], "runThreadN"), (["--outFilterMismatchNoverLmax", "0.3x"], "outFilterMismatchNoverLmax"),
                        (["--outFilterMultimapNmax", "ten"], "outFilterMultimapNmax"), (["--outSJfilterOverhangMin", "30", "12", "12", "x"], "outSJfilterOverhangMin")]:
        with pytest.raises(RuntimeError) as e:
            capi.HostRun(base + flags)
        assert "expects a number" in str(e.value) and word in str(e.value), str(e.value)
    run = capi.HostRun(base + ["--outFilterMismatchNoverLmax", "3e-1", "--readMapNumber", "-1"])     # still numbers
    run.close()
