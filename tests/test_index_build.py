"""Index generation: `star_amd --runMode genomeGenerate` against the reference's genomeGenerate, file for file.

The suffix array sort and the SAindex are built by star_amd/csrc/index/index_core.h: on the GPU through rocPRIM sorts / scans
(index_gpu.hip, the `-m gpu` tests), here through the same code on a plain-loop backend (oracle/index_emul.cpp behind
oracle/cli_shim.cpp) so that the logic is pinned against the reference without a GPU.
"""
import filecmp
import os
import subprocess

import numpy as np
import pytest

import util
from star_amd import synth
from oracle import refstar

ROOT = util.ROOT
ORACLE_CLI = os.path.join(ROOT, "oracle", "_build", "star_amd_oracle_cli")
GPU_CLI = os.path.join(ROOT, "star_amd", "bin", "star_amd")
FILES = ["Genome", "SA", "SAindex", "chrName.txt", "chrStart.txt", "chrLength.txt", "chrNameLength.txt"]
FILES_SJDB = ["sjdbInfo.txt", "sjdbList.out.tab", "exonInfo.tab", "transcriptInfo.tab", "geneInfo.tab", "exonGeTrInfo.tab", "sjdbList.fromGTF.out.tab"]

CASES = {
    # name: (chr lengths, repeat families, N runs, SAindexNbases, with annotation, sjdbOverhang)
    "plain": ((300000, 200000, 150000), ((300, 100, 0.05), (2000, 8, 0.01), (60, 150, 0.0)), 12, 8, False, 0),
    "annot": ((300000, 250000), ((300, 100, 0.05), (60, 150, 0.0)), 6, 8, True, 100),
    "small_bins": ((70000, 1500, 90000, 300), (), 3, 5, True, 49),
    "deep_index": ((120000, 80000), ((500, 40, 0.02),), 4, 11, False, 0),
}


def _make_case(d, name):
    lens, fams, nruns, nb, annot, ov = CASES[name]
    rng = np.random.default_rng(sorted(CASES).index(name) + 7)
    seqs = synth.make_genome(rng, list(lens), repeat_families=fams, n_runs=nruns)
    if len(seqs[0]) > 30000:
        seqs[0][1000:1500] = ord("N")                       # a long N run: suffixes that only differ behind it
        big = max(range(1, len(seqs)), key=lambda i: len(seqs[i]))
        seqs[big][2000:6000] = seqs[0][20000:24000]          # a long exact duplicate: many doubling rounds
        seqs[0][len(seqs[0]) - 40:] = ord("A")              # poly-A up to the chromosome end: ties broken by the padding rule
    names = ["chr%d" % (i + 1) for i in range(len(seqs))]
    trs = synth.make_transcripts(rng, seqs, 60) if annot else []
    os.makedirs(d, exist_ok=True)
    fa, gtf = os.path.join(d, "genome.fa"), os.path.join(d, "annot.gtf")
    synth._write_fasta(fa, names, seqs)
    if annot:
        synth.write_gtf(gtf, names, trs, np.ones(len(trs), dtype=bool))
    return fa, (gtf if annot else None), nb, ov


def _generate(cli, fa, gtf, nb, ov, out, bins=None):
    cmd = [cli, "--runMode", "genomeGenerate", "--genomeDir", out, "--genomeFastaFiles", fa, "--genomeSAindexNbases", str(nb), "--runThreadN", "4",
           "--outFileNamePrefix", out + "/_log_"]
    if bins:
        cmd += ["--genomeChrBinNbits", str(bins)]
    if gtf:
        cmd += ["--sjdbGTFfile", gtf, "--sjdbOverhang", str(ov)]
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert p.returncode == 0, p.stderr[-2000:]
    return p.stderr


def _compare_dirs(ref, new, with_sjdb):
    for f in FILES + (FILES_SJDB if with_sjdb else []):
        assert os.path.isfile(os.path.join(new, f)), f + " missing"
        assert filecmp.cmp(os.path.join(ref, f), os.path.join(new, f), shallow=False), f + " differs from the reference's"

    def params(p):
        return [l for l in open(p) if not l.startswith("### ") or "GstrandBit" in l]
    assert params(os.path.join(ref, "genomeParameters.txt")) == params(os.path.join(new, "genomeParameters.txt"))


def _run_case(cli, tmp, name):
    if not refstar.have_ref():
        pytest.skip("oracle/_ref/STAR not built")
    fa, gtf, nb, ov = _make_case(tmp, name)
    bins = 12 if name == "small_bins" else None
    ref, new = os.path.join(tmp, "ref"), os.path.join(tmp, "new")
    refstar.genome_generate(fa, ref, gtf=gtf, sjdb_overhang=ov or 100, sa_index_nbases=nb, threads=4, chr_bin_nbits=bins or 18)
    os.makedirs(new, exist_ok=True)
    _generate(cli, fa, gtf, nb, ov, new, bins)
    _compare_dirs(ref, new, gtf is not None)
    return ref, new


@pytest.mark.parametrize("name", sorted(CASES))
def test_generate_logic_vs_reference(tmp_path, name):
    """same algorithm code as the device build, plain-loop backend: every genomeDir file equals the reference's"""
    _run_case(ORACLE_CLI, str(tmp_path), name)


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(CASES))
def test_generate_gpu_vs_reference(tmp_path, name):
    """the HIP build (rocPRIM radix sorts and scans on the MI355X): every genomeDir file equals the reference's"""
    _run_case(GPU_CLI, str(tmp_path), name)
