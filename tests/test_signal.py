"""--outWigType bedGraph | wiggle [read1_5p | read2] with --outWigStrand / --outWigNorm / --outWigReferencesPrefix: the coverage tracks the reference
writes from its sorted BAM at the end of an alignReads run (signalFromBAM.cpp), here from the sorted records still in memory (signal.cpp)."""
import os

import pytest

from util import capi, oracle_lib, prepare, refstar, run_with_engine

pytestmark = pytest.mark.skipif(not refstar.have_ref(), reason="oracle/_ref/STAR not built (no /root/reference here)")

CASES = [("pe101", ["--outWigType", "bedGraph"]),
         ("pe101", ["--outWigType", "wiggle", "--outWigStrand", "Unstranded", "--outWigNorm", "None"]),
         ("pe101_sparse3", ["--outWigType", "bedGraph", "read2", "--outWigNorm", "None", "--outFilterMultimapNmax", "30"]),
         ("pe76_overlap", ["--outWigType", "bedGraph", "read1_5p", "--outWigReferencesPrefix", "chr2"]),
         ("se50", ["--outWigType", "wiggle", "read1_5p", "--outSAMunmapped", "Within"])]


@pytest.mark.parametrize("name,more", CASES)
def test_signal_tracks(name, more, tmp_path, built):
    info = dict(prepare(name, str(tmp_path), need_ref=False))
    d = os.path.dirname(info["fastq"][0])
    info["extra"] = [x for x in info["extra"]] + more + ["--outSAMtype", "BAM", "SortedByCoordinate"]
    ref = refstar.align(info["idx"], info["fastq"], os.path.join(d, "refS_"), threads=1, extra=info["extra"])
    new = run_with_engine(info, os.path.join(d, "newS_"), lambda g, p: oracle_lib.Oracle(g, p))
    files = sorted(f[len("refS_"):] for f in os.listdir(d) if f.startswith("refS_Signal."))
    assert len(files) == (2 if "Unstranded" in more else 4)
    total = 0
    for f in files:
        a, b = open(os.path.join(d, "refS_" + f), "rb").read(), open(os.path.join(d, "newS_" + f), "rb").read()
        assert a == b, f
        total += len(a)
    assert total > 10000


def test_signal_needs_sorted_bam(tmp_path, built):
    info = prepare("se50", str(tmp_path), need_ref=False)
    with pytest.raises(RuntimeError) as e:
        capi.HostRun(["--genomeDir", info["idx"], "--readFilesIn"] + info["fastq"] + ["--outFileNamePrefix", str(tmp_path / "e_"), "--outWigType", "bedGraph"])
    assert "generating signal with --outWigType requires sorted BAM" in str(e.value)


def test_signal_from_bam_file_with_the_cli(tmp_path, built):
    """--runMode inputAlignmentsFromBAM --inputBAMfile x --outWigType ...: ENCODE's second step.  A tool mode of the star_amd binary: no index, no reads, no GPU."""
    import subprocess
    cli = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "star_amd", "bin", "star_amd")
    info = dict(prepare("pe101", str(tmp_path), need_ref=False))
    d = os.path.dirname(info["fastq"][0])
    ref = refstar.align(info["idx"], info["fastq"], os.path.join(d, "refB_"), threads=1, extra=list(info["extra"]) + ["--outSAMtype", "BAM", "SortedByCoordinate"])
    bam = ref + "Aligned.sortedByCoord.out.bam"
    for k, flags in enumerate((["--outWigType", "bedGraph"], ["--outWigType", "wiggle", "read1_5p", "--outWigStrand", "Unstranded", "--outWigNorm", "None", "--outWigReferencesPrefix", "chr1"])):
        a, b = os.path.join(d, "sigref%d_" % k), os.path.join(d, "signew%d_" % k)
        subprocess.check_call([refstar.REF_BIN, "--runMode", "inputAlignmentsFromBAM", "--inputBAMfile", bam, "--outFileNamePrefix", a] + flags, stdout=subprocess.DEVNULL)
        subprocess.check_call([cli, "--runMode", "inputAlignmentsFromBAM", "--inputBAMfile", bam, "--outFileNamePrefix", b] + flags)
        files = sorted(f[len("sigref%d_" % k):] for f in os.listdir(d) if f.startswith("sigref%d_Signal." % k))
        assert len(files) in (2, 4)
        for f in files:
            assert open(a + f, "rb").read() == open(b + f, "rb").read(), f
    r = subprocess.run([cli, "--runMode", "inputAlignmentsFromBAM", "--inputBAMfile", bam, "--outFileNamePrefix", os.path.join(d, "x_")], stderr=subprocess.PIPE)
    assert r.returncode != 0 and b"--runMode inputFromBAM only works with --outWigType" in r.stderr
