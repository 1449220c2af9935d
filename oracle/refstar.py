"""Thin runner for the REFERENCE binary built by oracle/Makefile.ref (oracle/_ref/STAR).

TEST INFRASTRUCTURE ONLY.  Used by tests/, by tests/golden/make_golden.py, by
__graft_entry__.smoke() and by bench.py's data preparation + cpu_baseline leg.  The product
(star_amd/) never imports this module.
"""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
REF_BIN = os.path.join(HERE, "_ref", "STAR")


def have_ref():
    return os.path.isfile(REF_BIN) and os.access(REF_BIN, os.X_OK)


def build_ref(jobs=8):
    """Compile oracle/_ref/STAR if /root/reference is present."""
    if not os.path.isdir("/root/reference/source"):
        return have_ref()
    targets = ["all"]
    subprocess.check_call(["make", "-f", os.path.join(HERE, "Makefile.ref"), "-j%d" % jobs] + targets,
                          cwd=os.path.dirname(HERE), stdout=subprocess.DEVNULL)
    return have_ref()


def genome_generate(fasta, outdir, gtf=None, sjdb_overhang=100, sa_index_nbases=14, threads=8,
                    chr_bin_nbits=18, extra=()):
    """reference `--runMode genomeGenerate` (index building is out of scope: SURVEY.md section 2 row 10)."""
    os.makedirs(outdir, exist_ok=True)
    cmd = [REF_BIN, "--runMode", "genomeGenerate", "--genomeDir", outdir, "--genomeFastaFiles", fasta,
           "--genomeSAindexNbases", str(sa_index_nbases), "--runThreadN", str(threads),
           "--genomeChrBinNbits", str(chr_bin_nbits),
           "--outFileNamePrefix", outdir.rstrip("/") + "/_gg_", "--outTmpDir", outdir.rstrip("/") + "/_gg_tmp"]
    if gtf is not None:
        cmd += ["--sjdbGTFfile", gtf, "--sjdbOverhang", str(sjdb_overhang)]
    cmd += list(extra)
    subprocess.check_call(cmd, stdout=subprocess.DEVNULL)
    return outdir


def align(genome_dir, fastqs, outprefix, threads=1, extra=(), binary=None, timeout=None):
    """reference `--runMode alignReads`; returns the output prefix."""
    os.makedirs(os.path.dirname(outprefix) or ".", exist_ok=True)
    tmp = outprefix + "_STARtmp"
    if os.path.isdir(tmp):
        shutil.rmtree(tmp)
    cmd = [binary or REF_BIN, "--runMode", "alignReads", "--genomeDir", genome_dir, "--readFilesIn"] + list(fastqs) + \
          ["--runThreadN", str(threads), "--outFileNamePrefix", outprefix] + list(extra)
    subprocess.run(cmd, stdout=subprocess.DEVNULL, check=True, timeout=timeout)
    return outprefix


def sam_body_sorted(path):
    """Non-header SAM lines, byte-sorted (thread interleaving reorders records)."""
    with open(path, "rb") as f:
        lines = [l for l in f if not l.startswith(b"@")]
    lines.sort()
    return lines


def final_log_counters(path):
    """Counter lines of Log.final.out (drop the three time stamps and the speed line)."""
    out = []
    with open(path) as f:
        for l in f:
            if "|" not in l:
                continue
            k, v = l.split("|", 1)
            k = k.strip()
            if k.startswith(("Started", "Finished", "Mapping speed")):
                continue
            out.append((k, v.strip()))
    return out
