#!/usr/bin/env python3
"""Time `star_amd --runMode genomeGenerate` (suffix array + SAindex on the MI355X) on a synthetic genome of a given size,
optionally against the reference's genomeGenerate on the same FASTA (file-for-file comparison).
    python tools/index_scale.py --mb 1000 --nb 14 [--ref] [--gtf] [--dir /dev/shm/idxscale]"""
import argparse, filecmp, json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mb", type=int, default=100); ap.add_argument("--nb", type=int, default=0); ap.add_argument("--chr", type=int, default=0)
    ap.add_argument("--ref", action="store_true"); ap.add_argument("--gtf", action="store_true"); ap.add_argument("--dir", default="/dev/shm/idxscale")
    ap.add_argument("--keep", action="store_true")
    a = ap.parse_args()
    import math
    import numpy as np
    from star_amd import synth
    from oracle import refstar
    nb = a.nb or max(4, min(14, int(math.log2(a.mb * 1e6) / 2 - 1)))
    nchr = a.chr or max(1, min(24, a.mb // 40))
    d = os.path.join(a.dir, "g%d" % a.mb); os.makedirs(d, exist_ok=True)
    out = {"mb": a.mb, "nb": nb, "nchr": nchr}
    t = time.time(); seqs, frac = synth.make_genome_large(20260922, a.mb, nchr); out["genome_s"] = time.time() - t; out["repeat_frac"] = frac
    names = ["chr%d" % (i + 1) for i in range(nchr)]
    gtf = None
    if a.gtf:
        rng = np.random.default_rng(5)
        t = time.time(); trs = synth.make_transcripts(rng, seqs, 60 * a.mb); out["transcripts_s"] = time.time() - t; out["n_tr"] = len(trs)
        gtf = os.path.join(d, "annot.gtf"); synth.write_gtf(gtf, names, trs, np.ones(len(trs), dtype=bool))
    fa = os.path.join(d, "genome.fa")
    t = time.time(); synth._write_fasta(fa, names, seqs); out["fasta_write_s"] = time.time() - t
    del seqs
    new = os.path.join(d, "new"); os.makedirs(new, exist_ok=True)
    cmd = [os.path.join(ROOT, "star_amd", "bin", "star_amd"), "--runMode", "genomeGenerate", "--genomeDir", new, "--genomeFastaFiles", fa, "--genomeSAindexNbases", str(nb),
           "--runThreadN", str(os.cpu_count()), "--outFileNamePrefix", new + "/_log_"]
    if gtf:
        cmd += ["--sjdbGTFfile", gtf, "--sjdbOverhang", "100"]
    env = dict(os.environ, STARAMD_HOST_TIMING="1")
    t = time.time(); p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env); out["star_amd_generate_s"] = time.time() - t
    out["star_amd_rc"] = p.returncode; out["star_amd_stderr"] = p.stderr[-1500:]
    if a.ref and p.returncode == 0:
        ref = os.path.join(d, "ref")
        t = time.time(); refstar.genome_generate(fa, ref, gtf=gtf, sjdb_overhang=100, sa_index_nbases=nb, threads=os.cpu_count(),
                                                  extra=["--limitGenomeGenerateRAM", str(200 << 30)]); out["reference_generate_s"] = time.time() - t
        files = ["Genome", "SA", "SAindex", "chrStart.txt"] + (["sjdbInfo.txt", "sjdbList.out.tab", "exonInfo.tab"] if gtf else [])
        out["identical"] = {f: filecmp.cmp(os.path.join(ref, f), os.path.join(new, f), shallow=False) for f in files}
        out["reference_log"] = [l.strip() for l in open(os.path.join(ref, "Log.out")) if " ... " in l or "..... " in l or "Finished" in l][-20:]
    print(json.dumps(out, indent=1))
    if not a.keep:
        import shutil; shutil.rmtree(d, ignore_errors=True)


if __name__ == "__main__":
    main()
