"""Generate the committed golden fixtures under tests/golden/ by RUNNING THE REFERENCE (oracle/_ref/STAR, built from
/root/reference/source by oracle/Makefile.ref) in this container.

  tiny_pe/      a complete tiny case that travels without the reference binary:
                  idx/            genomeDir produced by the reference's genomeGenerate (80 kb genome, sjdb from a GTF)
                  reads_1.fq, reads_2.fq   600 synthetic 2x76 pairs (repeats, Ns, indels, annotated + novel junctions)
                  ref_Aligned.sorted.sam   body of the reference's Aligned.out.sam, sorted
                  ref_SJ.out.tab, ref_Log.final.counters.json
                  ref2p_*                  the same for `--twopassMode Basic --sjdbInsertSave All`, plus the 1st-pass SJ.out.tab and
                                           the sha256 of the index the reference left in _STARgenome/ after junction insertion
  digests.json  sha256 of the same three reference outputs for every data set of tests/util.py:DATASETS
                (the data sets are regenerated from fixed seeds by star_amd/synth.py)

Usage (from the repo root):  python tests/golden/make_golden.py
"""
import hashlib
import json
import os
import shutil
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from star_amd import synth  # noqa: E402
from oracle import refstar  # noqa: E402
import util  # noqa: E402

TINY = dict(seed=11, chr_lengths=(50000, 30000), n_tr=30, n_reads=600, read_len=76, paired=True, sub_rate=0.015, indel_rate=0.1,
            repeat_families=((200, 20, 0.02), (60, 60, 0.0)), n_rate=0.002)


def digest_outputs(prefix):
    body = b"".join(refstar.sam_body_sorted(prefix + "Aligned.out.sam"))
    sj = open(prefix + "SJ.out.tab", "rb").read()
    log = json.dumps(refstar.final_log_counters(prefix + "Log.final.out"), sort_keys=True).encode()
    return {"sam_sorted_sha256": hashlib.sha256(body).hexdigest(), "sj_sha256": hashlib.sha256(sj).hexdigest(),
            "log_counters_sha256": hashlib.sha256(log).hexdigest(), "n_sam_records": body.count(b"\n")}


def main():
    if not refstar.have_ref():
        raise SystemExit("oracle/_ref/STAR missing: python -c 'import __graft_entry__ as g; g.build()'")
    out = os.path.join(HERE, "tiny_pe")
    shutil.rmtree(out, ignore_errors=True)
    with tempfile.TemporaryDirectory() as tmp:
        info = synth.make_dataset(os.path.join(tmp, "tiny"), **TINY)
        idx = os.path.join(tmp, "tiny", "idx")
        refstar.genome_generate(info["fasta"], idx, gtf=info["gtf"], sjdb_overhang=75, sa_index_nbases=6)
        refstar.align(idx, info["fastq"], os.path.join(tmp, "ref_"), threads=1)
        os.makedirs(os.path.join(out, "idx"))
        for f in os.listdir(idx):
            if f in ("Log.out", "exonInfo.tab", "geneInfo.tab", "transcriptInfo.tab", "exonGeTrInfo.tab", "sjdbList.fromGTF.out.tab"):
                continue
            shutil.copy(os.path.join(idx, f), os.path.join(out, "idx", f))
        for i, fq in enumerate(info["fastq"]):
            shutil.copy(fq, os.path.join(out, "reads_%d.fq" % (i + 1)))
        with open(os.path.join(out, "ref_Aligned.sorted.sam"), "wb") as f:
            f.write(b"".join(refstar.sam_body_sorted(os.path.join(tmp, "ref_Aligned.out.sam"))))
        shutil.copy(os.path.join(tmp, "ref_SJ.out.tab"), os.path.join(out, "ref_SJ.out.tab"))
        json.dump(refstar.final_log_counters(os.path.join(tmp, "ref_Log.final.out")), open(os.path.join(out, "ref_Log.final.counters.json"), "w"), indent=1, sort_keys=True)
        # 2-pass run of the reference on the same tiny case
        refstar.align(idx, info["fastq"], os.path.join(tmp, "ref2p_"), threads=1, extra=["--twopassMode", "Basic", "--sjdbInsertSave", "All"])
        with open(os.path.join(out, "ref2p_Aligned.sorted.sam"), "wb") as f:
            f.write(b"".join(refstar.sam_body_sorted(os.path.join(tmp, "ref2p_Aligned.out.sam"))))
        shutil.copy(os.path.join(tmp, "ref2p_SJ.out.tab"), os.path.join(out, "ref2p_SJ.out.tab"))
        shutil.copy(os.path.join(tmp, "ref2p__STARpass1", "SJ.out.tab"), os.path.join(out, "ref2p_pass1_SJ.out.tab"))
        json.dump(refstar.final_log_counters(os.path.join(tmp, "ref2p_Log.final.out")), open(os.path.join(out, "ref2p_Log.final.counters.json"), "w"), indent=1, sort_keys=True)
        json.dump({f: hashlib.sha256(open(os.path.join(tmp, "ref2p__STARgenome", f), "rb").read()).hexdigest() for f in ("Genome", "SA", "SAindex", "sjdbInfo.txt", "sjdbList.out.tab")},
                  open(os.path.join(out, "ref2p_index_sha256.json"), "w"), indent=1, sort_keys=True)
        dig = {}
        for name in sorted(util.DATASETS):
            i2 = util.prepare(name, tmp)
            dig[name] = digest_outputs(i2["ref_prefix"])
        json.dump(dig, open(os.path.join(HERE, "digests.json"), "w"), indent=1, sort_keys=True)
    print("golden fixtures written:", out, os.path.join(HERE, "digests.json"))


if __name__ == "__main__":
    main()
