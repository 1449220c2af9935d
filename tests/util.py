"""Shared helpers for the parity tests."""
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from star_amd import capi, synth  # noqa: E402
from oracle import refstar, oracle_lib  # noqa: E402

# name -> (make_dataset kwargs, genomeGenerate kwargs, extra alignReads flags)
DATASETS = {
    # PE 2x101, spliced + genomic, repeats, Ns, annotated + novel junctions
    "pe101": (dict(seed=1, chr_lengths=(300000, 200000, 150000), n_tr=120, n_reads=3000, read_len=101, paired=True),
              dict(sa_index_nbases=8, use_gtf=True, sjdb_overhang=100), []),
    # SE 1x50, no annotation (config-1-like plumbing case)
    "se50": (dict(seed=2, chr_lengths=(250000,) * 4, n_tr=60, n_reads=3000, read_len=50, paired=False, n_rate=0.002),
             dict(sa_index_nbases=8, use_gtf=False), []),
    # PE 2x150 with indels, many repeats (multimappers, >50-locus seeds), higher error
    "pe150_indel": (dict(seed=3, chr_lengths=(400000, 300000), n_tr=100, n_reads=2500, read_len=150, paired=True,
                         sub_rate=0.02, indel_rate=0.15, repeat_families=((300, 150, 0.03), (80, 400, 0.0), (2000, 12, 0.01))),
                    dict(sa_index_nbases=8, use_gtf=True, sjdb_overhang=149), []),
    # short inserts: overlapping / protruding mates, plus non-default parameters
    "pe76_overlap": (dict(seed=4, chr_lengths=(200000, 200000), n_tr=80, n_reads=2500, read_len=76, paired=True, frag=(60, 160),
                          sub_rate=0.015),
                     dict(sa_index_nbases=7, use_gtf=True, sjdb_overhang=75),
                     ["--outFilterMultimapNmax", "20", "--alignSJoverhangMin", "8", "--outSAMattributes", "NH", "HI", "AS", "nM", "jM", "jI", "XS",
                      "--outSAMunmapped", "Within"]),
}


# non-default alignReads flags, one group per entry: every branch the flags of SURVEY.md 5.6 switch on the hot path and in post-map
PARAM_SWEEP = {
    "e2e": ["--alignEndsType", "EndToEnd"],
    "ext5p": ["--alignEndsType", "Extend5pOfRead1"],
    "ext5p12": ["--alignEndsType", "Extend5pOfReads12"],
    "intronmax": ["--alignIntronMax", "20000", "--alignMatesGapMax", "30000"],
    "motifs": ["--outFilterIntronMotifs", "RemoveNoncanonical"],
    "motifsU": ["--outFilterIntronMotifs", "RemoveNoncanonicalUnannotated", "--outSAMstrandField", "intronMotif"],
    "log2zero": ["--scoreGenomicLengthLog2scale", "0"],
    "scores": ["--scoreGap", "-1", "--scoreGapNoncan", "-6", "--scoreDelOpen", "-3", "--scoreInsBase", "-1", "--sjdbScore", "3", "--scoreStitchSJshift", "2"],
    "seeds": ["--seedSearchStartLmax", "30", "--seedPerWindowNmax", "30", "--winAnchorMultimapNmax", "100", "--seedMultimapNmax", "2000"],
    "lmax": ["--seedSearchLmax", "25"],
    "filters": ["--outFilterMismatchNmax", "3", "--outFilterScoreMinOverLread", "0.5", "--outFilterMatchNminOverLread", "0.5", "--outFilterMultimapNmax", "3",
                "--outFilterMultimapScoreRange", "3"],
    "sjo": ["--alignSJoverhangMin", "12", "--alignSJDBoverhangMin", "1", "--alignSplicedMateMapLmin", "20", "--alignSplicedMateMapLminOverLmate", "0.3"],
    "protrude": ["--alignEndsProtrude", "10", "ConcordantPair"],
    "insflush": ["--alignInsertionFlush", "Right"],
    "noclipref": ["--alignSoftClipAtReferenceEnds", "No"],
    "trN": ["--alignTranscriptsPerWindowNmax", "20", "--alignTranscriptsPerReadNmax", "200", "--alignWindowsPerReadNmax", "50"],
    "stitchmm": ["--alignSJstitchMismatchNmax", "1", "1", "1", "1"],
    "strands": ["--outFilterIntronStrands", "None"],
    "primary": ["--outSAMprimaryFlag", "AllBestScore", "--outSAMmapqUnique", "60", "--outSAMattrIHstart", "0", "--outSAMflagOR", "1024", "--outSAMunmapped", "Within"],
    "sjfilt": ["--outSJfilterReads", "Unique", "--outSJfilterOverhangMin", "20", "8", "8", "8", "--outSJfilterCountUniqueMin", "2", "1", "1", "1",
               "--outSJfilterDistToOtherSJmin", "5", "0", "3", "5"],
    "winbin": ["--winBinNbits", "14", "--winAnchorDistNbins", "5", "--winFlankNbins", "2"],
}


def prepare(name, workdir, need_ref=True):
    """Generate data set + reference index (+ reference outputs). Returns dict of paths."""
    ds_kw, gg_kw, extra = DATASETS[name]
    d = os.path.join(workdir, name)
    info = synth.make_dataset(d, **ds_kw)
    idx = os.path.join(d, "idx")
    gg = dict(gg_kw)
    use_gtf = gg.pop("use_gtf")
    refstar.genome_generate(info["fasta"], idx, gtf=info["gtf"] if use_gtf else None, **gg)
    info["idx"] = idx
    info["extra"] = extra
    if need_ref:
        refstar.align(idx, info["fastq"], os.path.join(d, "ref_"), threads=1, extra=extra)
        info["ref_prefix"] = os.path.join(d, "ref_")
    return info


def run_with_engine(info, prefix, engine_factory, batch_reads=777):
    """alignReads through the host library; `engine_factory(genome_p, params_p)` gives an object with map_batch/close."""
    argv = ["--genomeDir", info["idx"], "--readFilesIn"] + info["fastq"] + ["--outFileNamePrefix", prefix] + list(info.get("extra", []))
    run = capi.HostRun(argv)
    eng = engine_factory(run.genome, run.params)
    try:
        while True:
            while True:
                b = run.next_batch(batch_reads)
                if b is None:
                    break
                bufs = capi.ResultBuffers(b.nReads, tr_cap=b.nReads * 64)
                eng.map_batch(b, bufs)
                run.emit(bufs.res)
            if not run.in_pass1():
                break
            run.pass1_end()                       # --twopassMode Basic: junction insertion, then the 2nd pass
            eng.update_index(run.genome, run.params)
        run.finish()
    finally:
        eng.close()
        run.close()
    return prefix


def compare_outputs(ref_prefix, new_prefix):
    """Byte-exact: sorted SAM body, SJ.out.tab, Log.final.out counters. Returns list of problems."""
    problems = []
    a = refstar.sam_body_sorted(ref_prefix + "Aligned.out.sam")
    b = refstar.sam_body_sorted(new_prefix + "Aligned.out.sam")
    if a != b:
        sa, sb = set(a), set(b)
        only_a = sorted(sa - sb)[:3]
        only_b = sorted(sb - sa)[:3]
        problems.append("SAM differs: %d vs %d records; only-ref %r only-new %r" % (len(a), len(b), only_a, only_b))
    if open(ref_prefix + "SJ.out.tab", "rb").read() != open(new_prefix + "SJ.out.tab", "rb").read():
        problems.append("SJ.out.tab differs")
    if refstar.final_log_counters(ref_prefix + "Log.final.out") != refstar.final_log_counters(new_prefix + "Log.final.out"):
        problems.append("Log.final.out counters differ")
    return problems
