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
    # sparse suffix array (--genomeSAsparseD 3): every seed start is searched at 3 offsets (ReadAlign_maxMappableLength2strands.cpp:12-114)
    "pe101_sparse3": (dict(seed=6, chr_lengths=(300000, 200000, 150000), n_tr=120, n_reads=2500, read_len=101, paired=True),
                      dict(sa_index_nbases=8, use_gtf=True, sjdb_overhang=100, extra=("--genomeSAsparseD", "3")), []),
    # 2x150, 1 % errors, 5 % chimeric pairs (mates or read halves from different loci); chimeric detection stays
    # off by default, so these exercise multi-window stitching, soft clips and the "too short" path
    # found by the hardware fuzzer (round 3): with --alignEndsProtrude and 5' clipping the second mate can start before the first exon of the transcript;
    # the extension length of stitchAlignToTranscript.cpp:390 then goes "negative" and the reference's `(int) L` loop bound means no extension
    "pe125_protrude": (dict(seed=936057973, chr_lengths=(132791, 208918, 165515, 93736), n_tr=98, n_reads=2092, read_len=125, paired=True, sub_rate=0.02, n_rate=0.0, indel_rate=0.002,
                            frag=(62, 375)),
                       dict(sa_index_nbases=8, use_gtf=True, sjdb_overhang=100), []),
    # 2x300: the packed reads of 256 lanes do not fit into the dynamic LDS of a k_stitch_lane block; the batch takes the cooperative launches alone
    "pe300": (dict(seed=11, chr_lengths=(300000, 250000), n_tr=90, n_reads=900, read_len=300, paired=True, sub_rate=0.01, frag=(350, 900)),
              dict(sa_index_nbases=8, use_gtf=True, sjdb_overhang=100), []),
    "pe150_chim": (dict(seed=5, chr_lengths=(350000, 250000, 200000), n_tr=110, n_reads=3000, read_len=150, paired=True, sub_rate=0.01, chim_rate=0.05),
                   dict(sa_index_nbases=8, use_gtf=True, sjdb_overhang=149), []),
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
    "attrs": ["--outSAMattributes", "NH", "HI", "NM", "MD", "AS", "nM", "jM", "jI", "MC", "--outSAMunmapped", "Within"],
}


_PREPARED = {}


def _build_dataset(name, d):
    ds_kw, gg_kw, extra = DATASETS[name]
    info = synth.make_dataset(d, **ds_kw)
    gg = dict(gg_kw)
    use_gtf = gg.pop("use_gtf")
    refstar.genome_generate(info["fasta"], os.path.join(d, "idx"), gtf=info["gtf"] if use_gtf else None, **gg)
    return info


def prepare(name, workdir, need_ref=True):
    """Data set + reference index (+ reference outputs) under workdir/name. Returns dict of paths.  The genome, the reads and the index are generated once
    per test process (they are deterministic and read-only) and linked into every test's own directory, where its outputs go."""
    import pickle
    import shutil
    import tempfile
    ds_kw, gg_kw, extra = DATASETS[name]
    if name not in _PREPARED:
        root = tempfile.mkdtemp(prefix="staramd_data_%s_" % name)
        info0 = _build_dataset(name, os.path.join(root, name))
        _PREPARED[name] = (os.path.join(root, name), pickle.dumps(info0))
        import atexit
        atexit.register(shutil.rmtree, root, True)
    src, blob = _PREPARED[name]
    d = os.path.join(workdir, name)
    os.makedirs(d, exist_ok=True)
    info = pickle.loads(blob)
    for k, v in list(info.items()):                 # fasta, gtf, fastq: links into the test's directory
        if isinstance(v, str) and v.startswith(src):
            dst = os.path.join(d, os.path.relpath(v, src)); os.symlink(v, dst) if not os.path.lexists(dst) else None; info[k] = dst
        elif isinstance(v, list) and v and isinstance(v[0], str) and v[0].startswith(src):
            out = []
            for x in v:
                dst = os.path.join(d, os.path.relpath(x, src)); os.symlink(x, dst) if not os.path.lexists(dst) else None; out.append(dst)
            info[k] = out
    idx = os.path.join(d, "idx")
    if not os.path.lexists(idx):
        os.symlink(os.path.join(src, "idx"), idx)
    info["idx"] = idx
    info["extra"] = list(extra)
    if need_ref:
        refstar.align(idx, info["fastq"], os.path.join(d, "ref_"), threads=1, extra=extra)
        info["ref_prefix"] = os.path.join(d, "ref_")
    return info


def _map(eng, b):
    for per_read in (64, 256, 1024):          # the caller owns the result arrays: grow and call again when they are too small
        bufs = capi.ResultBuffers(b.nReads, tr_cap=b.nReads * per_read)
        try:
            eng.map_batch(b, bufs)
            return bufs
        except RuntimeError as e:
            if per_read == 1024 or not ("-3" in str(e) or "too small" in str(e)):
                raise


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
                bufs = _map(eng, b)
                mb = run.merged_batch()                   # --peOverlapNbasesMin: overlapping mates merged into single reads, a second batch
                mbufs = _map(eng, mb) if mb is not None else None
                wb = run.wasp_batch(bufs.res)             # --waspOutputMode: allele-swapped copies of some reads, one more batch
                wbufs = _map(eng, wb) if wb is not None else None
                run.wasp_results(bufs.res, wbufs.res if wbufs is not None else None)
                run.emit(bufs.res, mbufs.res if mbufs is not None else None)
            phase = run.next_phase()
            if phase == 0:
                break
            if phase == 1:                        # --twopassMode Basic: junctions were inserted into the host index
                eng.update_index(run.genome, run.params)
            else:                                 # --outFilterType BySJout: 2nd stage, held reads against the filtered junctions
                eng.set_novel_junctions(*run.novel_junctions())
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
        short = lambda rec: b" ".join(f for i, f in enumerate(rec.rstrip(b"\n").split(b"\t")) if i not in (9, 10))      # (without SEQ / QUAL: the message stays readable)
        only_a = [short(r) for r in sorted(sa - sb)[:3]]
        only_b = [short(r) for r in sorted(sb - sa)[:3]]
        problems.append("SAM differs: %d vs %d records; only-ref %r only-new %r" % (len(a), len(b), only_a, only_b))
    if open(ref_prefix + "SJ.out.tab", "rb").read() != open(new_prefix + "SJ.out.tab", "rb").read():
        problems.append("SJ.out.tab differs")
    if refstar.final_log_counters(ref_prefix + "Log.final.out") != refstar.final_log_counters(new_prefix + "Log.final.out"):
        problems.append("Log.final.out counters differ")
    return problems


def _read_fasta(path):
    seqs, cur = [], []
    for l in open(path):
        if l.startswith(">"):
            if cur:
                seqs.append("".join(cur))
            cur = []
        else:
            cur.append(l.strip())
    seqs.append("".join(cur))
    return seqs


def _rc(s):
    return s[::-1].translate(str.maketrans("ACGTNacgtn", "TGCANtgcan"))


def make_edge_reads(info, outdir, paired=True, seed=7):
    """Hand-made reads for the corner cases of the hot path, on the genome of a prepared data set: shortest / longest reads,
    all-N, homopolymers, N-riddled, chromosome ends, fully overlapping / chimeric / junk mates, lower case and IUPAC codes,
    heavy mismatches, tiny middle exon -- mixed into ordinary reads.  Returns the FASTQ paths."""
    import numpy as np
    rng = np.random.default_rng(seed)
    G = _read_fasta(info["fasta"])
    g0 = G[0]

    def pick(L, c=None):
        c = G[int(rng.integers(len(G)))] if c is None else c
        p = int(rng.integers(0, len(c) - L))
        return c[p:p + L]

    def mutate(s, n):
        s = list(s)
        for p in rng.choice(len(s), size=min(n, len(s)), replace=False):
            s[p] = "ACGT"[("ACGT".index(s[p]) + 1 + int(rng.integers(3))) % 4] if s[p] in "ACGT" else "A"
        return "".join(s)

    junk = "".join("ACGT"[i] for i in rng.integers(0, 4, 150))
    frag = pick(400)
    pairs = [
        ("A", _rc(frag[250:400])),                                    # 1-base mate
        ("N" * 50, "N" * 50),                                         # nothing mappable
        (pick(325, g0), _rc(pick(324, g0))),                          # Lread = 650, the maximum
        (frag[:12], _rc(frag[200:213])),                              # seedSplitMin edges
        ("A" * 150, "T" * 150), ("AC" * 75, "GT" * 75),              # low complexity: very many loci
        ("".join(c if i % 10 else "N" for i, c in enumerate(frag[:150])), _rc(frag[250:400])),
        (frag[:150], _rc(frag[:150])),                                # mates overlap completely
        (pick(150, G[0]), _rc(pick(150, G[-1]))),                     # mates from different chromosomes
        (frag[:150], junk),                                           # one mate is junk
        (frag[:150].lower(), _rc(frag[250:400]).replace("A", "R", 2).replace("C", "Y", 1)),
        (mutate(frag[:150], 20), _rc(mutate(frag[250:400], 25))),     # far too many mismatches
        (g0[:120], _rc(g0[150:300])), (G[-1][-300:-160], _rc(G[-1][-140:])),      # chromosome start / end
        (g0[5000:5060] + g0[6000:6006] + g0[7000:7084], _rc(g0[7100:7250])),      # 6-base middle exon
        (frag[:30], _rc(frag[250:400])), (frag[:150], _rc(frag[370:400])),        # very different mate lengths
    ]
    for _ in range(300):                                              # ordinary company, some with indels / mismatches
        f = pick(int(rng.integers(160, 420)))
        a, b = f[:int(rng.integers(40, 151))], _rc(f[-int(rng.integers(40, 151)):])
        if rng.random() < 0.3:
            a = mutate(a, int(rng.integers(1, 6)))
        if rng.random() < 0.2 and len(b) > 60:
            b = b[:30] + b[33:]
        if rng.random() < 0.2 and len(a) > 60:
            a = a[:40] + "GATTACA"[:int(rng.integers(1, 8))] + a[40:]
        pairs.append((a, b))
    order = rng.permutation(len(pairs))
    os.makedirs(outdir, exist_ok=True)
    paths = [os.path.join(outdir, "edge_1.fq")] + ([os.path.join(outdir, "edge_2.fq")] if paired else [])
    fo = [open(p, "w") for p in paths]
    for k, i in enumerate(order):
        for m in range(len(paths)):
            s = pairs[i][m]
            fo[m].write("@edge%d/%d\n%s\n+\n%s\n" % (k, m + 1, s, "".join(chr(33 + (7 * j + k) % 40) for j in range(len(s)))))
    for f in fo:
        f.close()
    return paths


def bam_parts(path):
    """(header text, reference block, list of records) of a BAM file, BGZF blocks decompressed with Python's gzip module."""
    import gzip
    import struct
    d = gzip.open(path, "rb").read()
    assert d[:4] == b"BAM\x01"
    lt = struct.unpack("<i", d[4:8])[0]
    text = d[8:8 + lt]
    p0 = p = 8 + lt
    nref = struct.unpack("<i", d[p:p + 4])[0]
    p += 4
    for _ in range(nref):
        ln = struct.unpack("<i", d[p:p + 4])[0]
        p += 4 + ln + 4
    recs = []
    q = p
    while q < len(d):
        bs = struct.unpack("<i", d[q:q + 4])[0]
        recs.append(d[q:q + 4 + bs])
        q += 4 + bs
    return text, d[p0:p], recs
