"""Synthetic genomes, annotations and reads for the parity tests and for bench.py.

There is no genome / GTF / FASTQ in the image and no network, so every data set used
by this project is generated here from fixed seeds.  Everything is
vectorised with numpy so that bench-size inputs (100+ Mb genome, millions of pairs)
are produced in seconds.

Layout of a generated data set directory:
    genome.fa          FASTA, 60 bases per line
    annot.gtf          exon records of the annotated transcripts (input of --sjdbGTFfile)
    reads_1.fq [reads_2.fq]
"""
import os
import numpy as np

_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
_COMP = np.zeros(256, dtype=np.uint8)
for a, b in zip(b"ACGTN", b"TGCAN"):
    _COMP[a] = b


def _write_fasta(path, names, seqs, width=60):
    with open(path, "wb") as f:
        for name, s in zip(names, seqs):
            f.write(b">" + name.encode() + b"\n")
            n = len(s)
            full = (n // width) * width
            if full:
                body = np.empty((n // width, width + 1), dtype=np.uint8)
                body[:, :width] = s[:full].reshape(-1, width)
                body[:, width] = 10
                body.tofile(f)
            if n > full:
                s[full:].tofile(f)
                f.write(b"\n")


def make_genome(rng, chr_lengths, repeat_families=(), n_runs=0):
    """Random ACGT chromosomes with optional repeat families and N runs.

    repeat_families: iterable of (unit_len, copies, divergence): a random unit is pasted
    `copies` times at random places (either strand), each copy with per-base substitutions.
    """
    seqs = [_ACGT[rng.integers(0, 4, size=n)] for n in chr_lengths]
    for unit_len, copies, div in repeat_families:
        unit = _ACGT[rng.integers(0, 4, size=unit_len)]
        for _ in range(copies):
            c = int(rng.integers(0, len(seqs)))
            if len(seqs[c]) <= unit_len + 2:
                continue
            p = int(rng.integers(0, len(seqs[c]) - unit_len))
            cp = unit.copy()
            m = rng.random(unit_len) < div
            cp[m] = _ACGT[rng.integers(0, 4, size=int(m.sum()))]
            if rng.random() < 0.5:
                cp = _COMP[cp[::-1]]
            seqs[c][p:p + unit_len] = cp
    for _ in range(n_runs):
        c = int(rng.integers(0, len(seqs)))
        ln = int(rng.integers(1, 30))
        if len(seqs[c]) <= ln + 2:
            continue
        p = int(rng.integers(0, len(seqs[c]) - ln))
        seqs[c][p:p + ln] = ord("N")
    return seqs


def make_transcripts(rng, seqs, n_tr, exon_len=(40, 300), intron_len=(30, 4000), n_exons=(1, 6),
                     noncanon_frac=0.05):
    """Multi-exon transcripts; introns get GT..AG (+) or CT..AC (-) written INTO the genome
    (unless non-canonical) so that junction motifs are meaningful.
    Returns list of dicts {chr, strand, exons: [(start,end) 0-based inclusive]}.
    Transcripts never overlap each other so that motif edits do not collide.
    """
    trs = []
    cursor = [1000] * len(seqs)
    order = rng.integers(0, len(seqs), size=n_tr)
    for c in order:
        c = int(c)
        ne = int(rng.integers(n_exons[0], n_exons[1] + 1))
        el = rng.integers(exon_len[0], exon_len[1] + 1, size=ne)
        il = rng.integers(intron_len[0], intron_len[1] + 1, size=max(ne - 1, 0))
        span = int(el.sum() + il.sum())
        gap = int(rng.integers(50, 3000))
        start = cursor[c] + gap
        if start + span + 1000 >= len(seqs[c]):
            continue
        cursor[c] = start + span
        strand = "+" if rng.random() < 0.5 else "-"
        exons = []
        p = start
        for i in range(ne):
            exons.append((p, p + int(el[i]) - 1))
            p += int(el[i])
            if i < ne - 1:
                # intron [p, p+il-1]
                q = p + int(il[i]) - 1
                if rng.random() >= noncanon_frac:
                    if strand == "+":
                        seqs[c][p:p + 2] = np.frombuffer(b"GT", dtype=np.uint8)
                        seqs[c][q - 1:q + 1] = np.frombuffer(b"AG", dtype=np.uint8)
                    else:
                        seqs[c][p:p + 2] = np.frombuffer(b"CT", dtype=np.uint8)
                        seqs[c][q - 1:q + 1] = np.frombuffer(b"AC", dtype=np.uint8)
                p = q + 1
        trs.append({"chr": c, "strand": strand, "exons": exons})
    return trs


def write_gtf(path, names, trs, annotated_mask):
    with open(path, "w") as f:
        for i, (t, keep) in enumerate(zip(trs, annotated_mask)):
            if not keep:
                continue
            for (s, e) in t["exons"]:
                f.write("%s\tsynth\texon\t%d\t%d\t.\t%s\t.\tgene_id \"g%d\"; transcript_id \"t%d\";\n"
                        % (names[t["chr"]], s + 1, e + 1, t["strand"], i, i))


def _fastq_block(prefix, ids, seq, qual_char=b"I"):
    """Fixed-width FASTQ records as one 2-D uint8 array (fast to write)."""
    n, L = seq.shape
    name = np.char.add(prefix, np.char.zfill(ids.astype("U12"), 9)).astype("S")
    nw = name.dtype.itemsize
    rec = np.empty((n, 1 + nw + 1 + L + 3 + L + 1), dtype=np.uint8)
    rec[:, 0] = ord("@")
    rec[:, 1:1 + nw] = np.frombuffer(name.tobytes(), dtype=np.uint8).reshape(n, nw)
    o = 1 + nw
    rec[:, o] = 10
    rec[:, o + 1:o + 1 + L] = seq
    o += 1 + L
    rec[:, o] = 10
    rec[:, o + 1] = ord("+")
    rec[:, o + 2] = 10
    rec[:, o + 3:o + 3 + L] = qual_char[0]
    rec[:, o + 3 + L] = 10
    return rec


def make_reads(rng, seqs, trs, n_reads, read_len, paired, frac_spliced=0.7, sub_rate=0.01,
               n_rate=0.001, frag=(200, 500), indel_rate=0.0, chim_rate=0.0):
    """Sample reads (or pairs).  Spliced reads come from the concatenated transcript
    sequences ("transcriptome"), genomic reads from the chromosomes.  Returns
    (mate1[n,L], mate2[n,L] or None) as uint8 ASCII matrices.  Mate 2 is the reverse
    complement of the fragment's other end (standard FR library)."""
    L = read_len
    # transcriptome: concatenation of exon sequences of each transcript, separated by a
    # marker so that fragments never cross transcripts.
    pieces, tstart, tlen = [], [], []
    pos = 0
    for t in trs:
        s = np.concatenate([seqs[t["chr"]][a:b + 1] for a, b in t["exons"]])
        if t["strand"] == "-" and rng.random() < 0.5:
            pass  # strand of sampling is randomised below anyway
        pieces.append(s)
        tstart.append(pos)
        tlen.append(len(s))
        pos += len(s)
    fmin = L if not paired else max(frag[0], L)
    usable = [i for i, n in enumerate(tlen) if n >= fmin]
    n_spl = int(n_reads * frac_spliced) if usable else 0
    n_gen = n_reads - n_spl
    m1 = np.empty((n_reads, L), dtype=np.uint8)
    m2 = np.empty((n_reads, L), dtype=np.uint8) if paired else None

    def sample(src, starts, flens, out_lo):
        n = len(starts)
        idx = np.arange(L)
        a = src[starts[:, None] + idx]
        if paired:
            b = src[(starts + flens - L)[:, None] + idx]
            b = _COMP[b[:, ::-1]]
        flip = rng.random(n) < 0.5
        if paired:
            # flipping the fragment strand swaps the mates
            a2 = np.where(flip[:, None], b, a)
            b2 = np.where(flip[:, None], a, b)
            m1[out_lo:out_lo + n] = a2
            m2[out_lo:out_lo + n] = b2
        else:
            ar = _COMP[a[:, ::-1]]
            m1[out_lo:out_lo + n] = np.where(flip[:, None], ar, a)

    if n_spl:
        tr_seq = np.concatenate(pieces)
        tl = np.asarray(tlen)
        ts = np.asarray(tstart)
        us = np.asarray(usable)
        w = tl[us].astype(np.float64)
        pick = us[rng.choice(len(us), size=n_spl, p=w / w.sum())]
        if paired:
            fl = rng.integers(frag[0], frag[1] + 1, size=n_spl)
            fl = np.minimum(np.maximum(fl, L), tl[pick])
        else:
            fl = np.full(n_spl, L)
        off = (rng.random(n_spl) * (tl[pick] - fl + 1)).astype(np.int64)
        sample(tr_seq, ts[pick] + off, fl, 0)
    if n_gen:
        glen = np.asarray([len(s) for s in seqs])
        gstart = np.concatenate([[0], np.cumsum(glen)[:-1]])
        gseq = np.concatenate(seqs)
        if paired:
            fl = rng.integers(max(frag[0], L), frag[1] + 1, size=n_gen)
        else:
            fl = np.full(n_gen, L)
        ok = np.flatnonzero(glen > frag[1] + 2 if paired else glen > L + 2)
        c = ok[rng.choice(len(ok), size=n_gen, p=glen[ok] / glen[ok].sum())]
        off = (rng.random(n_gen) * (glen[c] - fl)).astype(np.int64)
        sample(gseq, gstart[c] + off, fl, n_spl)

    for m in ([m1, m2] if paired else [m1]):
        if sub_rate > 0:
            mask = rng.random(m.shape) < sub_rate
            # substitute with a DIFFERENT base where possible
            cur = m[mask]
            code = np.searchsorted(_ACGT, cur) % 4  # ACGT sorted ascending in ASCII
            newc = (code + rng.integers(1, 4, size=cur.shape[0])) % 4
            m[mask] = _ACGT[newc]
        if n_rate > 0:
            mask = rng.random(m.shape) < n_rate
            m[mask] = ord("N")
        if indel_rate > 0:
            # single-base deletion or insertion inside the read (kept at length L)
            rows = np.flatnonzero(rng.random(m.shape[0]) < indel_rate)
            for r in rows:
                p = int(rng.integers(10, L - 10))
                k = int(rng.integers(1, 4))
                if rng.random() < 0.5:   # deletion of k read bases (=insertion in genome)
                    m[r, p:L - k] = m[r, p + k:L].copy()
                    m[r, L - k:] = _ACGT[rng.integers(0, 4, size=k)]
                else:                    # insertion of k random bases
                    m[r, p + k:] = m[r, p:L - k].copy()
                    m[r, p:p + k] = _ACGT[rng.integers(0, 4, size=k)]
    # shuffle so spliced / genomic reads interleave
    perm = rng.permutation(n_reads)
    m1 = m1[perm]
    if paired:
        m2 = m2[perm]
    if chim_rate > 0:
        # chimeric pairs: mates from different loci, or a read whose two halves come from different
        # loci.  Own generator: data sets without chimeras keep their random stream.
        rng2 = np.random.default_rng(977)
        rows = np.flatnonzero(rng2.random(n_reads) < chim_rate)
        other = rng2.integers(0, n_reads, size=rows.shape[0])
        for r, o in zip(rows, other):
            if paired and rng2.random() < 0.5:
                m2[r] = m2[o].copy()
            else:
                cut = int(rng2.integers(L // 4, 3 * L // 4))
                m1[r, cut:] = m1[o, cut:].copy()
    return m1, m2


def write_fastq(prefix, m1, m2=None):
    n = m1.shape[0]
    ids = np.arange(n)
    paths = [prefix + "_1.fq"]
    _fastq_block("r", ids, m1).tofile(paths[0])
    if m2 is not None:
        paths.append(prefix + "_2.fq")
        _fastq_block("r", ids, m2).tofile(paths[1])
    return paths


def make_dataset(outdir, seed=1, chr_lengths=(300000, 200000, 150000), n_tr=120, n_reads=4000,
                 read_len=101, paired=True, annotated_frac=0.6, repeat_families=((300, 40, 0.05), (60, 30, 0.0)),
                 n_runs=20, **read_kw):
    """One self-contained data set; returns a dict of paths and sizes."""
    os.makedirs(outdir, exist_ok=True)
    rng = np.random.default_rng(seed)
    names = ["chr%d" % (i + 1) for i in range(len(chr_lengths))]
    seqs = make_genome(rng, chr_lengths, repeat_families, n_runs)
    trs = make_transcripts(rng, seqs, n_tr)
    fa = os.path.join(outdir, "genome.fa")
    _write_fasta(fa, names, seqs)
    gtf = os.path.join(outdir, "annot.gtf")
    write_gtf(gtf, names, trs, rng.random(len(trs)) < annotated_frac)
    m1, m2 = make_reads(rng, seqs, trs, n_reads, read_len, paired, **read_kw)
    fq = write_fastq(os.path.join(outdir, "reads"), m1, m2)
    return {"fasta": fa, "gtf": gtf, "fastq": fq, "n_reads": n_reads, "read_len": read_len,
            "paired": paired, "genome_bases": int(sum(chr_lengths)), "n_transcripts": len(trs)}


# ---------------------------------------------------------------------------------------------------------------------
# bench-size genomes: vectorised end to end, seconds per gigabase

HUMAN_LIKE_FAMILIES = (
    # (consensus length, copies per Mb, divergence range, minimum fraction of the consensus kept (5' truncation))
    (300, 350, (0.08, 0.16), 0.8),        # Alu-like: ~10 % of the genome
    (6000, 130, (0.02, 0.20), 0.05),      # L1-like, mostly truncated: ~17 %
    (250, 300, (0.20, 0.30), 0.5),        # MIR-like, old: ~5 %
    (2500, 18, (0.05, 0.20), 0.3), (1800, 22, (0.05, 0.25), 0.3), (900, 40, (0.04, 0.18), 0.4), (3500, 10, (0.03, 0.15), 0.3),
    (1200, 30, (0.06, 0.22), 0.4), (5000, 6, (0.01, 0.10), 0.2), (700, 45, (0.10, 0.25), 0.5),   # DNA / LTR-like families: ~12 %
    (60, 50, (0.0, 0.0), 1.0),            # short exact repeats
)


def make_genome_large(seed, total_mb, n_chr, families=HUMAN_LIKE_FAMILIES, segdup_per_mb=0.02, n_runs_per_mb=2, microsat_per_mb=30, device=None):
    """Random ACGT genome of total_mb megabases in n_chr equal chromosomes with interspersed repeat families (about half of
    the bases), a few recent segmental duplications (10-40 kb, ~1 % divergence), microsatellites and short N runs.
    Generated with torch on `device` (the GPU when there is one: a 3.1 Gb genome takes seconds; the random stream depends on the
    device type), one chromosome at a time (tensors stay far below 2^31 elements); the family consensus sequences are shared by all
    chromosomes.  Returns (list of per-chromosome uint8 ASCII numpy arrays (views into one buffer), bases written by repeats / n)."""
    import torch
    dev = torch.device(device if device is not None else ("cuda" if torch.cuda.is_available() else "cpu"))
    gen = torch.Generator(device=dev); gen.manual_seed(int(seed))
    n = int(total_mb) * 1000000
    ri = lambda lo, hi, size, dt=torch.int64: torch.randint(lo, hi, (int(size),), generator=gen, device=dev, dtype=dt)
    ru = lambda size: torch.rand(int(size), generator=gen, device=dev)
    acgt = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
    units = [ri(0, 4, f[0], torch.uint8) for f in families]
    out = np.empty(n, dtype=np.uint8)
    clen = n // n_chr
    bounds = [(i * clen, (i + 1) * clen if i < n_chr - 1 else n) for i in range(n_chr)]
    covered = 0

    def flat(length):                        # per-element copy id and offset inside the copy
        tot = int(length.sum()); start = torch.cumsum(length, 0) - length
        cid = torch.repeat_interleave(torch.arange(length.numel(), device=dev), length)
        return tot, cid, torch.arange(tot, device=dev) - start[cid]

    for c0, c1 in bounds:
        m = c1 - c0; mbc = m / 1e6
        g = ri(0, 4, m, torch.uint8)
        for (unit_len, per_mb, (d0, d1), min_frac), unit in zip(families, units):
            copies = int(per_mb * mbc)
            if copies == 0 or m <= unit_len + 2:
                continue
            step = max(1, min(copies, (1 << 27) // unit_len))
            for lo in range(0, copies, step):
                k = min(step, copies - lo)
                length = (unit_len * (min_frac + (1 - min_frac) * ru(k) ** 2)).to(torch.int64).clamp(20, unit_len)   # 3' end kept, 5' truncated
                pos = ri(0, m - unit_len - 1, k)
                div = d0 + (d1 - d0) * ru(k)
                rev = ru(k) < 0.5
                tot, cid, off = flat(length)
                base = unit[unit_len - length[cid] + off]
                mut = ru(tot) < div[cid]
                base = torch.where(mut, (base + ri(1, 4, tot, torch.uint8)) & 3, base)
                r = rev[cid]
                base = torch.where(r, 3 - base, base)
                dst = torch.where(r, pos[cid] + (length[cid] - 1 - off), pos[cid] + off)
                g[dst] = base
                covered += tot
        for _ in range(int(segdup_per_mb * mbc) + 1):         # recent segmental duplications (inside the chromosome)
            ln = int(ri(10000, 40000, 1).item())
            if m <= 3 * ln:
                break
            a, b = int(ri(0, m - ln, 1).item()), int(ri(0, m - ln, 1).item())
            cp = g[a:a + ln].clone()
            mm = ru(ln) < 0.01
            g[b:b + ln] = torch.where(mm, (cp + ri(1, 4, ln, torch.uint8)) & 3, cp)
            covered += ln
        k = int(microsat_per_mb * mbc)
        if k and m > 400:
            for motif_len in (1, 2, 3, 4):
                kk = max(1, k // 4)
                pos = ri(0, m - 200, kk); ln = ri(15, 60, kk)
                motif = ri(0, 4, kk * motif_len, torch.uint8).reshape(kk, motif_len)
                tot, cid, off = flat(ln)
                g[pos[cid] + off] = motif[cid, off % motif_len]
        g = acgt[g.long()]
        kn = int(n_runs_per_mb * mbc)
        if kn and m > 100:
            pos = ri(0, m - 40, kn); ln = ri(1, 30, kn)
            tot, cid, off = flat(ln)
            g[pos[cid] + off] = ord("N")
        out[c0:c1] = g.cpu().numpy()
        del g
    seqs = [out[a:b] for a, b in bounds]
    return seqs, min(1.0, covered / n)


class ReadSampler:
    """Bench-size read sampling: the transcriptome / genome concatenations are built once, then any number of chunks of pairs is
    drawn from them (each chunk from its own seed, so chunks can be made by different processes and always come out the same).
    Same model as make_reads: a fraction of the fragments comes from spliced transcripts, the rest from the chromosomes; FR pairs;
    substitutions and Ns per base."""

    def __init__(self, seqs, trs, read_len, frag=(200, 500)):
        self.L = read_len; self.frag = frag
        self.glen = np.asarray([len(s) for s in seqs], dtype=np.int64)
        base = seqs[0].base if seqs[0].base is not None else None
        contiguous = base is not None and all(s.base is base for s in seqs) and base.size == int(self.glen.sum())
        self.gseq = base if contiguous else np.concatenate(seqs)
        self.gstart = np.concatenate([[0], np.cumsum(self.glen)[:-1]])
        fmin = max(frag[0], read_len)
        # transcriptome as an index map into gseq (no copy of the sequence): per transcript the exon pieces
        ex_g, ex_len, tlen = [], [], []
        for t in trs:
            n = 0
            for a, b in t["exons"]:
                ex_g.append(self.gstart[t["chr"]] + a); ex_len.append(b - a + 1); n += b - a + 1
            tlen.append(n)
        ex_g = np.asarray(ex_g, dtype=np.int64); ex_len = np.asarray(ex_len, dtype=np.int64)
        tot = int(ex_len.sum())
        if tot:
            start = np.cumsum(ex_len) - ex_len
            eid = np.repeat(np.arange(len(ex_len)), ex_len)
            self.tr_seq = self.gseq[ex_g[eid] + (np.arange(tot) - start[eid])]
        else:
            self.tr_seq = np.zeros(0, dtype=np.uint8)
        self.tlen = np.asarray(tlen, dtype=np.int64)
        self.tstart = np.cumsum(self.tlen) - self.tlen if len(tlen) else np.zeros(0, dtype=np.int64)
        self.usable = np.flatnonzero(self.tlen >= fmin)
        w = self.tlen[self.usable].astype(np.float64)
        self.tcum = np.cumsum(w / w.sum()) if len(w) else None
        ok = np.flatnonzero(self.glen > frag[1] + 2)
        self.gok = ok; self.gcum = np.cumsum(self.glen[ok] / self.glen[ok].sum())

    def sample(self, seed, n, frac_spliced=0.85, sub_rate=0.01, n_rate=0.001, chim_rate=0.0):
        rng = np.random.default_rng(seed)
        L = self.L
        n_spl = int(n * frac_spliced) if len(self.usable) else 0
        idx = np.arange(L)
        m1 = np.empty((n, L), dtype=np.uint8); m2 = np.empty((n, L), dtype=np.uint8)

        def put(src, starts, flens, lo):
            k = len(starts)
            a = src[starts[:, None] + idx]
            b = _COMP[src[(starts + flens - L)[:, None] + idx][:, ::-1]]
            flip = rng.random(k) < 0.5
            m1[lo:lo + k] = np.where(flip[:, None], b, a)
            m2[lo:lo + k] = np.where(flip[:, None], a, b)
        if n_spl:
            pick = self.usable[np.minimum(np.searchsorted(self.tcum, rng.random(n_spl)), len(self.usable) - 1)]
            fl = np.minimum(np.maximum(rng.integers(self.frag[0], self.frag[1] + 1, size=n_spl), L), self.tlen[pick])
            off = (rng.random(n_spl) * (self.tlen[pick] - fl + 1)).astype(np.int64)
            put(self.tr_seq, self.tstart[pick] + off, fl, 0)
        n_gen = n - n_spl
        if n_gen:
            c = self.gok[np.minimum(np.searchsorted(self.gcum, rng.random(n_gen)), len(self.gok) - 1)]
            fl = rng.integers(max(self.frag[0], L), self.frag[1] + 1, size=n_gen)
            off = (rng.random(n_gen) * (self.glen[c] - fl)).astype(np.int64)
            put(self.gseq, self.gstart[c] + off, fl, n_spl)
        if chim_rate > 0 and n > 4:
            # chimeric pairs: half of them get the second mate of another pair (mates from two loci), the other half a
            # first mate whose second half comes from another read (a chimeric junction inside the mate)
            k = int(n * chim_rate); who = rng.choice(n, size=k, replace=False); other = rng.integers(0, n, size=k)
            h = k // 2
            m2[who[:h]] = m2[other[:h]].copy()
            m1[who[h:], L // 2:] = m1[other[h:], L // 2:].copy()
        for m in (m1, m2):
            if sub_rate > 0:
                mask = rng.random(m.shape, dtype=np.float32) < sub_rate
                cur = m[mask]
                code = np.searchsorted(_ACGT, cur) % 4
                m[mask] = _ACGT[(code + rng.integers(1, 4, size=cur.shape[0])) % 4]
            if n_rate > 0:
                m[rng.random(m.shape, dtype=np.float32) < n_rate] = ord("N")
        perm = rng.permutation(n)
        return m1[perm], m2[perm]


def write_fastq_ids(prefix, m1, m2, first_id):
    """FASTQ pair files with read names r<first_id + i> (fixed width)."""
    ids = np.arange(first_id, first_id + m1.shape[0])
    _fastq_block("r", ids, m1).tofile(prefix + "_1.fq")
    _fastq_block("r", ids, m2).tofile(prefix + "_2.fq")
