#!/usr/bin/env python3
"""Junction insertion at scale: the same genome + GTF through `star_amd --runMode genomeGenerate` twice, junctions inserted by the host
restatement (STARAMD_SJDB_HOST=1) and on the device, and the resulting SA / SAindex / Genome compared byte for byte.
    python tools/sjdb_scale_check.py --mb 3100"""
import argparse, json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def first_diff(a, b, chunk=1 << 28):
    import numpy as np
    sa, sb = os.path.getsize(a), os.path.getsize(b)
    if sa != sb:
        return {"sizes": [sa, sb]}
    off = 0
    with open(a, "rb") as fa, open(b, "rb") as fb:
        while True:
            x, y = fa.read(chunk), fb.read(chunk)
            if not x:
                return None
            if x != y:
                d = np.flatnonzero(np.frombuffer(x, dtype=np.uint8) != np.frombuffer(y, dtype=np.uint8))
                return {"first_byte": off + int(d[0]), "differing_bytes_in_chunk": int(len(d))}
            off += len(x)


def main():
    ap = argparse.ArgumentParser(); ap.add_argument("--mb", type=int, default=1000); ap.add_argument("--dir", default="/dev/shm/sjdbcheck")
    a = ap.parse_args()
    import math
    import numpy as np
    from star_amd import synth
    d = os.path.join(a.dir, "g%d" % a.mb); os.makedirs(d, exist_ok=True)
    nchr = max(1, min(24, a.mb // 40))
    seqs, _ = synth.make_genome_large(20260922, a.mb, nchr)
    rng = np.random.default_rng(20260923)
    trs = synth.make_transcripts(rng, seqs, 65 * a.mb)
    names = ["chr%d" % (i + 1) for i in range(nchr)]
    synth._write_fasta(os.path.join(d, "genome.fa"), names, seqs)
    synth.write_gtf(os.path.join(d, "annot.gtf"), names, trs, rng.random(len(trs)) < 0.7)
    del seqs
    nb = max(4, min(14, int(math.log2(a.mb * 1e6) / 2 - 1)))
    out = {"mb": a.mb}
    for tag, env in (("host", {"STARAMD_SJDB_HOST": "1"}), ("device", {})):
        o = os.path.join(d, tag); os.makedirs(o, exist_ok=True)
        cmd = [os.path.join(ROOT, "star_amd", "bin", "star_amd"), "--runMode", "genomeGenerate", "--genomeDir", o, "--genomeFastaFiles", os.path.join(d, "genome.fa"),
               "--genomeSAindexNbases", str(nb), "--sjdbGTFfile", os.path.join(d, "annot.gtf"), "--sjdbOverhang", "100", "--runThreadN", str(os.cpu_count()), "--outFileNamePrefix", o + "/_log_"]
        t = time.time(); p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=dict(os.environ, STARAMD_HOST_TIMING="1", **env)); out[tag + "_s"] = time.time() - t
        out[tag + "_rc"] = p.returncode; out[tag + "_log"] = p.stderr.strip().splitlines()[-6:]
    for f in ("Genome", "SA", "SAindex", "sjdbInfo.txt"):
        out["diff_" + f] = first_diff(os.path.join(d, "host", f), os.path.join(d, "device", f))
    # first entries of the two SAindex tables, decoded
    def entries(path, nb, bits, k=24):
        import numpy as np
        raw = np.fromfile(path, dtype=np.uint8, count=8 * (nb + 2) + (k * bits + 7) // 8 + 16)[8 * (nb + 2):]
        val = int.from_bytes(raw.tobytes(), "little")
        return [(val >> (i * bits)) & ((1 << bits) - 1) for i in range(k)]
    gsb = None
    for l in open(os.path.join(d, "host", "genomeParameters.txt")):
        if l.startswith("### GstrandBit"): gsb = int(l.split()[2])
    out["GstrandBit"] = gsb
    def level_diff(nb, bits):          # per prefix length: entries that differ (decoded from the packed tables, vectorised)
        import numpy as np
        A = np.fromfile(os.path.join(d, "host", "SAindex"), dtype=np.uint8)[8 * (nb + 2):]
        B = np.fromfile(os.path.join(d, "device", "SAindex"), dtype=np.uint8)[8 * (nb + 2):]
        res = {}; start = 0
        for L in range(1, nb + 1):
            n = 4 ** L
            idx = np.arange(start, start + min(n, 2000000), dtype=np.uint64)
            def dec(X):
                b = idx * np.uint64(bits); by = (b >> np.uint64(3)).astype(np.int64); sh = (b & np.uint64(7))
                w = np.zeros(len(idx), dtype=np.uint64)
                for k in range(8):
                    w |= X[by + k].astype(np.uint64) << np.uint64(8 * k)
                return (w >> sh) & np.uint64((1 << bits) - 1)
            a, b2 = dec(A), dec(B)
            bad = np.flatnonzero(a != b2)
            res[str(L)] = {"checked": int(len(idx)), "differ": int(len(bad)), "first": [[int(i), int(a[i]), int(b2[i])] for i in bad[:4]]}
            start += n
        return res
    try:
        out["level_diff"] = level_diff(nb, gsb + 3)
    except Exception as e:
        out["level_diff"] = repr(e)
    out["sai_host_first"] = entries(os.path.join(d, "host", "SAindex"), nb, gsb + 3)
    out["sai_device_first"] = entries(os.path.join(d, "device", "SAindex"), nb, gsb + 3)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
