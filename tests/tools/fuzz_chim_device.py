#!/usr/bin/env python3
"""GPU side of the bug hunt for staramd_params::resultSelect 2 (needs an MI355X): random data sets with chimeric fragments and random chimeric / hot-path flags through the
shipped binary, the partner of chimeric detection chosen on the device (default) against the same run with STARAMD_CHIM_ON_DEVICE=0 (the host loop over every transcript of
every window, which tests/test_chimeric.py pins against the reference): Chimeric.out.junction, the SAM records, SJ.out.tab and the Log counters must be the same.
usage (on the GPU box): python tests/tools/fuzz_chim_device.py [iterations] [seed]"""
import os
import random
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
from util import refstar            # noqa: E402
from star_amd import synth          # noqa: E402

CLI = os.path.join(ROOT, "star_amd", "bin", "star_amd")
CHIM = [["--chimJunctionOverhangMin", "8"], ["--chimJunctionOverhangMin", "25"], ["--chimScoreMin", "10"], ["--chimScoreDropMax", "60"], ["--chimScoreDropMax", "5"], ["--chimScoreSeparation", "1"],
        ["--chimScoreSeparation", "25"], ["--chimScoreJunctionNonGTAG", "-6"], ["--chimScoreJunctionNonGTAG", "0"], ["--chimSegmentReadGapMax", "5"], ["--chimSegmentReadGapMax", "40"],
        ["--chimMainSegmentMultNmax", "1"], ["--chimMainSegmentMultNmax", "3"], ["--chimFilter", "None"], ["--chimOutJunctionFormat", "1"]]
HOT = [["--outFilterMultimapScoreRange", "4"], ["--outFilterMultimapNmax", "2"], ["--alignIntronMax", "3000"], ["--scoreGap", "-2"], ["--seedSearchStartLmax", "25"], ["--winAnchorMultimapNmax", "20"],
       ["--alignSJoverhangMin", "3"], ["--outFilterMismatchNmax", "3"], ["--alignTranscriptsPerReadNmax", "60", "--alignTranscriptsPerWindowNmax", "6"], ["--outSAMmultNmax", "1"],
       ["--outFilterType", "BySJout"], ["--twopassMode", "Basic"], ["--outSAMunmapped", "Within"], ["--alignEndsType", "EndToEnd"]]


def one(it, rng):
    work = tempfile.mkdtemp(prefix="fzchim%04d_" % it)
    rl = rng.choice([50, 76, 100, 125, 150])
    pe = rng.random() < 0.6
    d = os.path.join(work, "d")
    info = synth.make_dataset(d, seed=rng.randrange(1 << 30), chr_lengths=tuple(rng.randrange(80000, 300000) for _ in range(rng.randrange(2, 5))), n_tr=rng.randrange(30, 120),
                              n_reads=rng.randrange(1500, 4000), read_len=rl, paired=pe, sub_rate=rng.choice([0.0, 0.005, 0.02]), n_rate=rng.choice([0.0, 0.002]),
                              indel_rate=rng.choice([0.0, 0.002]), chim_rate=rng.choice([0.2, 0.5, 0.8]), frag=(max(rl // 2, 40), max(rl * 3, 200)))
    idx = os.path.join(d, "idx")
    refstar.genome_generate(info["fasta"], idx, gtf=info["gtf"] if rng.random() < 0.7 else None, sa_index_nbases=rng.choice([7, 8, 9]), sjdb_overhang=rl - 1)
    flags = ["--chimSegmentMin", str(rng.choice([10, 12, 15, 20, 30]))]
    for fl in rng.sample(CHIM, rng.randrange(0, 5)) + rng.sample(HOT, rng.randrange(0, 3)):
        if fl[0] not in flags:
            flags += fl
    print("run  [%d] %s%d %s" % (it, "pe" if pe else "se", rl, " ".join(flags)), flush=True)
    outs = {}
    for on in ("1", "0"):
        new = os.path.join(work, "o%s_" % on)
        p = subprocess.run([CLI, "--runMode", "alignReads", "--genomeDir", idx, "--readFilesIn"] + info["fastq"] + ["--outFileNamePrefix", new, "--runThreadN", "4", "--gpuBatchReads", str(rng.choice([300, 700, 5000]))] + flags,
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=dict(os.environ, STARAMD_CHIM_ON_DEVICE=on, STARAMD_VERBOSE="1"), timeout=600)
        if p.returncode != 0:
            print("FAIL [%d] exit code %d (STARAMD_CHIM_ON_DEVICE=%s): %s" % (it, p.returncode, on, p.stderr[-600:]), flush=True)
            return False
        dev = "partner of chimeric detection chosen on the device" in p.stderr
        if dev != (on == "1"):
            print("FAIL [%d] STARAMD_CHIM_ON_DEVICE=%s but the run says device=%s" % (it, on, dev), flush=True)
            return False
        outs[on] = ([l for l in open(new + "Chimeric.out.junction") if not l.startswith("# 2.7")], refstar.sam_body_sorted(new + "Aligned.out.sam"), open(new + "SJ.out.tab").read(),
                    refstar.final_log_counters(new + "Log.final.out"))
    same = outs["1"] == outs["0"]
    print("%s [%d] %d junction lines" % ("ok  " if same else "DIFF", it, len(outs["1"][0])), flush=True)
    if same:
        shutil.rmtree(work, ignore_errors=True)
    return same


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    bad = sum(0 if one(i, rng) else 1 for i in range(n))
    print("%d of %d combinations differ" % (bad, n))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
