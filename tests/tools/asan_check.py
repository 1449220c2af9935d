#!/usr/bin/env python3
"""Memory / undefined-behaviour check of the host side (CPU): main.cpp + the host library + the oracle behind the engine's C ABI (oracle/cli_shim.cpp), built into
one binary with -fsanitize=address,undefined and run over feature-rich flag sets and odd inputs (SAM text, multi-line FASTA, several files, edge reads).
usage: tests/tools/asan_check.py [thread]     (exit code 1 when a sanitizer reports anything; `thread` builds with -fsanitize=thread instead:
the reader / mapper / post-map / writer threads of the front end, the sliced input reads, the threaded junction collapse)"""
import glob
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
from util import make_edge_reads, prepare            # noqa: E402
import test_fasta_reads, test_output_options, test_sam_reads, test_wasp   # noqa: E402

work = tempfile.mkdtemp(prefix="asan_")
cli = os.path.join(work, "cli")
TSAN = len(sys.argv) > 1 and sys.argv[1] == "thread"
subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=thread" if TSAN else "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-pthread", "-mavx2", "-DSTARAMD_NO_RESIDENT_SJDB",
                       "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "star_amd/csrc")]
                      + sorted(glob.glob(os.path.join(ROOT, "star_amd/csrc/host/*.cpp")))
                      + [os.path.join(ROOT, "oracle/cli_shim.cpp"), os.path.join(ROOT, "oracle/star_oracle.cpp"), os.path.join(ROOT, "oracle/index_emul.cpp"), "-o", cli, "-lz"])
reports = 0


def run(tag, info, fq, flags):
    global reports
    d = os.path.dirname(info["fastq"][0])
    r = subprocess.run([cli, "--runMode", "alignReads", "--genomeDir", info["idx"], "--readFilesIn"] + fq + ["--outFileNamePrefix", os.path.join(d, "asan_%s_" % tag), "--runThreadN", "3", "--gpuBatchReads", "600"] + flags,
                       stderr=subprocess.PIPE, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0", TSAN_OPTIONS="halt_on_error=0 second_deadlock_stack=1"))
    bad = [l for l in r.stderr.decode().split("\n") if "ERROR: AddressSanitizer" in l or "runtime error" in l or "WARNING: ThreadSanitizer" in l]
    if bad and TSAN:
        open(os.path.join(work, "tsan_%s.log" % tag), "wb").write(r.stderr)
    print("%-10s rc %d, sanitizer reports: %d" % (tag, r.returncode, len(bad)))
    for l in bad[:8]:
        print("    " + l[:240])
    reports += len(bad) + (1 if r.returncode else 0)


a = dict(prepare("pe76_overlap", os.path.join(work, "a"), need_ref=False))
run("merge_chim", a, a["fastq"], ["--peOverlapNbasesMin", "10", "--peOverlapMMp", "0.1", "--chimSegmentMin", "10", "--chimJunctionOverhangMin", "10", "--chimMultimapNmax", "20", "--chimOutType", "WithinBAM", "Junctions",
                                  "--outSAMtype", "BAM", "Unsorted", "SortedByCoordinate", "--quantMode", "TranscriptomeSAM", "GeneCounts", "--outSAMunmapped", "Within", "KeepPairs"])
b = dict(prepare("pe101", os.path.join(work, "b"), need_ref=False)); db = os.path.dirname(b["fastq"][0])
run("clip_2pass", b, b["fastq"], ["--twopassMode", "Basic", "--outFilterType", "BySJout", "--outMultimapperOrder", "Random", "--clip3pNbases", "3", "400", "--clip3pAdapterSeq", "AGATCGGAAG", "-", "--clip3pAdapterMMp", "0.1", "0.1",
                                  "--outReadsUnmapped", "Fastx", "--outSAMunmapped", "Within"])
run("sam_in", b, test_sam_reads._to_sam(b["fastq"], db), ["--readFilesType", "SAM", "PE", "--outSAMtype", "BAM", "Unsorted", "--outFilterType", "BySJout", "--outSAMunmapped", "Within", "--outReadsUnmapped", "Fastx"])
run("fasta_in", b, test_fasta_reads._to_fasta(b["fastq"], db, 40), ["--outSAMunmapped", "Within", "--outReadsUnmapped", "Fastx", "--twopassMode", "Basic", "--outSAMtype", "BAM", "SortedByCoordinate"])
run("two_files", b, test_output_options._split(b["fastq"], db), ["--outSAMattrRGline", "ID:a", ",", "ID:b", "--outFilterType", "BySJout", "--outSAMmultNmax", "1", "--outSAMunmapped", "Within", "KeepPairs"])
run("edge_reads", b, list(make_edge_reads(b, db, paired=True)), ["--outSAMunmapped", "Within", "--chimSegmentMin", "12", "--peOverlapNbasesMin", "5", "--chimMultimapNmax", "5", "--clip5pNbases", "2", "1"])
c = dict(prepare("pe150_chim", os.path.join(work, "c"), need_ref=False)); dc = os.path.dirname(c["fastq"][0])
run("wasp_chim", c, c["fastq"], ["--chimSegmentMin", "12", "--chimOutType", "WithinBAM", "SeparateSAMold", "Junctions", "--waspOutputMode", "SAMtag", "--varVCFfile", test_wasp._vcf(c, dc),
                                 "--outSAMattributes", "NH", "HI", "AS", "nM", "vA", "vG", "rB", "cN", "MC", "NM", "MD", "--outSAMtype", "BAM", "SortedByCoordinate", "--outWigType", "bedGraph"])
print("%d problem(s); work directory %s" % (reports, work))
sys.exit(1 if reports else 0)
