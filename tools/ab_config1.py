#!/usr/bin/env python3
"""tools/ab_kernels.py on the config-1 stand-in of bench.py (12 Mb genome, single-end 1x50 reads):  tools/ab_config1.py [ab_kernels options] "tag|lib|ENV=V" ..."""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench, ab_kernels
wd = "/dev/shm/star_amd_bench" if os.path.isdir("/dev/shm") else "/tmp/star_amd_bench"
args = argparse.Namespace(read_len=50, reads=400000, workdir=wd)
g, _ = bench.build_genome(args, 12, lambda s: print("ab_config1:", s, file=sys.stderr))
n = 3 * args.reads
fq = bench.make_reads(args, g, os.path.join(g, "abse_n%d" % n), "se", n, 8200, read_len=50)[:1]
sys.argv = [sys.argv[0], "--genome-dir", os.path.join(g, "idx"), "--fastq", fq[0], "--reads", str(args.reads)] + sys.argv[1:]
ab_kernels.main()
