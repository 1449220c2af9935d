#!/bin/bash
# Data-race check of the pipelined front end on a machine without a GPU: host library + star_amd CLI built with ThreadSanitizer, the oracle behind the engine's C ABI
# (oracle/cli_shim.cpp), one 2-pass + BySJout run with two "devices" x two contexts, input read in slices, the junction table collapsed in the background after every
# batch.  usage: tests/tools/tsan_cli.sh [data set of tests/util.py, default pe101]      prints the number of ThreadSanitizer reports (0 = clean)
set -e
cd "$(dirname "$0")/../.."
DS=${1:-pe101}; O=$(mktemp -d /tmp/staramd_tsan.XXXXXX)
g++ -O1 -g -std=c++17 -fPIC -pthread -fsanitize=thread -shared $(ls star_amd/csrc/host/*.cpp | grep -v "main.cpp\|cli_run.cpp") -o $O/libstaramd_host.so -lz
g++ -O1 -g -std=c++17 -fPIE -pthread -fsanitize=thread -DSTARAMD_NO_RESIDENT_SJDB star_amd/csrc/host/main.cpp star_amd/csrc/host/cli_run.cpp oracle/cli_shim.cpp -o $O/star_amd_oracle_cli \
    -L$O -lstaramd_host -Loracle/_build -loracle -lindex_emul -Wl,-rpath,$O -Wl,-rpath,$PWD/oracle/_build
python - "$DS" "$O" <<'PY'
import os, subprocess, sys
sys.path.insert(0, "tests")
from util import prepare
ds, out = sys.argv[1], sys.argv[2]
info = prepare(ds, out, need_ref=False)
cmd = [out + "/star_amd_oracle_cli", "--runMode", "alignReads", "--genomeDir", info["idx"], "--readFilesIn"] + info["fastq"] + ["--outFileNamePrefix", out + "/run_", "--runThreadN", "6",
       "--gpuBatchReads", "300", "--twopassMode", "Basic", "--outFilterType", "BySJout", "--gpuDevices", "0,1"] + list(info["extra"]) + ([] if "--outSAMunmapped" in info["extra"] else ["--outSAMunmapped", "Within"])
env = dict(os.environ, STARAMD_SJ_KICK="100", STARAMD_READ_SLICE_MIN="1", STARAMD_SJDB_HOST="1", TSAN_OPTIONS="halt_on_error=0 report_signal_unsafe=0")
p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=3000)
n = p.stderr.count("WARNING: ThreadSanitizer")
print("exit code %d, ThreadSanitizer reports: %d" % (p.returncode, n))
if n:
    print(p.stderr[:4000])
sys.exit(1 if (n or p.returncode) else 0)
PY
rm -rf $O
