"""The cross-rank exchanges of star_amd/multi_gpu.py over RCCL on hardware (-m gpu): ONE rank, backend "nccl" (= RCCL on ROCm), the shipped front end
with the HIP engine behind it.  A process group of world size 1 still runs every collective of the path through RCCL on device tensors -- the size
all_gather, the padded junction-table all_gather, the gene-count all_gather -- which the CPU suite (gloo, world 2: tests/test_multi_rank_cpu.py) cannot
do; what a second rank would add is covered there.  Flags: --twopassMode Basic + --outFilterType BySJout + --quantMode GeneCounts, so that the exchange
before a phase (junctions of pass 1, junctions of BySJout stage 1) and the one at the end of the run (table, counters, gene counts) all fire.
The outputs must be the reference's own (one run over the same reads)."""
import os
import subprocess
import sys

import pytest

from util import ROOT, prepare, refstar

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not refstar.have_ref(), reason="oracle/_ref/STAR not built")]

WORKER = r"""
import os, sys, json
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import torch, torch.distributed as dist
from star_amd import capi, multi_gpu
idx, fq, prefix, flags = sys.argv[2], json.loads(sys.argv[3]), sys.argv[4], json.loads(sys.argv[5])
os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ.setdefault("MASTER_PORT", sys.argv[6])
assert torch.cuda.is_available()
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1)
assert dist.get_backend() == "nccl"
marks = []; moved = []
def exchange(h, last):
    if last:
        moved.append(multi_gpu.merge_handle_outputs(capi.host_lib(), h, dist, dev, 0, 1))
    else:
        multi_gpu.exchange_before_phase(capi.host_lib(), h, dist, dev, 0, 1)
    marks.append(int(last))
    return 0
def barrier():
    dist.barrier(); torch.cuda.synchronize(dev)
argv = ["--runMode", "alignReads", "--genomeDir", idx, "--readFilesIn"] + fq + ["--outFileNamePrefix", prefix, "--runThreadN", "2", "--gpuBatchReads", "400", "--benchWarmupReads", "400"] + flags
rc, rep = capi.run_cli(argv, warmup_done=barrier, exchange=exchange)
torch.cuda.synchronize(dev)
dist.barrier()
dist.destroy_process_group()
maps = open("/proc/self/maps").read()
print(json.dumps({"rc": rc, "marks": marks, "bytes_all_gathered": moved, "reads": int(rep.reads), "rccl_mapped": ("librccl" in maps) or ("libnccl" in maps), "engine_mapped": "libstaramd.so" in maps,
                  "oracle_mapped": "liboracle" in maps}))
"""


def test_exchanges_run_over_rccl_with_one_rank(tmp_path, built):
    info = prepare("pe101", str(tmp_path), need_ref=False)
    flags = ["--twopassMode", "Basic", "--outFilterType", "BySJout", "--quantMode", "GeneCounts"]
    ref = refstar.align(info["idx"], info["fastq"], os.path.join(str(tmp_path), "ref_"), threads=1, extra=flags)
    new = os.path.join(str(tmp_path), "rccl_")
    import json
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, "-c", WORKER, ROOT, info["idx"], json.dumps(info["fastq"]), new, json.dumps(flags), str(port)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    out = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert out["rc"] == 0 and out["reads"] > 0
    assert out["marks"] and out["marks"][-1] == 1 and 0 in out["marks"], out          # an exchange before a phase AND the one at the end of the run
    assert out["bytes_all_gathered"] and out["bytes_all_gathered"][0] > 32 * 8, out   # a junction table + 32 counters went through all_gather
    assert out["rccl_mapped"] and out["engine_mapped"] and not out["oracle_mapped"], out
    assert open(ref + "SJ.out.tab", "rb").read() == open(new + "SJ.out.tab", "rb").read()
    assert open(ref + "ReadsPerGene.out.tab", "rb").read() == open(new + "ReadsPerGene.out.tab", "rb").read()
    assert refstar.final_log_counters(ref + "Log.final.out") == refstar.final_log_counters(new + "Log.final.out")
    assert refstar.sam_body_sorted(new + "Aligned.out.sam") == refstar.sam_body_sorted(ref + "Aligned.out.sam")
