"""N>1 path on CPU: two ranks (gloo, world_size 2) each map their own shard of the reads, then the end-of-run
junction/Stats exchange of star_amd/multi_gpu.py runs and rank 0 writes SJ.out.tab + Log.final.out.
The union must be byte-identical to ONE reference run over all reads (SJ.out.tab, Log.final.out counters, sorted SAM).
The per-rank mapping uses the CPU oracle here (tests only: there is no GPU in this tier); on the GPU box the same
exchange is driven by bench.py with the HIP engine."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from util import DATASETS, ROOT, capi, oracle_lib, prepare, refstar

pytestmark = pytest.mark.skipif(not refstar.have_ref(), reason="oracle/_ref/STAR not built")


def _split_fastq(paths, world, outdir):
    shards = [[] for _ in range(world)]
    for im, p in enumerate(paths):
        lines = open(p).read().split("\n")
        if lines and lines[-1] == "":
            lines.pop()
        n = len(lines) // 4
        per = (n + world - 1) // world
        for r in range(world):
            q = os.path.join(outdir, "shard%d_%d.fq" % (r, im + 1))
            with open(q, "w") as f:
                chunk = lines[4 * r * per: 4 * min(n, (r + 1) * per)]
                f.write("\n".join(chunk) + ("\n" if chunk else ""))
            shards[r].append(q)
    return shards


def _free_port():
    """a port nobody listens on right now (a fixed port per process id collides now and then when test workers run side by side)"""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, idx, shards, extra, outdir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from star_amd import multi_gpu
    prefix = os.path.join(outdir, "r%d_" % rank)
    run = capi.HostRun(["--genomeDir", idx, "--readFilesIn"] + shards[rank] + ["--outFileNamePrefix", prefix] + list(extra))
    eng = oracle_lib.Oracle(run.genome, run.params)
    while True:
        while True:
            b = run.next_batch(500)
            if b is None:
                break
            bufs = capi.ResultBuffers(b.nReads, tr_cap=b.nReads * 64)
            eng.map_batch(b, bufs)
            run.emit(bufs.res)
        phase = multi_gpu.next_phase(run, dist, torch.device("cpu"), rank, world)     # exchanges what the next phase needs
        if phase == 0:
            break
        if phase == 1:
            eng.update_index(run.genome, run.params)
        else:
            eng.set_novel_junctions(*run.novel_junctions())
    multi_gpu.merge_run_outputs(run, dist, torch.device("cpu"), rank, world)
    run.finish()            # rank 0 holds the union; other ranks write their partial files (ignored)
    eng.close(); run.close()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("name", ["pe101", "se50"])
def test_two_ranks_match_single_reference_run(name, tmp_path, built):
    info = prepare(name, str(tmp_path))
    world = 2
    shards = _split_fastq(info["fastq"], world, str(tmp_path))
    port = _free_port()
    mp.spawn(_worker, args=(world, port, info["idx"], shards, info["extra"], str(tmp_path)), nprocs=world, join=True)
    ref = info["ref_prefix"]
    r0 = os.path.join(str(tmp_path), "r0_")
    assert open(ref + "SJ.out.tab", "rb").read() == open(r0 + "SJ.out.tab", "rb").read()
    assert refstar.final_log_counters(ref + "Log.final.out") == refstar.final_log_counters(r0 + "Log.final.out")
    union = []
    for r in range(world):
        union += refstar.sam_body_sorted(os.path.join(str(tmp_path), "r%d_Aligned.out.sam" % r))
    assert sorted(union) == refstar.sam_body_sorted(ref + "Aligned.out.sam")


def test_two_ranks_two_pass_by_sjout_gene_counts(tmp_path, built):
    """phases across ranks: the junctions of pass 1 and of BySJout stage 1 are exchanged before every rank inserts / filters with the
    union; gene counts are summed on rank 0.  One reference run over all reads is the truth."""
    info = prepare("pe101", str(tmp_path), need_ref=False)
    flags = ["--twopassMode", "Basic", "--outFilterType", "BySJout", "--quantMode", "GeneCounts"]
    ref = refstar.align(info["idx"], info["fastq"], os.path.join(str(tmp_path), "ref_"), threads=1, extra=flags)
    world = 2
    shards = _split_fastq(info["fastq"], world, str(tmp_path))
    port = _free_port()
    mp.spawn(_worker, args=(world, port, info["idx"], shards, flags, str(tmp_path)), nprocs=world, join=True)
    r0 = os.path.join(str(tmp_path), "r0_")
    assert open(ref + "SJ.out.tab", "rb").read() == open(r0 + "SJ.out.tab", "rb").read()
    assert open(ref + "ReadsPerGene.out.tab", "rb").read() == open(r0 + "ReadsPerGene.out.tab", "rb").read()
    assert refstar.final_log_counters(ref + "Log.final.out") == refstar.final_log_counters(r0 + "Log.final.out")
    assert open(ref + "_STARpass1/SJ.out.tab", "rb").read() == open(r0 + "_STARpass1/SJ.out.tab", "rb").read()
    union = []
    for r in range(world):
        union += refstar.sam_body_sorted(os.path.join(str(tmp_path), "r%d_Aligned.out.sam" % r))
    assert sorted(union) == refstar.sam_body_sorted(ref + "Aligned.out.sam")


# ---- the same through the in-process front end (include/star_amd_cli.h), as bench.py drives it: the whole pipeline of a rank is ONE call, the
# cross-rank exchanges happen inside its `exchange` hook (star_amd/multi_gpu.py exchange_before_phase / merge_handle_outputs) ----

CLI_ORACLE_LIB = os.path.join(ROOT, "oracle", "_build", "libstaramd_cli_oracle.so")


def _cli_worker(rank, world, port, idx, shards, extra, outdir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from star_amd import multi_gpu
    dev = torch.device("cpu")
    prefix = os.path.join(outdir, "c%d_" % rank)
    marks = []

    def exchange(h, last):
        if last:
            multi_gpu.merge_handle_outputs(capi.host_lib(), h, dist, dev, rank, world)
        else:
            multi_gpu.exchange_before_phase(capi.host_lib(), h, dist, dev, rank, world)
        marks.append(last)
        return 0

    argv = ["--runMode", "alignReads", "--genomeDir", idx, "--readFilesIn"] + shards[rank] + ["--outFileNamePrefix", prefix, "--runThreadN", "2", "--gpuBatchReads", "400",
            "--benchWarmupReads", "400"] + list(extra)
    rc, rep = capi.run_cli(argv, warmup_done=lambda: dist.barrier(), exchange=exchange, lib_path=CLI_ORACLE_LIB)
    assert rc == 0 and marks and marks[-1] == 1 and rep.timedReads > 0 and rep.reads >= rep.timedReads
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.skipif(not os.path.isfile(CLI_ORACLE_LIB), reason="oracle-backed front-end library not built")
@pytest.mark.parametrize("flags", [[], ["--twopassMode", "Basic", "--outFilterType", "BySJout", "--quantMode", "GeneCounts"]])
def test_two_ranks_through_the_front_end_hooks(flags, tmp_path, built):
    info = prepare("pe101", str(tmp_path), need_ref=False)
    ref = refstar.align(info["idx"], info["fastq"], os.path.join(str(tmp_path), "ref_"), threads=1, extra=flags)
    world = 2
    shards = _split_fastq(info["fastq"], world, str(tmp_path))
    port = _free_port()
    mp.spawn(_cli_worker, args=(world, port, info["idx"], shards, flags, str(tmp_path)), nprocs=world, join=True)
    c0 = os.path.join(str(tmp_path), "c0_")
    assert open(ref + "SJ.out.tab", "rb").read() == open(c0 + "SJ.out.tab", "rb").read()
    assert refstar.final_log_counters(ref + "Log.final.out") == refstar.final_log_counters(c0 + "Log.final.out")
    if flags:
        assert open(ref + "ReadsPerGene.out.tab", "rb").read() == open(c0 + "ReadsPerGene.out.tab", "rb").read()
    union = []
    for r in range(world):
        union += refstar.sam_body_sorted(os.path.join(str(tmp_path), "c%d_Aligned.out.sam" % r))
    assert sorted(union) == refstar.sam_body_sorted(ref + "Aligned.out.sam")
