"""GPU twins of the host-side features that feed EXTRA batches or unusual reads through the hot path: every case runs the shipped
binary (star_amd/bin/star_amd, HIP engine) against one run of the reference with the same flags and compares every output file.

  merged mates      --peOverlapNbasesMin: the overlapping pairs of a batch, merged into single-end reads, are a second engine batch under
                    PE parameters (ReadAlign_peOverlapMergeMap.cpp:31 re-enters mapOneRead)
  clipped reads     --clip*: the device maps the shorter reads, among them 0-length mates and mates shorter than a seed
  WASP              --waspOutputMode: allele-swapped copies re-mapped as one more batch (ReadAlign_waspMap.cpp:78), here with a batch
                    size small enough that the re-mapping batch goes through the engine in pieces
  phases            2-pass (index replaced in HBM), BySJout (whitelist), chimeric detection (every window returned)
  two contexts      --gpuDevices 0,0: two mapper threads with their own engine contexts on the one GPU of the test box
"""
import os

import pytest

import test_cli_pipeline as tcp
import test_pe_overlap as tpo
import test_clipping as tcl

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not tcp.refstar.have_ref(), reason="oracle/_ref/STAR not built")]
GPU_CLI = os.path.join(tcp.ROOT, "star_amd", "bin", "star_amd")


@pytest.mark.parametrize("name,more,batch", tcp.CASES)
def test_cli_cases_on_gpu(name, more, batch, tmp_path, built):
    tcp.run_cli_case(GPU_CLI, name, more, batch, tmp_path)


def test_uploads_run_ahead_of_their_batches(tmp_path, built):
    """staramd_prefetch_batch (include/star_amd_async.h): with one context the front end shows the engine the batch that is next in line; the run must report batches whose
    upload was done when staramd_map_batch was called for them -- the ordinary upload behind a prefetch that never matches returns the same bytes"""
    fp = tcp.run_cli_case(GPU_CLI, "pe101", [], 60, tmp_path)
    assert fp["prefetched"] >= 2 and fp["overlapped"] == 0, fp


def test_kernels_begin_beside_the_copy_of_the_results_before_them(tmp_path, built):
    """STARAMD_OVERLAP_COPIES=1: the mapper works through staramd_map_begin / _wait / _end (include/star_amd_async.h) -- same outputs, and the path really ran"""
    fp = tcp.run_cli_case(GPU_CLI, "pe101", ["--outSAMunmapped", "Within"], 60, tmp_path, env={"STARAMD_OVERLAP_COPIES": "1"})
    assert fp["overlapped"] >= 2 and fp["prefetched"] >= 2, fp


@pytest.mark.parametrize("name,more,batch,devices", [(c[0], c[1], c[2], "0,0") for c in tcp.MULTI[:3]])
def test_two_contexts_on_one_gpu(name, more, batch, devices, tmp_path, built):
    tcp.run_cli_case(GPU_CLI, name, more + ["--gpuDevices", devices], batch, tmp_path)


@pytest.mark.parametrize("name,more", tpo.CASES + tpo.CHIM)
def test_merged_mates_on_gpu(name, more, tmp_path, built):
    tcp.run_cli_case(GPU_CLI, name, more, 600, tmp_path)


@pytest.mark.parametrize("turns", ["0", "1"])
def test_two_contexts_share_one_resident_index(turns, tmp_path, built):
    """STARAMD_CONTEXTS_PER_GPU=2: the second context maps against the index of the first (staramd_create_shared), 2-pass included (junction insertion on the owner,
    the sharer follows); with STARAMD_KERNEL_TURNS=1 the kernel phases of the two contexts are taken in turns"""
    tcp.run_cli_case(GPU_CLI, "pe101", ["--twopassMode", "Basic"], 300, tmp_path, env={"STARAMD_CONTEXTS_PER_GPU": "2", "STARAMD_KERNEL_TURNS": turns})


def test_merged_chimeric_fragments_on_gpu(tmp_path, built):
    flags = ["--peOverlapNbasesMin", "10", "--peOverlapMMp", "0.1", "--chimSegmentMin", "12", "--chimJunctionOverhangMin", "10", "--chimMultimapNmax", "10", "--chimMultimapScoreRange", "3",
             "--chimNonchimScoreDropMin", "15", "--chimScoreDropMax", "80", "--chimSegmentReadGapMax", "5", "--outSAMunmapped", "Within"]
    tcp.run_cli_case(GPU_CLI, "pe101", flags, 700, tmp_path, fastq_hook=lambda info, d: tpo._chimeric_fragments(info, d))


@pytest.mark.parametrize("tag", sorted(tcl.PE))
def test_clipped_pairs_on_gpu(tag, tmp_path, built):
    """adapters at every position incl. position 0: 0-length mates and mates shorter than seedSplitMin reach the device"""
    tcp.run_cli_case(GPU_CLI, "pe101", tcl.PE[tag], 500, tmp_path, fastq_hook=lambda info, d: tcl._with_adapters(info["fastq"], d, (tcl.AD1, tcl.AD2), zero_len=True))


@pytest.mark.parametrize("tag", sorted(tcl.SE))
def test_clipped_single_on_gpu(tag, tmp_path, built):
    tcp.run_cli_case(GPU_CLI, "se50", tcl.SE[tag], 500, tmp_path, fastq_hook=lambda info, d: tcl._with_adapters(info["fastq"], d, (tcl.AD1,), zero_len=True))


def test_wasp_in_pieces_on_gpu(tmp_path, built):
    """batch of 150 reads: the WASP re-mapping batch (several copies per read) is larger than the context and goes through in rebased pieces"""
    tcp.run_cli_case(GPU_CLI, "pe101", ["--waspOutputMode", "SAMtag", "--varVCFfile", "VCF", "--outSAMtype", "BAM", "Unsorted", "--outSAMattributes", "NH", "HI", "AS", "nM", "vA", "vG", "vW"], 150, tmp_path)


@pytest.mark.parametrize("tag", ["pe", "se"])
def test_chimeric_partner_chosen_on_the_device(tag, tmp_path, built):
    """staramd_params::resultSelect 2: the partner loop of chimericDetectionOld in k_stitch_finish.  The stress sets of tests/test_chimeric.py (60 % chimeric fragments,
    hundreds of junction lines) through the shipped binary, partner chosen on the device (default) and on the host (STARAMD_CHIM_ON_DEVICE=0): both equal to the reference,
    and the device form returns a fraction of the transcripts"""
    import subprocess
    import test_chimeric as tch
    info, d = tch._stress(tag, tmp_path)
    ref = tcp.refstar.align(info["idx"], info["fastq"], os.path.join(d, "ref_"), threads=1, extra=info["extra"])
    lines = lambda p: [l for l in open(p + "Chimeric.out.junction") if not l.startswith("# 2.7.11b")]
    assert len(lines(ref)) >= 400
    for on in ("1", "0", "noprune"):             # noprune: on the device with every window of every read stitched (STARAMD_PRUNE without bit 3)
        new = os.path.join(d, "cli%s_" % on)
        env = dict(os.environ, STARAMD_CHIM_ON_DEVICE="1" if on == "noprune" else on, STARAMD_VERBOSE="1")
        if on == "noprune":
            env["STARAMD_PRUNE"] = "7"
        p = subprocess.run([GPU_CLI, "--runMode", "alignReads", "--genomeDir", info["idx"], "--readFilesIn"] + info["fastq"] + ["--outFileNamePrefix", new, "--runThreadN", "4", "--gpuBatchReads", "700"] + info["extra"],
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, timeout=600)
        assert p.returncode == 0, p.stderr[-1500:]
        assert ("partner of chimeric detection chosen on the device" in p.stderr) == (on != "0"), p.stderr[-800:]
        assert lines(ref) == lines(new)
        assert tcp.refstar.sam_body_sorted(ref + "Aligned.out.sam") == tcp.refstar.sam_body_sorted(new + "Aligned.out.sam")
        assert open(ref + "SJ.out.tab", "rb").read() == open(new + "SJ.out.tab", "rb").read()
        assert tcp.refstar.final_log_counters(ref + "Log.final.out") == tcp.refstar.final_log_counters(new + "Log.final.out")
