// runner.cpp -- host run orchestration (the alignReads part of STAR.cpp main(), source/STAR.cpp:58-313)
// and its C interface `sah_*`, used by the CLI (main.cpp) and, through ctypes, by tests/ and bench.py.
// The engine (HIP library behind include/star_amd.h) is NOT linked here: callers pass result buffers in,
// so the same post-map code is exercised with results produced by the HIP engine (product) or,
// in tests only, by the CPU oracle.
#include "host.h"
#include <unistd.h>
#include <fcntl.h>
#include <atomic>
#include <sys/stat.h>
#include <cerrno>
#include "../../../include/star_amd_host.h"
#include <cstring>
#include <algorithm>
#include <ctime>
#include <memory>
#include <thread>
#include <mutex>
#include <condition_variable>
#include <deque>
#include <array>
#include <random>
#include <chrono>
#include <sys/mman.h>

namespace staramd {

struct Runner {
    RunParams P;
    GenomeIndex gi;
    FastqReader reader;
    ReadBatch batch;
    staramd_batch batchView;
    static const int NSLOT = 24;        // batch slots of the pipelined CLI (3 + 2 per GPU are in use: cli_run.cpp)
    ReadBatch slots[NSLOT];             // pipelined CLI: parse / map / post-map work on different slots
    Variation variation;                      // --varVCFfile
    WaspBatch waspMain, waspSlots[NSLOT];         // --waspOutputMode SAMtag: the allele-swapped reads of batch / slots[k], mapped as one more batch
    MergedBatch mergedMain, mergedSlots[NSLOT];   // --peOverlapNbasesMin: the merged mates of batch / slots[k], mapped as a second batch
    std::unique_ptr<PostMap> post;
    OutSJ sj;
    Stats stats;
    FILE *samOut = nullptr;
    bool toolDone = false;                          // --runMode inputAlignmentsFromBAM: everything happened in init()
    bool generateMode = false; GenerateJob genJob;  // --runMode genomeGenerate: FASTA scanned in init(), SA / SAindex built by the caller on the device
    FILE *chimOut = nullptr;                        // Chimeric.out.junction
    FILE *chimSamOut = nullptr;                     // Chimeric.out.sam (--chimOutType SeparateSAMold)
    FILE *unmappedOut[2] = {nullptr, nullptr};      // --outReadsUnmapped Fastx: Unmapped.out.mate1 / mate2
    std::string error;
    // three stages of the pipelined front end run on their own threads; each reports through a string of its own (guarded by errM), and
    // sah_error hands the calling thread a private copy: parseError -- reader (sah_parse_slot), mapError -- mapper threads (sah_wasp_results_slot),
    // `error` -- the emit / control side (one thread at a time)
    std::string parseError, mapError; std::mutex errM;
    SjdbLoci sjdbLoci;                  // junctions known so far (generated genome, --sjdbFileChrStartEnd, 1st pass)
    std::string insertLog;
    bool pass1 = false;
    // --outFilterType BySJout (STAR.cpp:203-220): stage 1 maps everything and holds reads with unannotated junctions, stage 2 maps
    // the held reads again with the filtered novel junctions as a whitelist inside the stitcher
    int bySJoutStage = 0;               // 0 off, 1, 2
    OutSJ sj1;                          // junctions of every read of stage 1 (chunkOutSJ1)
    std::string heldText[2];            // held reads as FASTQ text
    std::vector<uint64_t> novelStart, novelEnd;
    GeneAnnotation genes; GeneCounts geneCounts;      // --quantMode GeneCounts
    TranscriptAnnotation transcripts; FILE *quantOut = nullptr;     // --quantMode TranscriptomeSAM -> Aligned.toTranscriptome.out.bam
    MultOrder multOrder;
    std::mt19937 rngMultOrder; std::uniform_real_distribution<double> rngUniformReal0to1{0.0, 1.0};   // ReadAlign.cpp:11-12 (one stream: iChunk 0)
    std::vector<std::string> coordChunks; std::vector<BamKey> coordKeys;     // --outSAMtype BAM SortedByCoordinate: every record, until finish()
    int wireTable = 0;                                // which junction table sah_sj_export / import / clear address: 0 = sj, 1 = sj1
    OutSJ &wire() { if (wireTable != 1) sjBg.drainInto(sj); return wireTable == 1 ? sj1 : sj; }
    // The output junction table grows by one record per junction per read.  Whenever it passes sjKick records it is handed to a thread that collapses it and folds it
    // into what was collapsed before (collapse is a sum / maximum per junction: the order does not matter), so that what is left to sort when the run ends
    // (finish() is inside the timed region of a run; SJ.out.tab needs the collapsed table) is less than sjKick records + the last batch, not everything since the
    // last 4 M-record collapse (ReadAlignChunk_mapChunk.cpp:66-86 collapses when the chunk's buffer is full, for bounded memory; this does the same, off the path)
    struct SjBackground {
        OutSJ folded, work; std::thread th;
        void join() { if (th.joinable()) th.join(); }
        void kick(OutSJ &live) { join(); work.data.swap(live.data); th = std::thread([this] { work.collapse(); folded.mergeFrom(work); work.data.clear(); folded.collapse(); }); }
        void drainInto(OutSJ &live) { join(); if (!folded.data.empty()) { live.mergeFrom(folded); folded.data.clear(); } }
        ~SjBackground() { join(); }
    } sjBg;
    size_t sjKick = 1000000;
    int64_t readMapNumberUser = -1;

    bool init(int argc, char **argv) {
        time(&stats.timeStart);
        error = P.parse(argc, argv);
        if (!error.empty()) return false;
        if (const char *e = getenv("STARAMD_SJ_KICK")) sjKick = (size_t)strtoull(e, nullptr, 10);      // (tests: a few hundred records, so that the background collapse runs on small data)
        {   // createDirectory (streamFuns.cpp:10-35): the directory part of --outFileNamePrefix is made, with its parents, mode S_IRWXU
            const std::string dirPath = P.outFileNamePrefix.substr(0, P.outFileNamePrefix.find_last_of('/') + 1);
            if (!dirPath.empty() && mkdir(dirPath.c_str(), S_IRWXU) == -1 && errno != EEXIST) {
                for (size_t i1 = dirPath.find_first_of('/', 1); i1 < dirPath.size(); i1 = dirPath.find_first_of('/', i1 + 1)) {
                    const std::string d1 = dirPath.substr(0, i1);
                    if (mkdir(d1.c_str(), S_IRWXU) == -1 && errno != EEXIST) {
                        error = "EXITING because of fatal OUTPUT FILE error: could not create output directory: " + d1 + " for --outFileNamePrefix " + P.outFileNamePrefix + "\n ERROR: " + strerror(errno) + "\nSOLUTION: check the path and permissions.\n";
                        return false;
                    }
                }
            }
        }
        if (P.runModeGenerate) {                                    // index generation: no reads; the device stages are run by the caller
            error = genomeGenerateScan(P, gi, genJob);
            generateMode = true;
            return error.empty();
        }
        if (P.runModeFromBAM) {                                     // a tool mode: no index, no reads, no engine
            error = signalFromBamFile(P, P.inputBAMfile, P.outFileNamePrefix + "Signal");
            toolDone = true;
            return error.empty();
        }
        error = gi.load(P.genomeDir);
        if (!error.empty()) return false;
        if (P.sjdbInsertYes()) {
            // sjdbOverhang of the run (Genome_genomeLoad.cpp:113-125) and the run-time directories (Parameters.cpp:817-824,1019-1034)
            uint32_t gov = gi.view.sjdbOverhang;
            if (!P.sjdbOverhangSet && gov > 0) P.sjdbOverhang = gov;
            else if (gi.sjdbInfoExists && P.sjdbOverhangSet && P.sjdbOverhang != gov) {
                error = "EXITING because of fatal PARAMETERS error: present --sjdbOverhang=" + std::to_string(P.sjdbOverhang) + " is not equal to the value at the genome generation step =" + std::to_string(gov) + "\nSOLUTION: \n";
                return false;
            }
            if (P.sjdbOverhang == 0 || P.sjdbOverhang > 500) { error = "EXITING because of fatal PARAMETERS error: pGe.sjdbOverhang <=0 (or > 500) while junctions are inserted on the fly"; return false; }
            gi.view.sjdbOverhang = P.sjdbOverhang; gi.view.sjdbLength = 2 * P.sjdbOverhang + 1;
            P.sjdbInsertOutDir = P.outFileNamePrefix + "_STARgenome/";
            error = makeRunDir(P.sjdbInsertOutDir, P.runDirPermAll);
            if (!error.empty()) return false;
            if (P.twopass) { P.twopassDir = P.outFileNamePrefix + "_STARpass1/"; error = makeRunDir(P.twopassDir, P.runDirPermAll); if (!error.empty()) return false; }
        }
        P.finalize(gi);
        if (P.sjdbInsertPass1()) {                                  // STAR.cpp:147-150: insertion before the (1st) mapping pass
            error = sjdbInsertJunctions(P, gi, sjdbLoci, false, "", insertLog);
            if (!error.empty()) return false;
        }
        error = reader.open(P.readFilesIn, P.readFilesCommand, P.readFilesSAMmates);
        if (!error.empty()) return false;
        if (!P.varVCFfile.empty()) { error = variation.load(P, gi); if (!error.empty()) return false; P.var = &variation; }
        post.reset(new PostMap(P, gi));
        rngMultOrder.seed((unsigned)P.runRNGseed);                  // ReadAlign.cpp:11 (iChunk 0)
        if (P.quantTrSAM) {
            error = transcripts.load(P.sjdbGTFfile.empty() ? P.genomeDir : P.sjdbInsertOutDir);
            if (!error.empty()) return false;
            post->transcripts = &transcripts;
            if (P.quantTrBAMcompression > -2) {
                std::string qp = P.outFileNamePrefix + "Aligned.toTranscriptome.out.bam";
                quantOut = P.outStd == "BAM_Quant" ? stdout : fopen(qp.c_str(), "wb");
                if (!quantOut) { error = "EXITING because of fatal ERROR: could not create output file " + qp; return false; }
                std::string h;
                if (!bgzfCompress(post->quantBamHeader(), P.quantTrBAMcompression, h)) { error = "EXITING because of fatal ERROR: BGZF compression failed"; return false; }
                fwrite(h.data(), 1, h.size(), quantOut);
            }
        }
        if (P.quantGeneCounts) {                                    // Transcriptome.cpp:12-16: a GTF given at the mapping stage wins
            error = genes.load(P.sjdbGTFfile.empty() ? P.genomeDir : P.sjdbInsertOutDir);
            if (!error.empty()) return false;
            geneCounts = GeneCounts(genes.geID.size());
            post->genes = &genes;
        }
        if (P.outFilterBySJout && !P.twopass) { bySJoutStage = 1; P.dev.outFilterBySJoutStage = 1; }
        if (P.twopass) {                                            // twoPassRunPass1.cpp:14-47: no SAM, own read limit
            pass1 = true; post->samOff = true; readMapNumberUser = P.readMapNumber;
            P.dev.chimSegmentMinPositive = 0;                       // twoPassRunPass1.cpp:24 (restored with the index re-upload after the pass)
            if (P.twopass1readsN >= 0) P.readMapNumber = P.readMapNumber < 0 ? P.twopass1readsN : std::min(P.readMapNumber, P.twopass1readsN);
        }
        if (P.chim.segmentMin > 0 && P.chim.outJunctions) {
            std::string cp = P.outFileNamePrefix + "Chimeric.out.junction";
            chimOut = fopen(cp.c_str(), "wb");
            if (!chimOut) { error = "EXITING because of fatal ERROR: could not create output file " + cp; return false; }
            if (P.chim.multimapNmax > 0)                            // column names of the multimapping algorithm's table (ParametersChimeric_initialize.cpp:48-72)
                fputs("chr_donorA\tbrkpt_donorA\tstrand_donorA\tchr_acceptorB\tbrkpt_acceptorB\tstrand_acceptorB\tjunction_type\trepeat_left_lenA\trepeat_right_lenB\tread_name\t"
                      "start_alnA\tcigar_alnA\tstart_alnB\tcigar_alnB\tnum_chim_aln\tmax_poss_aln_score\tnon_chim_aln_score\tthis_chim_aln_score\tbestall_chim_aln_score\tPEmerged_bool\treadgrp\n", chimOut);
        }
        if (P.chim.segmentMin > 0 && P.chim.outSamOld) {            // ParametersChimeric_initialize.cpp:39-42
            std::string cp = P.outFileNamePrefix + "Chimeric.out.sam";
            chimSamOut = fopen(cp.c_str(), "wb");
            if (!chimSamOut) { error = "EXITING because of fatal ERROR: could not create output file " + cp; return false; }
            std::string h = post->samHeader(); fwrite(h.data(), 1, h.size(), chimSamOut);
        }
        if (P.outReadsUnmappedFastx)
            for (uint32_t m = 0; m < P.dev.readNmates; m++) {
                std::string up = P.outFileNamePrefix + "Unmapped.out.mate" + std::to_string(m + 1);
                unmappedOut[m] = fopen(up.c_str(), "wb");
                if (!unmappedOut[m]) { error = "EXITING because of fatal ERROR: could not create output file " + up; return false; }
            }
        if (P.outSAMnone) post->samOff = true;                      // --outSAMtype None
        else if (P.outBAMcoord && !P.outBAMunsorted) {}             // only Aligned.sortedByCoord.out.bam, written at the end of the run
        else {
            std::string samPath = P.outFileNamePrefix + (P.outBAMunsorted ? "Aligned.out.bam" : "Aligned.out.sam");
            // a regular file is opened for reading too: the writer maps the part of the file a batch goes to, and a shared writable mapping needs a descriptor open read-write
            // (a FIFO or a device keeps "wb": opening a FIFO read-write would not wait for its reader)
            struct stat stOut; const bool special = stat(samPath.c_str(), &stOut) == 0 && !S_ISREG(stOut.st_mode);
            samOut = (P.outBAMunsorted ? P.outStd == "BAM_Unsorted" : P.outStd == "SAM") ? stdout : fopen(samPath.c_str(), special ? "wb" : "w+b");
            if (!samOut) { error = "EXITING because of fatal ERROR: could not create output file " + samPath; return false; }
            setvbuf(samOut, nullptr, _IOFBF, 1 << 22);
            std::string h;
            if (P.outBAMunsorted) { if (!bgzfCompress(post->bamHeader(), P.outBAMcompression, h)) { error = "EXITING because of fatal ERROR: BGZF compression failed"; return false; } }
            else h = post->samHeader();
            fwrite(h.data(), 1, h.size(), samOut);
        }
        {   // Log.out and Log.progress.out exist with the reference's first and last lines (pipelines look for the files and for "ALL DONE!"); what the
            // reference logs in between (its parameter dump, timing of its own stages) has no counterpart here
            FILE *lo = fopen((P.outFileNamePrefix + "Log.out").c_str(), "wb");
            if (lo) {
                fprintf(lo, "STAR version=2.7.11b\nstar_amd: MI355X seed-search-and-stitch engine behind STAR's alignReads command line\n##### Command Line:\n%s\n", P.commandLine.c_str());
                fprintf(lo, "Finished loading and checking parameters\nNumber of real (reference) chromosomes= %u\n", gi.view.nChrReal);
                for (uint32_t i = 0; i < gi.view.nChrReal; i++) fprintf(lo, "%u\t%s\t%llu\t%llu\n", i + 1, gi.chrName[i].c_str(), (unsigned long long)gi.chrLength[i], (unsigned long long)gi.chrStart[i]);
                if (!insertLog.empty()) fputs(insertLog.c_str(), lo);
                fclose(lo);
            }
            FILE *lp = fopen((P.outFileNamePrefix + "Log.progress.out").c_str(), "wb");
            if (lp) {
                fputs("           Time    Speed        Read     Read   Mapped   Mapped   Mapped   Mapped Unmapped Unmapped Unmapped Unmapped\n"
                      "                    M/hr      number   length   unique   length   MMrate    multi   multi+       MM    short    other\n", lp);
                fclose(lp);
            }
        }
        startWriter();
        time(&stats.timeStartMap);
        return true;
    }
    int nextBatch(uint64_t maxReads) {
        std::string err;
        bool ok = reader.nextBatch(batch, P, maxReads, err);
        if (!err.empty()) { error = err; return -1; }
        if (!ok) return 0;
        batchView = batch.view();
        if (P.peOverlapNbasesMin > 0 && P.dev.readNmates == 2) mergedMain.build(batch, P);
        return (int)batch.n;
    }
    // ---- SAM text goes to the file on its own thread: formatting of batch k+1 overlaps the write of batch k.  Two sets of
    // per-thread text buffers alternate and keep their capacity (no fresh pages per batch).
    struct OutSet { std::vector<std::string> sams, raws; uint32_t used = 0; };
    std::vector<OutSJ> sjScratch;            // emitBatch: junction records per thread (one batch at a time)
    OutSet outSets[2];
    std::mutex wm; std::condition_variable wcv;
    std::deque<int> freeSets, fullSets; bool writerStop = false, writerFailed = false;
    std::thread writerThread;
    int samFd = -1; uint64_t samPos = 0;     // positional writes of the SAM / unsorted BAM text (regular file)
    int samSeekable = -1;                    // -1 not looked at yet, 0 pipe / FIFO / character device (sequential fwrite), 1 regular file
    uint64_t nMappedWrites = 0;              // batches whose text went out through a mapping of the output file (sah_fast_path_counts)
    double tWriter = 0, tEmitWaitSet = 0, tEmitFormat = 0, tEmitTail = 0;      // seconds, whole run (STARAMD_HOST_TIMING prints them at the end)
    void writerLoop() {
        for (;;) {
            int k;
            { std::unique_lock<std::mutex> l(wm); wcv.wait(l, [&] { return !fullSets.empty() || writerStop; }); if (fullSets.empty()) return; k = fullSets.front(); fullSets.pop_front(); }
            OutSet &o = outSets[k];
            CpuScope cpuScope(CPU_WRITE);
            const auto tw0 = std::chrono::steady_clock::now();
            struct AddTime { double &acc; std::chrono::steady_clock::time_point t0; ~AddTime() { acc += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); } } addTime{tWriter, tw0};
            if (samSeekable < 0 && samOut) {            // a named pipe (mkfifo Aligned.out.sam | samtools ...) or a device has no offsets: pwrite fails with ESPIPE there
                struct stat st; samSeekable = (samOut != stdout && fstat(fileno(samOut), &st) == 0 && S_ISREG(st.st_mode) && ftello(samOut) >= 0) ? 1 : 0;
            }
            static const int wantMmap = getenv("STARAMD_WRITER_MMAP") ? atoi(getenv("STARAMD_WRITER_MMAP")) : 1;
            if (samOut && samSeekable == 1 && (o.used > 1 || samFd >= 0 || wantMmap >= 2)) {
                // a regular file: the per-thread text buffers go out side by side, each at its own offset (one fwrite stream tops out near
                // 2 GB/s on tmpfs, the SAM text of one GPU runs at about that)
                if (samFd < 0) { fflush(samOut); samFd = fileno(samOut); samPos = (uint64_t)ftello(samOut); }
                std::vector<uint64_t> at(o.used + 1, samPos);
                for (uint32_t t = 0; t < o.used; t++) at[t + 1] = at[t] + o.sams[t].size();
                // write() / pwrite() into ONE file are serialised by the inode lock (tmpfs: 2.9 GB/s from two threads on a box whose tmpfs takes 5.8 / 11.7 / 18.5 GB/s from
                // 1 / 2 / 4 streams into separate files: the SAM writer at 230 MB per batch was the slowest stage of the pipeline there, 80 ms against 52 ms of kernels).  So the
                // file is grown to the batch's end and the new part mapped: the threads copy their ranges into the mapping and take their page faults side by side.
                const uint32_t wantW = getenv("STARAMD_WRITER_THREADS") ? (uint32_t)std::max(1, atoi(getenv("STARAMD_WRITER_THREADS"))) : (uint32_t)std::max(1, std::min(8, P.runThreadN / 2));      // (follows --runThreadN like every helper count; 8 of 16: the writer is level with the kernels now, 4 copy threads lost 2 % to 8 on one box -- profiles/r06_e2e_session37_*)
                const uint64_t total = at[o.used] - samPos;
                char *map = nullptr; uint64_t mapOff = 0, mapLen = 0;
                // the blocks are reserved before they are written to (posix_fallocate also sets the new size): a full file system is an error return here, not a SIGBUS in a copy
                if (wantMmap && total > 0 && (wantMmap >= 2 || total >= (1u << 20)) && posix_fallocate(samFd, (off_t)samPos, (off_t)total) == 0) {      // (2: whatever the size -- tests)
                    static const uint64_t page = (uint64_t)std::max<long>(sysconf(_SC_PAGESIZE), 4096); mapOff = samPos & ~(page - 1); mapLen = at[o.used] - mapOff;
                    void *m = mmap(nullptr, (size_t)mapLen, PROT_READ | PROT_WRITE, MAP_SHARED, samFd, (off_t)mapOff);
                    if (m != MAP_FAILED) { map = (char *)m; nMappedWrites++; }
                }
                static const uint32_t pwriteW = getenv("STARAMD_WRITER_PWRITE_THREADS") ? (uint32_t)std::max(1, atoi(getenv("STARAMD_WRITER_PWRITE_THREADS"))) : 2u;      // (positional writes: 1 = one stream, no contention for the inode lock)
                const uint32_t W = std::min<uint32_t>(map ? wantW : pwriteW, o.used);
                std::atomic<uint32_t> next(0); std::atomic<bool> bad(false);
                auto put = [&] {
                    for (;;) {
                        uint32_t t = next.fetch_add(1);
                        if (t >= o.used) break;
                        const char *p = o.sams[t].data(); uint64_t left = o.sams[t].size(), off = at[t];
                        if (map) { if (left) memcpy(map + (off - mapOff), p, left); continue; }
                        while (left) { ssize_t w = pwrite(samFd, p, left, (off_t)off); if (w <= 0) { bad = true; return; } p += w; left -= (uint64_t)w; off += (uint64_t)w; }
                    }
                };
                std::vector<std::thread> th;
                for (uint32_t i = 1; i < W; i++) th.emplace_back([&] { CpuScope cs(CPU_WRITE); put(); });
                put();
                for (auto &x : th) x.join();
                if (map) munmap(map, (size_t)mapLen);
                samPos = at[o.used];
                if (bad) writerFailed = true;
            } else
            for (uint32_t t = 0; t < o.used; t++)
                if (!o.sams[t].empty() && samOut && fwrite(o.sams[t].data(), 1, o.sams[t].size(), samOut) != o.sams[t].size()) writerFailed = true;
            { std::lock_guard<std::mutex> l(wm); freeSets.push_back(k); }
            wcv.notify_all();
        }
    }
    void startWriter() { freeSets = {0, 1}; writerThread = std::thread([this] { writerLoop(); }); }
    void stopWriter() {
        if (!writerThread.joinable()) return;
        { std::lock_guard<std::mutex> l(wm); writerStop = true; }
        wcv.notify_all();
        writerThread.join();
    }
    // post-map of one batch on --runThreadN host threads: contiguous read ranges, per-thread SAM buffer / junctions / Stats
    // (what the reference keeps per ReadAlignChunk), SAM text written in read order
    bool emitBatch(const ReadBatch &bt, const staramd_results *r, const MergedBatch *mg = nullptr, const staramd_results *mgRes = nullptr, const WaspBatch *wasp = nullptr) {
        if (P.wasp && !pass1 && !wasp) { error = "EXITING because of FATAL ERROR: --waspOutputMode: the allele-swapped reads of the batch were not mapped (sah_wasp_batch / sah_wasp_results)"; return false; }
        const std::vector<int8_t> *waspType = (wasp && !pass1) ? &wasp->type : nullptr;
        if (mg && (mg->reads.n == 0 || !mgRes)) { if (mg->reads.n > 0) { error = "EXITING because of FATAL ERROR: --peOverlapNbasesMin: the merged mates of the batch were not mapped (sah_merged_batch / sah_emit_merged)"; return false; } mg = nullptr; }
        // T contiguous read ranges, each with buffers of its own, formatted by Wk worker threads that take the next range when they are done with one: with as many
        // ranges as threads the section lasts as long as its slowest thread, and on a shared host (the GPU boxes: 16 CPUs of a 256-thread machine) one descheduled
        // thread held the batch for several times the mean (per-thread busy 2.5 - 5.6 ms in a 24 ms section).  Four ranges per thread; gene counting keeps one
        // (a count table per range)
        const uint32_t Wk = (uint32_t)std::max(1, std::min(P.runThreadN, 256));
        uint32_t T = (P.quantGeneCounts || P.quantTrSAM) ? Wk : std::min<uint32_t>(4 * Wk, 256);
        T = std::max<uint32_t>(1, std::min<uint32_t>(T, bt.n / 256));       // at least 256 reads per range
        int k;
        auto te0 = std::chrono::steady_clock::now();
        { std::unique_lock<std::mutex> l(wm); wcv.wait(l, [&] { return !freeSets.empty(); }); k = freeSets.front(); freeSets.pop_front(); }
        auto te1 = std::chrono::steady_clock::now(); tEmitWaitSet += std::chrono::duration<double>(te1 - te0).count();
        OutSet &o = outSets[k];
        if (o.sams.size() < T) { o.sams.resize(T); o.raws.resize(T); }
        o.used = T;
        std::vector<std::string> errs(T); std::vector<Stats> sts(T);
        // per-thread junction records of the batch: the vectors are kept between batches (a fresh vector grows by reallocation up to a few hundred KB per thread and
        // batch, which glibc serves with mmap / munmap: page faults on every batch)
        std::vector<OutSJ> &sjs = sjScratch;
        if (sjs.size() < T) sjs.resize(T);
        for (uint32_t t = 0; t < T; t++) sjs[t].data.clear();
        const bool stage1 = bySJoutStage == 1;
        std::vector<OutSJ> sj1s(stage1 ? T : 0); std::vector<std::vector<uint32_t> > helds(stage1 ? T : 0);
        const bool quant = P.quantGeneCounts && !pass1;             // twoPassRunPass1.cpp:24-29: no quantification in the 1st pass
        std::vector<GeneCounts> gcs(quant ? T : 0, GeneCounts(quant ? genes.geID.size() : 0));
        std::vector<std::vector<BamKey> > keyss(P.outBAMcoord ? T : 0);
        const bool trSAM = P.quantTrSAM && quantOut && !pass1;       // twoPassRunPass1.cpp:24-29
        std::vector<std::string> qraws(trSAM ? T : 0); std::vector<std::vector<QuantPatch> > qpatches(trSAM ? T : 0);
        const bool chimOn = P.chim.segmentMin > 0 && !pass1;        // twoPassRunPass1.cpp:24: no chimeric detection in the 1st pass
        std::vector<std::string> chims(chimOn ? T : 0), chimSams(chimOn && chimSamOut ? T : 0);
        const bool unm = P.outReadsUnmappedFastx && !pass1;
        std::vector<std::array<std::string, 2> > unms(unm ? T : 0);
        const bool randomOrder = P.outMultimapperRandom;
        uint32_t per = (bt.n + T - 1) / T;
        if (randomOrder) {
            // with TranscriptomeSAM the shuffles of a read and its draw of the primary transcriptomic alignment alternate in one random stream: the
            // number of transcriptomic alignments of every read is needed first (it does not depend on the order), so the quantification runs
            // once into throw-away buffers, then all draws of the batch are made in read order, then the batch is formatted for real
            std::vector<uint32_t> nAlignT;
            if (trSAM) {
                nAlignT.assign(bt.n, 0);
                auto count = [&](uint32_t t) {
                    uint32_t lo = std::min(bt.n, t * per), hi = std::min(bt.n, lo + per);
                    std::string sam0, q0; OutSJ sj0, sj10; Stats st0; std::vector<uint32_t> held0; std::vector<QuantPatch> qp0;
                    const bool samOff0 = post->samOff;
                    (void)samOff0;
                    std::string chim0;      // reads whose chimera goes into the BAM are not quantified: the detection has to run here too
                    PostMap::RangeOut ro; ro.sam = &sam0; ro.sj = &sj0; ro.st = &st0; if (stage1) { ro.sj1 = &sj10; ro.held = &held0; }
                    if (chimOn) ro.chimJunction = &chim0;
                    ro.quantBam = &q0; ro.quantPatches = &qp0;
                    PostMap::RangeIn ri; ri.dry = true; ri.merged = mg; ri.mergedRes = mgRes;
                    errs[t] = post->processRange(bt, *r, lo, hi, ro, ri);
                    for (const QuantPatch &p : qp0) nAlignT[p.ir] = p.nAlignT + 1;
                };
                std::atomic<uint32_t> nextC(0);
                auto countLoop = [&] { for (;;) { const uint32_t t = nextC.fetch_add(1); if (t >= T) break; count(t); } };
                std::vector<std::thread> th;
                for (uint32_t w = 1; w < std::min(Wk, T); w++) th.emplace_back([&] { CpuScope cs(CPU_EMIT); countLoop(); });
                countLoop();
                for (auto &x : th) x.join();
            }
            post->drawMultOrder(bt, *r, [&] { return rngUniformReal0to1(rngMultOrder); }, multOrder, trSAM ? &nAlignT : nullptr, mg, mgRes);
        }
        static const bool hostTiming = getenv("STARAMD_HOST_TIMING") != nullptr;
        std::vector<double> tThread(hostTiming ? T : 0);
        const double msBefore = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - te1).count();
        int waspEndOfBatch = post->waspCarry;
        auto work = [&](uint32_t t) {
            struct Tm { std::vector<double> &v; uint32_t t; std::chrono::steady_clock::time_point t0; ~Tm() { if (!v.empty()) v[t] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); } } tm{tThread, t, std::chrono::steady_clock::now()};
            uint32_t lo = std::min(bt.n, t * per), hi = std::min(bt.n, lo + per);
            // the thread appends to objects of its own for the length of the range: neighbouring elements of the per-thread vectors share
            // cache lines, and a std::string / std::vector rewrites its size word on every append (measured: 4x per record with 8 threads)
            std::string samL, rawL; OutSJ sjL; Stats stL;
            samL.swap(o.sams[t]); rawL.swap(o.raws[t]); sjL.data.swap(sjs[t].data);
            struct Back { std::string &a, &al, &b, &bl; OutSJ &s, &sl; Stats &st, &stl; ~Back() { a.swap(al); b.swap(bl); s.data.swap(sl.data); st = stl; } } back{o.sams[t], samL, o.raws[t], rawL, sjs[t], sjL, sts[t], stL};
            samL.clear();
            const bool bamOut = (P.outBAMunsorted || P.outBAMcoord) && !post->samOff;   // BAM: this thread's records are compressed here, block by block (bgzf.cpp)
            std::string &raw = rawL;
            if (bamOut) raw.clear();
            PostMap::RangeOut ro;
            ro.sam = bamOut ? &raw : &samL; ro.sj = &sjL; ro.st = &stL;
            if (stage1) { ro.sj1 = &sj1s[t]; ro.held = &helds[t]; }
            if (quant) ro.gc = &gcs[t];
            if (bamOut && P.outBAMcoord) ro.bamKeys = &keyss[t];
            if (unm) ro.unmappedFastx = unms[t].data();
            if (chimOn) ro.chimJunction = &chims[t];
            if (!chimSams.empty()) ro.chimSam = &chimSams[t];
            if (trSAM) { ro.quantBam = &qraws[t]; ro.quantPatches = &qpatches[t]; }
            PostMap::RangeIn ri;
            ri.order = randomOrder ? &multOrder : nullptr; ri.merged = mg; ri.mergedRes = mgRes; ri.waspType = waspType;
            if (waspType && hi == bt.n && lo < hi) ro.waspEnd = &waspEndOfBatch;     // (one range ends the batch)
            errs[t] = post->processRange(bt, *r, lo, hi, ro, ri);
            if (!bamOut) return;
            bool cut = false;
            if (P.outBAMcoord) for (const BamKey &k : keyss[t]) if (k.len & 0x80000000u) { cut = true; break; }
            if (cut) {                                           // KeepPairs with both BAM files: records that belong to the sorted one only
                std::string only; size_t pos = 0;
                for (BamKey &k : keyss[t]) if (k.len & 0x80000000u) { k.len &= 0x7fffffffu; only.append(raw, pos, k.off - pos); pos = k.off + k.len; }
                only.append(raw, pos, std::string::npos);
                if (errs[t].empty() && !bgzfCompress(only, P.outBAMcompression, samL)) errs[t] = "EXITING because of fatal ERROR: BGZF compression failed";
                return;
            }
            if (errs[t].empty() && P.outBAMunsorted && !bgzfCompress(raw, P.outBAMcompression, samL)) errs[t] = "EXITING because of fatal ERROR: BGZF compression failed";
        };
        if (T == 1) work(0);
        else {
            std::atomic<uint32_t> nextR(0);
            auto workLoop = [&] { for (;;) { const uint32_t t = nextR.fetch_add(1); if (t >= T) break; work(t); } };
            std::vector<std::thread> th;
            for (uint32_t w = 1; w < std::min(Wk, T); w++) th.emplace_back([&] { CpuScope cs(CPU_EMIT); workLoop(); });
            workLoop();
            for (auto &x : th) x.join();
        }
        auto te2 = std::chrono::steady_clock::now(); tEmitFormat += std::chrono::duration<double>(te2 - te1).count();
        if (hostTiming && T > 0) { double mn = 1e9, mx = 0, sm = 0; for (double x : tThread) { mn = std::min(mn, x); mx = std::max(mx, x); sm += x; }
            fprintf(stderr, "  emit: %u ranges on %u threads, section %.2f ms, per-range min %.2f / mean %.2f / max %.2f ms\n", T, std::min(Wk, T), std::chrono::duration<double, std::milli>(te2 - te1).count(), mn, sm / T, mx); }
        struct AddTail { double &acc; std::chrono::steady_clock::time_point t0; ~AddTail() { acc += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); } } addTail{tEmitTail, te2};
        for (uint32_t t = 0; t < T; t++) if (!errs[t].empty() && error.empty()) error = errs[t];
        if (!error.empty()) o.used = 0;
        { std::lock_guard<std::mutex> l(wm); fullSets.push_back(k); }
        wcv.notify_all();
        if (!error.empty()) return false;
        for (uint32_t t = 0; t < T; t++) { sj.mergeFrom(sjs[t]); stats.add(sts[t]); if (quant) geneCounts.add(gcs[t]); }
        if (hostTiming) fprintf(stderr, "  emit tail: junction records merged %.2f ms (%zu in the table)\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - te2).count(), sj.data.size());
        if (trSAM) {
            // one random number per mapped read, in read order, picks the primary transcriptomic alignment (ReadAlign_quantTranscriptome.cpp:69);
            // the flag is patched into the records (FLAG is the high half of the 5th word), then the text is compressed and written
            for (uint32_t t = 0; t < T; t++) {
                for (const QuantPatch &qp : qpatches[t]) {
                    uint32_t pick = randomOrder ? multOrder.quantPick[qp.ir] : (uint32_t)(int)(rngUniformReal0to1(rngMultOrder) * qp.nAlignT);
                    for (size_t k = 0; k < qp.recOffset.size(); k++) {
                        uint8_t &hi = (uint8_t &)qraws[t][qp.recOffset[k] + 19];
                        if (qp.recAlign[k] == pick) hi &= (uint8_t)~1u; else hi |= 1u;
                    }
                }
                std::string z;
                if (!bgzfCompress(qraws[t], P.quantTrBAMcompression, z) || fwrite(z.data(), 1, z.size(), quantOut) != z.size()) { error = "EXITING because of fatal ERROR: could not write Aligned.toTranscriptome.out.bam"; return false; }
            }
        }
        if (chimOn && chimOut) for (uint32_t t = 0; t < T; t++) if (!chims[t].empty()) fwrite(chims[t].data(), 1, chims[t].size(), chimOut);
        for (const std::string &cs : chimSams) if (!cs.empty()) fwrite(cs.data(), 1, cs.size(), chimSamOut);
        if (unm) for (uint32_t t = 0; t < T; t++) for (uint32_t m = 0; m < P.dev.readNmates; m++)
            if (!unms[t][m].empty() && unmappedOut[m]) fwrite(unms[t][m].data(), 1, unms[t][m].size(), unmappedOut[m]);
        if (P.outBAMcoord && !post->samOff)                         // keep the records for the coordinate sort at the end of the run (in memory)
            for (uint32_t t = 0; t < T; t++) {
                if (keyss[t].empty()) continue;
                uint32_t chunk = (uint32_t)coordChunks.size();
                for (BamKey &k : keyss[t]) { k.chunk = chunk; coordKeys.push_back(k); }
                coordChunks.emplace_back(P.outBAMunsorted ? o.raws[t] : std::move(o.raws[t]));
            }
        if (stage1) {
            for (uint32_t t = 0; t < T; t++) {
                sj1.mergeFrom(sj1s[t]);
                for (uint32_t ir : helds[t])                         // held reads, in input order (ReadAlign_outputAlignments.cpp:108-121)
                    for (uint32_t m = 0; m < P.dev.readNmates; m++) {
                        std::string &x = heldText[m];
                        x.push_back('@'); x += bt.name(ir); x += bt.filter[ir] == 'Y' ? " 0:Y:0 " : " 0:N:0 "; x += std::to_string(bt.readIndex(ir)); x.push_back(' '); x += std::to_string(bt.fileOf(ir));
                        if (!bt.extra((int)m, ir).empty()) { x.push_back('\x01'); x += bt.extra((int)m, ir); }
                        x.push_back('\n');
                        x += bt.seq((int)m, ir); x += "\n+\n"; x += bt.qual((int)m, ir); x.push_back('\n');
                    }
            }
            if (sj1.data.size() > 4000000) sj1.collapse();
        }
        if (waspType && !waspType->empty()) post->waspCarry = waspEndOfBatch;
        if (writerFailed) { error = "EXITING because of fatal ERROR: could not write Aligned.out.sam"; return false; }
        if (sj.data.size() > sjKick) sjBg.kick(sj);      // (bounded memory: ReadAlignChunk_mapChunk.cpp:66-86)
        if (hostTiming) fprintf(stderr, "  emit: before the threads %.2f ms, threads %.2f ms, tail %.2f ms\n", std::chrono::duration<double, std::milli>(te1 - te0).count() + msBefore, std::chrono::duration<double, std::milli>(te2 - te1).count() - msBefore,
                                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - te2).count());
        return true;
    }
    // Aligned.sortedByCoord.out.bam (BAMbinSortByCoordinate.cpp, BAMbinSortUnmapped.cpp): mapped records by (refID << 32 | pos, read order,
    // order of production), then the unmapped ones in read order.  The reference sorts genomic bins from temporary files; here all
    // records of the run are held in memory, ordered once and compressed on the host threads, slice by slice.
    std::string writeSortedBam() {
        std::string path = P.outFileNamePrefix + "Aligned.sortedByCoord.out.bam";
        const bool toStdout = P.outStd == "BAM_SortedByCoordinate";
        FILE *f = toStdout ? stdout : fopen(path.c_str(), "wb");
        if (!f) return "EXITING because of fatal ERROR: could not create output file " + path;
        std::string h;
        if (!bgzfCompress(post->bamHeader(true), P.outBAMcompression, h)) { if (!toStdout) fclose(f); return "EXITING because of fatal ERROR: BGZF compression failed"; }
        fwrite(h.data(), 1, h.size(), f);
        std::vector<uint64_t> ord(coordKeys.size());
        for (uint64_t i = 0; i < ord.size(); i++) ord[i] = i;
        const std::vector<BamKey> &K = coordKeys;
        std::sort(ord.begin(), ord.end(), [&K](uint64_t a, uint64_t b) { return K[a].g != K[b].g ? K[a].g < K[b].g : K[a].r != K[b].r ? K[a].r < K[b].r : a < b; });
        const uint64_t n = ord.size();
        const uint64_t T = (uint64_t)std::max(1, std::min(P.runThreadN, 64));
        const uint64_t per = std::max<uint64_t>(4096, (n + 4 * T - 1) / (4 * T));          // records per slice; slices are written in order
        bool failed = false;
        for (uint64_t base = 0; base < n && !failed; base += per * T) {
            std::vector<std::string> outS(T); std::vector<std::thread> th;
            auto work = [&](uint64_t t) {
                uint64_t lo = std::min(n, base + t * per), hi = std::min(n, lo + per);
                std::string raw;
                for (uint64_t i = lo; i < hi; i++) { const BamKey &k = K[ord[i]]; raw.append(coordChunks[k.chunk], k.off, k.len); }
                if (!bgzfCompress(raw, P.outBAMcompression, outS[t])) failed = true;
            };
            for (uint64_t t = 1; t < T; t++) th.emplace_back(work, t);
            work(0);
            for (auto &x : th) x.join();
            for (uint64_t t = 0; t < T; t++) if (!outS[t].empty() && fwrite(outS[t].data(), 1, outS[t].size(), f) != outS[t].size()) failed = true;
        }
        std::string e; bgzfEof(e); fwrite(e.data(), 1, e.size(), f);
        if (toStdout) fflush(stdout); else fclose(f);
        if (P.wig.yes && !failed) {                                 // STAR.cpp:275-283: signal tracks from the sorted alignments
            std::vector<const char *> recs(n);
            for (uint64_t i = 0; i < n; i++) { const BamKey &k = K[ord[i]]; recs[i] = coordChunks[k.chunk].data() + k.off; }
            std::string werr = writeSignal(P, std::vector<std::string>(gi.chrName.begin(), gi.chrName.begin() + gi.view.nChrReal), std::vector<uint64_t>(gi.chrLength.begin(), gi.chrLength.begin() + gi.view.nChrReal), P.outFileNamePrefix + "Signal", recs);
            if (!werr.empty()) return werr;
        }
        coordChunks.clear(); coordKeys.clear();
        return failed ? "EXITING because of fatal ERROR: could not write " + path : "";
    }
    bool emit(const staramd_results *r, const staramd_results *rMerged = nullptr) { return emitBatch(batch, r, P.peOverlapNbasesMin > 0 && P.dev.readNmates == 2 ? &mergedMain : nullptr, rMerged, P.wasp ? &waspMain : nullptr); }
    // end of the 1st pass (twoPassRunPass1.cpp:75-96): junctions + Log.final.out of the pass into _STARpass1/, insertion of the
    // junctions into the index, reads rewound.  The caller then re-uploads the index (staramd_update_index).
    bool endPass1() {
        if (!pass1) { error = "not in the 1st pass"; return false; }
        static const bool timing = getenv("STARAMD_HOST_TIMING") != nullptr;
        auto T0 = std::chrono::steady_clock::now();
        auto lap = [&](const char *what) { if (timing) { auto t = std::chrono::steady_clock::now(); fprintf(stderr, "  end of pass 1: %-28s %.1f ms\n", what, std::chrono::duration<double, std::milli>(t - T0).count()); T0 = t; } };
        sjBg.drainInto(sj);
        lap("junction table drained");
        error = sj.filterAndWrite(P, gi, P.twopassDir + "SJ.out.tab");
        if (!error.empty()) return false;
        stats.reportFinal(P.twopassDir + "Log.final.out");
        lap("SJ.out.tab + Log.final.out");
        error = sjdbInsertJunctions(P, gi, sjdbLoci, true, P.twopassDir + "SJ.out.tab", insertLog);
        if (!error.empty()) return false;
        lap("sjdbInsertJunctions");
        error = reader.reopen();
        if (!error.empty()) return false;
        time_t t0 = stats.timeStart;
        stats = Stats(); stats.timeStart = t0; time(&stats.timeStartMap);
        sj.data.clear();
        P.readMapNumber = readMapNumberUser; post->samOff = P.outSAMnone; pass1 = false;
        rngMultOrder.seed((unsigned)P.runRNGseed);                  // the 2nd pass runs on fresh ReadAlign objects (twoPassRunPass1.cpp:31-37)
        P.dev.chimSegmentMinPositive = P.chim.segmentMin > 0 ? 1 : 0;
        if (P.outFilterBySJout) { bySJoutStage = 1; P.dev.outFilterBySJoutStage = 1; }
        return true;
    }
    // end of the 1st stage of BySJout (STAR.cpp:203-216, outputSJ.cpp:139-161): whitelist = filtered unannotated junctions of
    // ALL reads of stage 1; the held reads become the input of stage 2.  The caller hands the whitelist to the engine.
    bool endStage1() {
        if (bySJoutStage != 1) { error = "not in the 1st stage of BySJout"; return false; }
        sj1.novelWhitelist(P, novelStart, novelEnd);
        sj1.data.clear(); sj1.data.shrink_to_fit();
        reader.openMemory(std::move(heldText[0]), std::move(heldText[1]), (int)P.dev.readNmates);
        P.readMapNumber = -1;
        bySJoutStage = 2; P.dev.outFilterBySJoutStage = 2;
        return true;
    }
    // what the caller has to do after the last batch of a phase: 0 = finish(); 1 = the index changed (staramd_update_index), map
    // again; 2 = the junction whitelist changed (staramd_set_novel_junctions), map again
    int nextPhase() {
        if (pass1) return endPass1() ? 1 : -1;
        if (bySJoutStage == 1) return endStage1() ? 2 : -1;
        return 0;
    }
    bool finish() {
        // SJ.out.tab (collapse of the junction table, filters, half a million lines) does not depend on the last writes of the alignments:
        // it is produced on a thread of its own while the writer drains (outputSJ.cpp:84,129; STAR.cpp:251)
        std::string sjError;
        std::thread sjThread;
        sjBg.drainInto(sj);
        if (!P.outSJnone) sjThread = std::thread([&] { sjError = sj.filterAndWrite(P, gi, P.outFileNamePrefix + "SJ.out.tab", bySJoutStage == 2); });
        struct Join { std::thread &t; ~Join() { if (t.joinable()) t.join(); } } joinSj{sjThread};
        stopWriter();
        for (FILE *&u : unmappedOut) if (u) { fclose(u); u = nullptr; }
        if (chimSamOut) { fclose(chimSamOut); chimSamOut = nullptr; }
        if (quantOut) { std::string e; bgzfEof(e); fwrite(e.data(), 1, e.size(), quantOut); if (quantOut == stdout) fflush(stdout); else fclose(quantOut); quantOut = nullptr; }
        if (chimOut) {
            if (P.chim.outJunctionFormat == 1)              // Stats::writeLines (Stats.cpp:147-155, STAR.cpp:285)
                fprintf(chimOut, "# 2.7.11b   %s\n# Nreads %llu\tNreadsUnique %llu\tNreadsMulti %llu\n", P.commandLine.c_str(), (unsigned long long)stats.readN,
                        (unsigned long long)stats.mappedReadsU, (unsigned long long)stats.mappedReadsM);
            fclose(chimOut); chimOut = nullptr;
        }
        if (writerFailed) { error = "EXITING because of fatal ERROR: could not write Aligned.out.sam"; return false; }
        if (samOut) {
            if (samFd >= 0) { fflush(samOut); fseeko(samOut, (off_t)samPos, SEEK_SET); }      // the batches went out through positional writes: the stream goes on behind them
            if (P.outBAMunsorted) { std::string e; bgzfEof(e); fwrite(e.data(), 1, e.size(), samOut); }
            if (samOut == stdout) fflush(stdout); else fclose(samOut);
            samOut = nullptr;
        }
        if (P.outBAMcoord && !P.outSAMnone) { error = writeSortedBam(); if (!error.empty()) return false; }
        if (sjThread.joinable()) sjThread.join();
        error = sjError;
        if (!error.empty()) return false;
        stats.reportFinal(P.outFileNamePrefix + "Log.final.out");
        for (const char *f : {"Log.out", "Log.progress.out"}) { FILE *l = fopen((P.outFileNamePrefix + f).c_str(), "ab"); if (l) { fputs("ALL DONE!\n", l); fclose(l); } }
        if (P.quantGeneCounts) { error = geneCounts.write(P.outFileNamePrefix + "ReadsPerGene.out.tab", genes, stats); if (!error.empty()) return false; }
        return true;
    }
    ~Runner() { stopWriter(); if (getenv("STARAMD_HOST_TIMING")) fprintf(stderr, "  host stages, whole run: writer %.3f s, emit: wait for a text buffer set %.3f s, format on threads %.3f s, serial tail %.3f s\n", tWriter, tEmitWaitSet, tEmitFormat, tEmitTail); if (samOut && samOut != stdout) fclose(samOut); if (quantOut == stdout) quantOut = nullptr; for (FILE *u : unmappedOut) if (u) fclose(u); if (chimOut) fclose(chimOut); if (quantOut) fclose(quantOut); }
};

} // namespace staramd

using staramd::Runner;

extern "C" {

void *sah_create(int argc, char **argv, char *errbuf, int errlen) {
    Runner *r = new Runner();
    if (!r->init(argc, argv)) {
        if (errbuf && errlen > 0) { strncpy(errbuf, r->error.c_str(), errlen - 1); errbuf[errlen - 1] = 0; }
        delete r;
        return nullptr;
    }
    return r;
}
const staramd_genome *sah_genome(void *h) { return &((Runner *)h)->gi.view; }
const staramd_params *sah_params(void *h) { return &((Runner *)h)->P.dev; }
uint64_t sah_batch_reads(void *h) { return ((Runner *)h)->P.gpuBatchReads; }
// --runMode genomeGenerate: sah_create has scanned the FASTA files; the caller builds SA + SAindex with staramd_index_build
// (include/star_amd_index.h) into the buffers handed out here, then sah_generate_finish inserts the annotated junctions and writes genomeDir.
// junction insertion (2-pass, --sjdbGTFfile / --sjdbFileChrStartEnd at the mapping stage, genomeGenerate with annotations) on the device:
// fn = staramd_sjdb_insert of the engine library (include/star_amd_index.h); process-wide; NULL restores the host restatement
void sah_set_sjdb_device_fn(int (*fn)(int, const staramd_sjdb_args *, staramd_sjdb_result *), int device) { staramd::setSjdbDeviceFn(fn, device); }
// the same on the arrays resident in the engine contexts (staramd_insert_junctions for every context): fn(user, args, result); the front end calls
// sah_engines_ready once its contexts hold the index, and asks sah_index_in_engine after a phase change whether a re-upload is needed at all
void sah_set_sjdb_resident_fn(int (*fn)(void *, const staramd_sjdb_args *, staramd_sjdb_result *), void *user) { staramd::setSjdbResidentFn(fn, user); }
int sah_chim_select_on_device(void *h) {
    Runner *r = (Runner *)h;
    staramd::RunParams &P = r->P;
    const bool mergedMates = P.peOverlapNbasesMin > 0 && P.dev.readNmates == 2;
    // (chimSegmentMinPositive is 0 during the 1st pass of a 2-pass run -- twoPassRunPass1.cpp:24 -- and comes back with the 2nd: the choice is made once, for the run)
    if (!(P.chim.segmentMin > 0 && P.chim.multimapNmax == 0 && !mergedMates && P.dev.resultSelect == 0)) return 0;
    P.dev.resultSelect = 2;
    return 1;
}
void sah_engines_ready(void *h) {
    // every engine context holds the index now, and junction insertion runs on the resident arrays: the host copy of the suffix array (26 GB of a human
    // index, per process -- one process per GPU on a node) is not needed any more.  SAindex and genome stay (small; the SAM writer reads the genome)
    Runner *r = (Runner *)h;
    r->gi.engineHoldsIndex = true;
    // (--sjdbInsertSave All writes the suffix array into _STARgenome, also when no junction was inserted: it stays)
    if (!r->generateMode && !r->P.sjdbInsertSaveAll && !getenv("STARAMD_KEEP_HOST_SA")) { std::vector<uint8_t>().swap(r->gi.SA); r->gi.view.SA = nullptr; }
}
int sah_index_in_engine(void *h) { Runner *r = (Runner *)h; int v = r->gi.indexInEngine ? 1 : 0; r->gi.indexInEngine = false; return v; }
int sah_generate_mode(void *h) { return ((Runner *)h)->generateMode ? 1 : 0; }
int sah_generate_buffers(void *h, const uint8_t **G, uint64_t *nGenome, uint32_t *GstrandBit, uint32_t *saIndexNbases, uint8_t **SA, uint64_t *saCap, uint8_t **SAi, uint64_t *saiCap) {
    Runner *r = (Runner *)h;
    if (!r->generateMode) { r->error = "sah_generate_buffers: not in --runMode genomeGenerate"; return -1; }
    *G = r->gi.G.data(); *nGenome = r->gi.view.nGenome; *GstrandBit = r->genJob.GstrandBit; *saIndexNbases = r->gi.view.gSAindexNbases;
    *SA = r->gi.SA.data(); *saCap = r->gi.SA.size(); *SAi = r->gi.SAi.data(); *saiCap = r->gi.SAi.size();
    return 0;
}
int sah_generate_finish(void *h, uint64_t nSA, uint64_t nSAbyte, uint64_t nSAibyte) {
    Runner *r = (Runner *)h;
    if (!r->generateMode) { r->error = "sah_generate_finish: not in --runMode genomeGenerate"; return -1; }
    if (nSA != r->genJob.nSA || nSAbyte != r->genJob.saBytes || nSAibyte != r->genJob.saiBytes) { r->error = "EXITING because of FATAL problem while generating the suffix array: the device build returned " + std::to_string(nSA) + " indices, expected nSA=" + std::to_string(r->genJob.nSA); return -1; }
    r->error = genomeGenerateFinish(r->P, r->gi, r->genJob, r->sjdbLoci, r->insertLog);
    return r->error.empty() ? 0 : -1;
}
int sah_tool_done(void *h) { return ((Runner *)h)->toolDone ? 1 : 0; }     // 1: the run was a tool mode (--runMode inputAlignmentsFromBAM) and is finished
int sah_device(void *h) { return ((Runner *)h)->P.gpuDevice; }
double sah_genome_load_seconds(void *h) { return ((Runner *)h)->gi.loadSeconds; }
void sah_cpu_add(int stage, uint64_t ns) { staramd::cpuAdd(stage, ns); }
void sah_cpu_seconds(double out[8], int reset) { for (int i = 0; i < staramd::CPU_NSTAGE; i++) out[i] = (double)staramd::cpuTake(i, reset != 0) * 1e-9; }
// 1: a batch of this run can be followed by a second one built from it (merged mates of --peOverlapNbasesMin, allele-swapped reads of --waspOutputMode)
int sah_needs_second_batch(void *h) { Runner *r = (Runner *)h; return ((r->P.peOverlapNbasesMin > 0 && r->P.dev.readNmates == 2) || r->P.wasp) ? 1 : 0; }
void sah_fast_path_counts(void *h, uint64_t out[2]) { Runner *r = (Runner *)h; out[0] = r->nMappedWrites; out[1] = r->reader.mappedBatches.load(); }
void sah_emit_seconds(void *h, double out[4]) { Runner *r = (Runner *)h; out[0] = r->tEmitWaitSet; out[1] = r->tEmitFormat; out[2] = r->tEmitTail; out[3] = r->tWriter; }
int sah_next_batch(void *h, uint64_t maxReads, staramd_batch *out) {
    Runner *r = (Runner *)h;
    int n = r->nextBatch(maxReads);
    if (n > 0 && out) *out = r->batchView;
    return n;
}
int sah_emit(void *h, const staramd_results *res) { return ((Runner *)h)->emit(res) ? 0 : -1; }
// --peOverlapNbasesMin > 0: after sah_next_batch, sah_merged_batch gives the pairs of the batch whose mates overlap, merged into single reads (0 = none);
// map them with the same engine and hand both result sets to sah_emit_merged.  Same for the slots of the pipelined variant.
int sah_merged_batch(void *h, staramd_batch *out) { Runner *r = (Runner *)h; if (!(r->P.peOverlapNbasesMin > 0 && r->P.dev.readNmates == 2) || r->mergedMain.reads.n == 0) return 0; if (out) *out = r->mergedMain.reads.view(); return (int)r->mergedMain.reads.n; }
int sah_emit_merged(void *h, const staramd_results *res, const staramd_results *resMerged) { return ((Runner *)h)->emit(res, resMerged) ? 0 : -1; }
// --waspOutputMode SAMtag: after mapping a batch, sah_wasp_batch builds from its results the allele-swapped reads that WASP maps again (0 = none, still call
// sah_wasp_results with NULL); map them with the same engine, hand the results to sah_wasp_results, then emit as usual.  *_slot: the pipelined variant.
int sah_wasp_batch(void *h, const staramd_results *res, staramd_batch *out) {
    Runner *r = (Runner *)h;
    if (!r->P.wasp || r->pass1) return 0;
    r->waspMain.build(r->P, r->gi, r->variation, r->batch, *res);
    if (out) *out = r->waspMain.reads.view();
    return (int)r->waspMain.reads.n;
}
int sah_wasp_results(void *h, const staramd_results *res, const staramd_results *resWasp) {
    Runner *r = (Runner *)h;
    if (!r->P.wasp || r->pass1) return 0;
    if (r->waspMain.reads.n > 0 && !resWasp) { r->error = "EXITING because of FATAL ERROR: --waspOutputMode: results of the re-mapped reads are missing"; return -1; }
    if (r->waspMain.reads.n > 0) r->waspMain.finish(r->P, r->batch, *res, *resWasp);
    return 0;
}
int sah_wasp_slot(void *h, int slot, const staramd_results *res, staramd_batch *out) {
    Runner *r = (Runner *)h;
    if (!r->P.wasp || r->pass1) return 0;
    r->waspSlots[slot].build(r->P, r->gi, r->variation, r->slots[slot], *res);
    if (out) *out = r->waspSlots[slot].reads.view();
    return (int)r->waspSlots[slot].reads.n;
}
int sah_wasp_results_slot(void *h, int slot, const staramd_results *res, const staramd_results *resWasp) {
    Runner *r = (Runner *)h;
    if (!r->P.wasp || r->pass1) return 0;
    if (r->waspSlots[slot].reads.n > 0) { if (!resWasp) { std::lock_guard<std::mutex> l(r->errM); r->mapError = "EXITING because of FATAL ERROR: --waspOutputMode: results of the re-mapped reads are missing"; return -1; } r->waspSlots[slot].finish(r->P, r->slots[slot], *res, *resWasp); }
    return 0;
}
int sah_merged_slot(void *h, int slot, staramd_batch *out) { Runner *r = (Runner *)h; if (!(r->P.peOverlapNbasesMin > 0 && r->P.dev.readNmates == 2) || r->mergedSlots[slot].reads.n == 0) return 0; if (out) *out = r->mergedSlots[slot].reads.view(); return (int)r->mergedSlots[slot].reads.n; }
int sah_emit_slot_merged(void *h, int slot, const staramd_results *res, const staramd_results *resMerged) {
    Runner *r = (Runner *)h;
    return r->emitBatch(r->slots[slot], res, r->P.peOverlapNbasesMin > 0 && r->P.dev.readNmates == 2 ? &r->mergedSlots[slot] : nullptr, resMerged, r->P.wasp ? &r->waspSlots[slot] : nullptr) ? 0 : -1;
}
// pipelined variant (star_amd CLI): three batch slots so that FASTQ parsing of batch k+1, the device mapping of batch k and
// the post-map / SAM writing of batch k-1 overlap.  parse and emit are each called from ONE thread, in batch order.
int sah_parse_slot(void *h, int slot, uint64_t maxReads, staramd_batch *out) {
    Runner *r = (Runner *)h;
    std::string err;
    bool ok = r->reader.nextBatch(r->slots[slot], r->P, maxReads, err);
    if (!err.empty()) { std::lock_guard<std::mutex> l(r->errM); r->parseError = err; return -1; }
    if (!ok) return 0;
    if (out) *out = r->slots[slot].view();
    if (r->P.peOverlapNbasesMin > 0 && r->P.dev.readNmates == 2) r->mergedSlots[slot].build(r->slots[slot], r->P);
    return (int)r->slots[slot].n;
}
// page-locked (or any other) memory for the numeric arrays of the batches; call once, before the first batch is parsed and never with batches alive
void sah_set_batch_alloc(void *(*alloc)(uint64_t), void (*release)(void *)) { staramd::g_batchAllocFn = alloc; staramd::g_batchFreeFn = release; }
int sah_fill_slot(void *h, int slot, uint64_t maxReads) {
    Runner *r = (Runner *)h;
    std::string err;
    bool ok = r->reader.fillBatch(r->slots[slot], r->P, maxReads, err);
    if (!err.empty()) { std::lock_guard<std::mutex> l(r->errM); r->parseError = err; return -1; }
    return ok ? (int)r->slots[slot].n : 0;
}
int sah_convert_slot(void *h, int slot, staramd_batch *out) {
    Runner *r = (Runner *)h;
    std::string err;
    bool ok = r->reader.convertBatch(r->slots[slot], r->P, err);
    if (!ok || !err.empty()) { std::lock_guard<std::mutex> l(r->errM); r->parseError = err.empty() ? "convertBatch failed" : err; return -1; }
    if (out) *out = r->slots[slot].view();
    if (r->P.peOverlapNbasesMin > 0 && r->P.dev.readNmates == 2) r->mergedSlots[slot].build(r->slots[slot], r->P);
    return (int)r->slots[slot].n;
}
int sah_emit_slot(void *h, int slot, const staramd_results *res) { Runner *r = (Runner *)h; return r->emitBatch(r->slots[slot], res, nullptr, nullptr, r->P.wasp ? &r->waspSlots[slot] : nullptr) ? 0 : -1; }
int sah_threads(void *h) { return ((Runner *)h)->P.runThreadN; }
// 2-pass mapping: sah_in_pass1() is 1 after sah_create when --twopassMode Basic was given; map all batches, call sah_pass1_end()
// (junction insertion on the host), re-upload sah_genome()/sah_params() with staramd_update_index(), map all batches again.
int sah_in_pass1(void *h) { return ((Runner *)h)->pass1 ? 1 : 0; }
uint64_t sah_limit_sjdb_insert(void *h) { return (uint64_t)((Runner *)h)->P.limitSjdbInsertNsj; }
uint32_t sah_sjdb_length(void *h) { Runner *r = (Runner *)h; const uint32_t ov = r->gi.view.sjdbOverhang ? r->gi.view.sjdbOverhang : r->P.sjdbOverhang; return 2 * ov + 1; }
int sah_pass1_end(void *h) { return ((Runner *)h)->endPass1() ? 0 : -1; }
// general form (2-pass and/or --outFilterType BySJout): after the last batch call sah_next_phase(): 0 = done, call sah_finish();
// 1 = the index was rewritten: staramd_update_index(sah_genome(), sah_params()) and map every batch again; 2 = the junction
// whitelist was built: staramd_set_novel_junctions(sah_novel_junctions(...), stage 2) and map every batch again; < 0 = error
int sah_next_phase(void *h) { return ((Runner *)h)->nextPhase(); }
uint64_t sah_novel_junctions(void *h, const uint64_t **start, const uint64_t **end) {
    Runner *r = (Runner *)h;
    *start = r->novelStart.data(); *end = r->novelEnd.data();
    return r->novelStart.size();
}
const char *sah_insert_log(void *h) { return ((Runner *)h)->insertLog.c_str(); }
int sah_finish(void *h) { return ((Runner *)h)->finish() ? 0 : -1; }
// ---- end-of-run exchange between ranks (one process per GPU) -------------------------------------
// The reference merges per-thread junction tables and Stats inside one process (outputSJ.cpp:39-83 k-way merge,
// ReadAlignChunk_mapChunk.cpp:124-127 Stats::addStats).  With one process per GPU the same two reductions go over
// RCCL: every rank exports its COLLAPSED table as fixed 32-byte records + 32 counters, rank 0 imports them and then
// runs the unchanged collapse/filter/write.
struct SjWire { uint64_t start; uint32_t gap; uint32_t countUnique, countMultiple; uint16_t overhangLeft, overhangRight; int8_t strand, motif, annot; uint8_t pad[5]; };
static_assert(sizeof(SjWire) == 32, "junction wire record is 32 bytes");

uint64_t sah_sj_export(void *h, void *buf, uint64_t capRecords) {
    Runner *r = (Runner *)h;
    r->wire().collapse();
    uint64_t n = r->wire().data.size();
    if (!buf) return n;
    if (n > capRecords) return (uint64_t)-1;
    SjWire *w = (SjWire *)buf;
    for (uint64_t i = 0; i < n; i++) {
        const staramd::Junction &j = r->wire().data[i];
        SjWire x; memset(&x, 0, sizeof(x));
        x.start = j.start; x.gap = j.gap; x.countUnique = j.countUnique; x.countMultiple = j.countMultiple;
        x.overhangLeft = j.overhangLeft; x.overhangRight = j.overhangRight; x.strand = j.strand; x.motif = j.motif; x.annot = j.annot;
        w[i] = x;
    }
    return n;
}
int sah_sj_import(void *h, const void *buf, uint64_t nRecords) {
    Runner *r = (Runner *)h;
    const SjWire *w = (const SjWire *)buf;
    for (uint64_t i = 0; i < nRecords; i++) {
        staramd::Junction j; j.start = w[i].start; j.gap = w[i].gap; j.strand = w[i].strand; j.motif = w[i].motif; j.annot = w[i].annot;
        j.countUnique = w[i].countUnique; j.countMultiple = w[i].countMultiple; j.overhangLeft = w[i].overhangLeft; j.overhangRight = w[i].overhangRight;
        r->wire().data.push_back(j);
    }
    return 0;
}
void sah_sj_clear(void *h) { ((Runner *)h)->wire().data.clear(); }
// which table the three calls above address: 0 = the output table, 1 = the table of the 1st BySJout stage (all reads' junctions)
void sah_sj_select(void *h, int which) { ((Runner *)h)->wireTable = which; }
int sah_in_stage1(void *h) { return ((Runner *)h)->bySJoutStage == 1 ? 1 : 0; }
// --quantMode GeneCounts across ranks: 7 counters + 3 x nGe gene counts
uint64_t sah_quant_export(void *h, uint64_t *out, uint64_t cap) {
    Runner *r = (Runner *)h;
    if (!r->P.quantGeneCounts) return 0;
    const staramd::GeneCounts &g = r->geneCounts;
    uint64_t nGe = g.gCount[0].size(), n = 7 + 3 * nGe;
    if (!out) return n;
    if (cap < n) return (uint64_t)-1;
    out[0] = g.cMulti; for (int t = 0; t < 3; t++) { out[1 + t] = g.cAmbig[t]; out[4 + t] = g.cNone[t]; for (uint64_t i = 0; i < nGe; i++) out[7 + t * nGe + i] = g.gCount[t][i]; }
    return n;
}
int sah_quant_import_add(void *h, const uint64_t *in, uint64_t n) {
    Runner *r = (Runner *)h;
    staramd::GeneCounts &g = r->geneCounts;
    uint64_t nGe = g.gCount[0].size();
    if (n != 7 + 3 * nGe) return -1;
    g.cMulti += in[0]; for (int t = 0; t < 3; t++) { g.cAmbig[t] += in[1 + t]; g.cNone[t] += in[4 + t]; for (uint64_t i = 0; i < nGe; i++) g.gCount[t][i] += in[7 + t * nGe + i]; }
    return 0;
}
#define SAH_NSTAT 32
// counters as 32 x u64; mappedPortion (double) travels as its bit pattern and is summed by the importer
int sah_stats_export(void *h, uint64_t *out) {
    const staramd::Stats &s = ((Runner *)h)->stats;
    memset(out, 0, SAH_NSTAT * 8);
    uint64_t v[] = {s.readN, s.readBases, s.mappedMismatchesN, s.mappedInsN, s.mappedDelN, s.mappedInsL, s.mappedDelL, s.mappedBases, s.mappedReadsU,
                    s.mappedReadsM, s.unmappedOther, s.unmappedShort, s.unmappedMismatch, s.unmappedMulti, s.unmappedAll, s.chimericAll, s.splicesNsjdb,
                    s.splicesN[0], s.splicesN[1], s.splicesN[2], s.splicesN[3], s.splicesN[4], s.splicesN[5], s.splicesN[6]};
    for (size_t i = 0; i < sizeof(v) / 8; i++) out[i] = v[i];
    memcpy(&out[24], &s.mappedPortion, 8);
    return SAH_NSTAT;
}
int sah_stats_import_add(void *h, const uint64_t *in) {
    staramd::Stats &s = ((Runner *)h)->stats;
    staramd::Stats a;
    a.readN = in[0]; a.readBases = in[1]; a.mappedMismatchesN = in[2]; a.mappedInsN = in[3]; a.mappedDelN = in[4]; a.mappedInsL = in[5]; a.mappedDelL = in[6];
    a.mappedBases = in[7]; a.mappedReadsU = in[8]; a.mappedReadsM = in[9]; a.unmappedOther = in[10]; a.unmappedShort = in[11]; a.unmappedMismatch = in[12];
    a.unmappedMulti = in[13]; a.unmappedAll = in[14]; a.chimericAll = in[15]; a.splicesNsjdb = in[16];
    for (int i = 0; i < 7; i++) a.splicesN[i] = in[17 + i];
    memcpy(&a.mappedPortion, &in[24], 8);
    s.add(a);
    return 0;
}
const char *sah_error(void *h) {
    Runner *r = (Runner *)h;
    static thread_local std::string mine;
    std::lock_guard<std::mutex> l(r->errM);
    if (!r->parseError.empty()) mine = r->parseError; else if (!r->mapError.empty()) mine = r->mapError; else return r->error.c_str();
    return mine.c_str();
}
void sah_destroy(void *h) { delete (Runner *)h; }

}

// struct sizes of the C ABI, so that language bindings can verify their mirrors
extern "C" uint64_t sah_sizeof(int which) {
    switch (which) {
        case 0: return sizeof(staramd_genome);
        case 1: return sizeof(staramd_params);
        case 2: return sizeof(staramd_batch);
        case 3: return sizeof(staramd_read_result);
        case 4: return sizeof(staramd_transcript);
        case 5: return sizeof(staramd_exon);
        case 6: return sizeof(staramd_results);
        default: return 0;
    }
}
