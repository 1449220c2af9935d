// host.h -- host side of the MI355X STAR hot path: index loader, parameter set, FASTQ batcher,
// post-map selection and SAM / SJ.out.tab / Log.final.out writers.
//
// These are the "boundary glue" rows of SURVEY.md section 2 (4*,5*,6*,7*,8): cheap integer host code
// that must reproduce the reference byte for byte; none of it is accelerated.  Each function
// cites the reference file:line whose behaviour it reproduces.
#pragma once
#include <ctime>
#include <cstdint>
#include <string>
#include <vector>
#include <atomic>
#include <array>
#include <cstdio>
#include <istream>
#include <string_view>
#include "../../../include/star_amd.h"
#include "../../../include/star_amd_index.h"

namespace staramd {
// CPU seconds per pipeline stage (thread CPU clocks: what a stage COSTS, beside the wall-clock busy times that say how long it TAKES): every thread of the front end
// carries exactly one CpuScope for its whole life -- the five stage threads of cli_run.cpp and each helper thread a stage starts for a batch (sah_cpu_seconds)
enum { CPU_FILL = 0, CPU_CONVERT, CPU_MAP, CPU_EMIT, CPU_WRITE, CPU_OTHER, CPU_NSTAGE = 8 };
void cpuAdd(int stage, uint64_t ns);
uint64_t cpuTake(int stage, bool reset);
struct CpuScope {
    int stage; timespec t0;
    explicit CpuScope(int st) : stage(st) { clock_gettime(CLOCK_THREAD_CPUTIME_ID, &t0); }
    ~CpuScope() { timespec t1; clock_gettime(CLOCK_THREAD_CPUTIME_ID, &t1); cpuAdd(stage, (uint64_t)((t1.tv_sec - t0.tv_sec) * 1000000000ll + (t1.tv_nsec - t0.tv_nsec))); }
};

// ---- genomeDir on disk -> host arrays (Genome::genomeLoad, source/Genome_genomeLoad.cpp:18-467) ----
struct GenomeIndex {
    std::string dir;
    std::vector<uint8_t> G, SA, SAi;
    std::vector<uint64_t> chrStart, chrLength, sjDstart, sjAstart, sjdbStart, sjdbEnd;
    std::vector<uint32_t> chrBin;
    std::vector<uint8_t> sjdbMotif, sjdbShiftLeft, sjdbShiftRight, sjdbStrand;
    std::vector<std::string> chrName;
    staramd_genome view;      // pointers into the vectors above
    double loadSeconds = 0;
    bool sjdbInfoExists = false;
    bool engineHoldsIndex = false;   // set by the front end once every engine context holds this index (staramd_create): insertion may then run on the resident arrays
    bool indexInEngine = false;      // the last junction insertion ran on the resident arrays: the engines need staramd_update_tables, not a re-upload
    // returns empty string on success, else the error text
    std::string load(const std::string &genomeDir);
    void refreshView();       // re-point `view` at the vectors after they were rewritten (sjdb insertion)
};

// ---- Parameters: defaults of source/parametersDefault + command-line subset ----
// ---- chimeric detection parameters (source/parametersDefault:682-729, ParametersChimeric_initialize.cpp) ----
struct ChimParams {
    uint64_t segmentMin = 0, junctionOverhangMin = 20, segmentReadGapMax = 0, mainSegmentMultNmax = 10;
    int scoreMin = 0, scoreDropMax = 20, scoreSeparation = 10, scoreJunctionNonGTAG = -1;
    bool filterGenomicN = true;
    int outJunctionFormat = 0;
    uint64_t multimapNmax = 0, multimapScoreRange = 1, nonchimScoreDropMin = 20;   // --chimMultimapNmax > 0: the multimapping algorithm
    bool outSamOld = false;                                                         // --chimOutType SeparateSAMold
    bool outJunctions = true, outBam = false, bamHardClip = true;                   // --chimOutType Junctions | WithinBAM [HardClip | SoftClip]
};

struct WigParams {                     // --outWigType / --outWigStrand / --outWigNorm / --outWigReferencesPrefix (Parameters.cpp:511-560)
    bool yes = false; int format = 0, type = 0, norm = 1; bool strand = true; std::string referencesPrefix;
};

struct RunParams {
    staramd_params dev;                 // what reaches the device hot path
    // run
    std::string genomeDir, outFileNamePrefix = "./";
    std::vector<std::string> readFilesIn;   // one entry per mate; each may be a comma-separated list of files (read in turn)
    std::string readFilesCommand;        // --readFilesCommand, "" = read the files directly
    int runThreadN = 1;
    int64_t readMapNumber = -1;
    std::string commandLine;
    // host-side filters / output (defaults: parametersDefault:455-530, 241, 266-290)
    uint32_t outFilterMultimapNmax = 10;
    uint32_t outFilterMismatchNmax = 10;
    double outFilterMismatchNoverReadLmax = 1.0;
    int32_t outFilterScoreMin = 0;
    double outFilterScoreMinOverLread = 0.66;
    uint32_t outFilterMatchNmin = 0;
    double outFilterMatchNminOverLread = 0.66;
    int32_t outSJfilterOverhangMin[4] = {30, 12, 12, 12};
    int32_t outSJfilterCountUniqueMin[4] = {3, 1, 1, 1};
    int32_t outSJfilterCountTotalMin[4] = {3, 1, 1, 1};
    int32_t outSJfilterDistToOtherSJmin[4] = {10, 0, 5, 10};
    std::vector<uint64_t> outSJfilterIntronMaxVsReadN = {50000, 100000, 200000};
    bool outSJfilterReadsUnique = false;   // outSJfilterReads All | Unique
    int outSAMmapqUnique = 255;
    int outSAMattrIHstart = 1;
    uint32_t outSAMflagOR = 0, outSAMflagAND = 65535;
    bool outSAMunmappedWithin = false;
    bool outSAMprimaryAllBest = false;
    bool outSAMmodeNoQS = false;
    bool outBAMunsorted = false, outBAMcoord = false; bool outSAMnone = false; int outBAMcompression = 1;   // --outSAMtype BAM Unsorted | None, --outBAMcompression
    std::vector<std::string> outSAMattrOrder = {"NH", "HI", "AS", "nM"};   // Standard
    bool attrNMorMD = false, attrHasCh = false;
    std::vector<std::string> outSAMattrOrderQuant;   // attributes of Aligned.toTranscriptome.out.bam: NH HI, then RG / MC if requested (Parameters_samAttributes.cpp:43-47,96-111)
    std::string readNameSeparator = "/";
    uint64_t gpuBatchReads = 65536;      // reads per device batch (ours; --gpuBatchReads)
    int gpuDevice = 0;
    // 2-pass mapping and junction insertion at the mapping stage (Parameters.cpp:779-826, 1000-1034)
    bool twopass = false;                // --twopassMode Basic
    int64_t twopass1readsN = -1; bool twopass1Set = false;
    std::vector<std::string> sjdbFileChrStartEnd;
    std::string sjdbGTFfile, sjdbGTFchrPrefix = "-", sjdbGTFfeatureExon = "exon", sjdbGTFtagExonParentTranscript = "transcript_id", sjdbGTFtagExonParentGene = "gene_id";
    std::vector<std::string> sjdbGTFtagExonParentGeneName = {"gene_name"}, sjdbGTFtagExonParentGeneType = {"gene_type", "gene_biotype"};
    uint32_t sjdbOverhang = 100; bool sjdbOverhangSet = false;
    bool sjdbInsertSaveAll = false;      // --sjdbInsertSave Basic | All
    uint64_t limitSjdbInsertNsj = 1000000;
    std::string sjdbInsertOutDir, twopassDir;
    bool sjdbInsertPass1() const { return !sjdbFileChrStartEnd.empty() || !sjdbGTFfile.empty(); }
    bool sjdbInsertYes() const { return twopass || sjdbInsertPass1(); }
    bool outFilterBySJout = false;       // --outFilterType BySJout
    ChimParams chim;                     // --chim* (chimeric.cpp)
    WigParams wig;                       // --outWig* (signal.cpp)
    std::vector<std::string> outSAMattrRG, outSAMattrRGlineSplit;   // --outSAMattrRGline (Parameters_readFilesInit.cpp:64-93)
    bool outReadsUnmappedFastx = false;  // --outReadsUnmapped Fastx
    bool outSAMreadIDnumber = false;     // --outSAMreadID Number
    int outSAMtlen = 1;                  // --outSAMtlen 1 | 2
    int64_t outSAMmultNmax = -1;         // --outSAMmultNmax
    bool quantGeneCounts = false;        // --quantMode GeneCounts
    bool quantTrSAM = false, quantTrIndel = false, quantTrSoftClip = false, quantTrSingleEnd = false;   // --quantMode TranscriptomeSAM, --quantTranscriptomeSAMoutput
    int quantTrBAMcompression = 1;       // --quantTranscriptomeBAMcompression
    int runRNGseed = 777;                // --runRNGseed
    // read clipping, Hamming adapter type (ParametersClip_initialize.cpp, ClipMate_clip.cpp): per mate
    struct ClipEnd { bool active = false; uint32_t N = 0, NafterAd = 0; std::string adSeq; double adMMp = 0; } clip[2][2];   // [mate][0 = 5', 1 = 3']
    bool clipYes = false;
    std::string readFilesPrefix, readFilesManifest;         // --readFilesPrefix, --readFilesManifest (Parameters_readFilesInit.cpp:41-139)
    std::vector<std::string> outSAMheaderHD, outSAMheaderPG; std::string outSAMheaderCommentFile;   // samHeaders.cpp:56-96
    bool runDirPermAll = false, genomeLoadShared = false;
    // --runMode genomeGenerate (genome_generate.cpp): FASTA -> genomeDir, suffix array + SAindex built on the device
    bool runModeGenerate = false; std::vector<std::string> genomeFastaFiles; uint32_t genomeSAindexNbases = 14, genomeChrBinNbits = 18, genomeSAsparseD = 1;
    bool runModeFromBAM = false; std::string inputBAMfile;   // --runMode inputAlignmentsFromBAM --inputBAMfile: signal tracks from an existing BAM, no mapping
    std::string varVCFfile; bool varHeteroOnly = false, wasp = false; const struct Variation *var = nullptr;   // --varVCFfile, --waspOutputMode SAMtag (Parameters.cpp:854-890)
    int readFilesSAMmates = 0;           // --readFilesType SAM SE | PE: 1 | 2 (0 = Fastx)
    bool samAttrKeepAll = true, samAttrKeepNone = false; std::vector<std::string> samAttrKeep;   // --readFilesSAMattrKeep (BAM output only, Parameters_readFilesInit.cpp:13-31)
    uint32_t peOverlapNbasesMin = 0; double peOverlapMMp = 0.01;   // --peOverlapNbasesMin, --peOverlapMMp
    bool outSAMorderKeep = false;        // --outSAMorder PairedKeepInputOrder
    bool outSJnone = false;              // --outSJtype None
    int outQSconversionAdd = 0;          // --outQSconversionAdd (readLoad.cpp:71-82)
    bool outMultimapperRandom = false;   // --outMultimapperOrder Random (ReadAlign_multMapSelect.cpp:62-92)
    bool outSAMunmappedKeepPairs = false;   // --outSAMunmapped Within KeepPairs
    std::string outStd = "Log";          // --outStd Log | SAM | BAM_Unsorted | BAM_SortedByCoordinate | BAM_Quant

    RunParams();
    // STAR-style "--name v1 v2 ..." ; returns error text or ""
    std::string parse(int argc, char **argv);
    // derive window parameters from the genome (Genome_genomeLoad.cpp:382-410)
    void finalize(const GenomeIndex &gi);
};

// Text of a batch: a vector whose resize() does not write zeros into the ~100 MB that a read from the input file overwrites at once
template <class T> struct NoInitAlloc : std::allocator<T> {
    template <class U> struct rebind { typedef NoInitAlloc<U> other; };
    NoInitAlloc() = default;
    template <class U> NoInitAlloc(const NoInitAlloc<U> &) {}
    template <class U> void construct(U *p) { ::new ((void *)p) U; }                                  // default-initialised: left as it is
    template <class U, class... A> void construct(U *p, A &&...a) { ::new ((void *)p) U(std::forward<A>(a)...); }
};
typedef std::vector<char, NoInitAlloc<char>> TextBuf;

// The numeric arrays of a batch are what staramd_map_batch copies to the device: with a caller that provides page-locked memory (sah_set_batch_alloc: the star_amd CLI
// passes staramd_pinned_alloc / staramd_pinned_free) the copies are DMA transfers straight out of them instead of being staged through the runtime (SURVEY.md 8b:
// "caller-owned pinned buffers").  The hooks are process-wide, set once before the first batch is allocated; without them: malloc.  Elements are left uninitialised by resize().
extern void *(*g_batchAllocFn)(uint64_t bytes);
extern void (*g_batchFreeFn)(void *p);
void *batchAllocate(size_t bytes);       // through the hook; ordinary memory when there is no hook or the hook has none left (reads.cpp)
void batchRelease(void *p);
template <class T> struct BatchAlloc {
    typedef T value_type;
    template <class U> struct rebind { typedef BatchAlloc<U> other; };
    BatchAlloc() = default;
    template <class U> BatchAlloc(const BatchAlloc<U> &) {}
    T *allocate(size_t n) { void *p = batchAllocate(n > 0 ? n * sizeof(T) : 1); if (!p) throw std::bad_alloc(); return (T *)p; }
    void deallocate(T *p, size_t) { batchRelease(p); }
    template <class U> void construct(U *p) { ::new ((void *)p) U; }
    template <class U, class... A> void construct(U *p, A &&...a) { ::new ((void *)p) U(std::forward<A>(a)...); }
    template <class U> bool operator==(const BatchAlloc<U> &) const { return true; }
    template <class U> bool operator!=(const BatchAlloc<U> &) const { return false; }
};


// ---- one batch of reads in the layout of staramd_batch + the text needed for SAM ----
struct TextSpan { uint64_t off; uint32_t len; };
struct ReadBatch {
    uint32_t n = 0;
    std::vector<uint8_t, BatchAlloc<uint8_t>> bases;    // combined numeric reads (every byte is written by the parser: no zero fill on resize)
    std::vector<uint64_t, BatchAlloc<uint64_t>> readOffset;     // n+1
    std::vector<uint16_t, BatchAlloc<uint16_t>> mate1Length, mmMaxTotal;
    TextBuf text[2];                      // the FASTQ text of the batch as read from each mate file; the spans below point into it
    const char *mapped[2] = {nullptr, nullptr};   // ... or, for a regular input file, the batch's part of the MAPPING of the file (FastqReader): nothing is copied, text[] stays empty
    const char *txt(int m) const { return mapped[m] ? mapped[m] : text[m].data(); }
    std::vector<TextSpan> nameSpan;       // read ID without '@', trimmed at readNameSeparator (from mate 1's ID line)
    std::vector<TextSpan> seqSpan[2], qualSpan[2];
    std::vector<char> filter;             // 'Y'/'N' Illumina pass-filter field
    std::vector<uint16_t> clipN[2][2];    // [mate][0 = 5', 1 = 3']: bases clipped off before mapping (--clip*); empty when nothing is clipped
    uint32_t clipped(int m, int p, uint32_t i) const { return clipN[m][p].empty() ? 0u : clipN[m][p][i]; }
    uint64_t firstReadIndex = 0;
    std::vector<uint64_t> origIndex;      // 2nd stage of BySJout: index of the read in the original input (empty otherwise)
    uint32_t fileIndex = 0;               // which of the comma-separated input files the batch came from (a batch never spans two)
    std::vector<uint32_t> heldFile;       // 2nd stage of BySJout: the file every held read came from (empty otherwise)
    uint32_t fileOf(uint32_t i) const { return heldFile.empty() ? fileIndex : heldFile[i]; }
    std::vector<uint64_t> lineStart[2], lineEnd[2];   // line table of text[]: written by the reader's fill stage, turned into the spans above by its convert stage
    std::vector<TextSpan> extraSpan[2];   // SAM input: the attributes of the input record, tab-separated text (readNameExtra, readLoad.cpp:28-29); empty otherwise
    std::string_view extra(int m, uint32_t i) const { return extraSpan[m].empty() ? std::string_view() : std::string_view(txt(m) + extraSpan[m][i].off, extraSpan[m][i].len); }
    bool fasta = false;                   // the reads came without qualities (FASTA input, readLoad.cpp:84-88): QUAL is * in SAM, 0xFF in BAM, Fastx output is FASTA
    uint64_t readIndex(uint32_t i) const { return origIndex.empty() ? firstReadIndex + i : origIndex[i]; }
    std::string_view name(uint32_t i) const { return std::string_view(txt(0) + nameSpan[i].off, nameSpan[i].len); }
    std::string_view seq(int m, uint32_t i) const { return std::string_view(txt(m) + seqSpan[m][i].off, seqSpan[m][i].len); }
    std::string_view qual(int m, uint32_t i) const { return std::string_view(txt(m) + qualSpan[m][i].off, qualSpan[m][i].len); }
    staramd_batch view() const;
    void clear();
};

// variation.cpp: the SNVs of --varVCFfile, sorted by genome coordinate; nt[i] = {reference, allele 1, allele 2} as 0..3
struct VarOverlap { std::vector<uint32_t> ind, readCoord; std::vector<int32_t> genCoord; std::vector<char> allele; };   // allele: 1 / 2 = matches that allele, 3 neither, 4 = N in the read
struct Variation {
    std::vector<uint64_t> loci; std::vector<std::array<uint8_t, 3> > nt;
    std::string load(const RunParams &P, const GenomeIndex &gi);
    void overlap(const staramd_transcript &t, const staramd_exon *ex, const uint8_t *Read1, uint64_t Lread, uint64_t chrStart, VarOverlap &o) const;
};
struct ReadBatch;
// --waspOutputMode SAMtag: the allele-swapped copies of the reads of a batch that WASP maps again (one more batch), and the verdict (vW) per read
struct WaspBatch;

// --peOverlapNbasesMin > 0: the pairs of a batch whose mates overlap, merged into single-end reads that are mapped as a second batch
// (ReadAlign::peMergeMates, ReadAlign_peOverlapMergeMap.cpp:77-134).  index[i] = position of pair i in `reads`, or -1
struct MergedBatch {
    ReadBatch reads;
    std::vector<int32_t> index; std::vector<uint32_t> nOv; std::vector<std::array<uint32_t, 2> > mateStart;
    void build(const ReadBatch &b, const RunParams &P);
};

struct WaspBatch {
    ReadBatch reads;
    std::vector<uint32_t> first, count; std::vector<int8_t> type;
    void build(const RunParams &P, const GenomeIndex &gi, const Variation &var, const ReadBatch &b, const staramd_results &r);
    void finish(const RunParams &P, const ReadBatch &b, const staramd_results &r, const staramd_results &rw);
};

class FastqReader {
public:
    ~FastqReader();
    // readCommand: --readFilesCommand (e.g. "zcat", "gunzip -c"); the text then comes from a pipe (Parameters_openReadsFiles.cpp:23-96)
    // paths[m] = comma-separated list for mate m; samMates > 0: the one list holds SAM text with samMates (1 or 2) consecutive records per read (--readFilesType SAM SE|PE)
    std::string open(const std::vector<std::string> &paths, const std::string &readCommand = "", int samMates = 0);
    void openMemory(std::string mate1, std::string mate2, int nMatesIn);   // FASTQ text held in memory (2nd stage of BySJout)
    std::string reopen();                 // rewind to the first read (Parameters::closeReadsFiles/openReadsFiles between the two passes)
    // mimics ReadAlignChunk::processChunks FASTQ branch (:111-157) + readLoad (readLoad.cpp:4-100)
    // + the PE concatenation of ReadAlign::oneRead (ReadAlign_oneRead.cpp:35-78).
    // The text is read in blocks and the records of a batch are converted on --runThreadN threads.
    bool nextBatch(ReadBatch &b, const RunParams &P, uint64_t maxReads, std::string &err) { return fillBatch(b, P, maxReads, err) && convertBatch(b, P, err); }
    // the two halves of nextBatch, for callers that run them as two pipeline stages on two threads (star_amd CLI): fillBatch reads the text of the next batch
    // and builds its line table (everything that touches the input and decides where the batch ends); convertBatch turns text + line table into numeric
    // reads and spans and touches nothing that fillBatch of the NEXT batch uses.  Each half is called from one thread, in batch order.
    bool fillBatch(ReadBatch &b, const RunParams &P, uint64_t maxReads, std::string &err);
    bool convertBatch(ReadBatch &b, const RunParams &P, std::string &err);
    uint64_t readsSoFar = 0;
    std::atomic<uint64_t> mappedBatches{0};          // batches handed out as ranges of the input files' mappings (reads.cpp fillMapped)
private:
    FILE *f[2] = {nullptr, nullptr};
    unsigned readSlices = 4;                         // slices a block of a regular input file is read in (threads per mate): --runThreadN / 2 in [4, 16], STARAMD_READ_SLICES
    std::atomic<uint64_t> slicedBlocks{0};           // blocks of a regular file read in slices on threads (reads.cpp fill)
    int nMates = 0;
    std::vector<std::string> paths_; std::string command_;
    std::vector<std::string> files_[2]; size_t curFile = 0;
    std::string openCurrent();
    void closeFiles();
    std::string mem[2]; size_t memPos[2] = {0, 0}; bool fromMemory = false;
    std::vector<char> carry[2];           // text read from the file but not yet part of a batch
    // A regular, uncompressed input file is MAPPED: a batch is a range of the mapping (ReadBatch::mapped), found by scanning for line ends in place -- the ~250 bytes per read
    // and mate that the block reads copied out of the page cache are not copied at all.  Mappings live until the reader is reopened or destroyed (batches in flight point
    // into them).  Not for pipes (--readFilesCommand), SAM / FASTA input, or options that rewrite the text of a batch in place (--outQSconversionAdd, --outSAMreadID Number).
    struct Mapped { const char *p = nullptr; size_t n = 0; };
    std::vector<Mapped> allMaps; Mapped curMap[2]; size_t mapPos[2] = {0, 0}; int useMap = -1;      // useMap: -1 undecided, 0 copy, 1 mapping
    void dropMaps();
    uint64_t fillMapped(int m, uint64_t want, ReadBatch &b);
    bool eof[2] = {false, false};
    std::atomic<int> ioError{0};          // errno of a failed read of an input file (sliced pread path): reported by nextBatch, never taken for the end of the input
    double bytesPerRecord[2] = {512, 512};   // running estimate, sizes the next block read
    std::vector<uint64_t> lineRaw[2];     // positions of the newline bytes of the batch text (fill): the line table is built from them on threads
    bool noQualities = false;             // held FASTA reads (2nd stage of BySJout)
    int samMates_ = 0; bool extras = false;   // SAM text input (ReadAlignChunk_processChunks.cpp:28-107); ID lines may carry attributes after a \x01
    TextBuf samText2; std::vector<uint64_t> samLs2, samLe2;   // mate 2 of the records fillSam parsed for mate 1
    uint64_t fillSam(uint64_t want, ReadBatch &b);
    std::string samError; uint64_t firstFlag = 0; std::string lastExtra[2];
    bool fasta = false;                   // '>' records, possibly with the sequence over several lines (ReadAlignChunk_processChunks.cpp:158-190)
    uint64_t fillFasta(int m, uint64_t want, ReadBatch &b);
    // moves text of up to `want` records into b.text[m]; fills b.lineStart[m] / b.lineEnd[m]; returns the number of complete lines
    uint64_t fill(int m, uint64_t want, ReadBatch &b);
};

// ---- junction insertion into the loaded index (sjdb_insert.cpp) ----
struct SjdbLoci { std::vector<std::string> chr; std::vector<uint64_t> start, end; std::vector<char> str; std::vector<uint8_t> priority; };
void sjdbLoadFromStream(std::istream &in, SjdbLoci &loci);            // sjdbLoadFromStream.cpp:2-28
// sjdbInsertJunctions.cpp:11-102: rewrites gi (G, SA, SAi, junction table) and P.dev.winBinN; returns error text or ""
std::string sjdbInsertJunctions(RunParams &P, GenomeIndex &gi, SjdbLoci &loci, bool pass2, const std::string &pass1sjFile, std::string &log);
std::string makeRunDir(const std::string &d, bool allRWX = false);
// junction insertion on the device: the front end hands in staramd_sjdb_insert (include/star_amd_index.h); null = host restatement
typedef int (*SjdbDeviceFn)(int device, const staramd_sjdb_args *a, staramd_sjdb_result *res);
void setSjdbDeviceFn(SjdbDeviceFn fn, int device);
typedef int (*SjdbResidentFn)(void *user, const staramd_sjdb_args *a, staramd_sjdb_result *res);
void setSjdbResidentFn(SjdbResidentFn fn, void *user);
// --sjdbGTFfile at the mapping stage (gtf.cpp): junctions of the annotation appended to `loci` with priority 20
std::string loadGTFjunctions(const RunParams &P, const GenomeIndex &gi, SjdbLoci &loci, const std::string &dirOut, std::string &log);

// ---- --runMode genomeGenerate (genome_generate.cpp; Genome::genomeGenerate, source/Genome_genomeGenerate.cpp:96-416) ----
// scan: FASTA -> gi.G / chromosome tables (genomeScanFastaFiles.cpp:5-92), chr*.txt written, index geometry decided; SA / SAi of `gi`
// are sized for the device build (staramd_index_build, include/star_amd_index.h), which the caller runs between the two calls.
struct GenerateJob { uint32_t GstrandBit = 0; uint64_t nSA = 0, saBytes = 0, saiBytes = 0; uint64_t saiStart[17] = {0}; };
std::string genomeGenerateScan(RunParams &P, GenomeIndex &gi, GenerateJob &job);
std::string genomeGenerateFinish(RunParams &P, GenomeIndex &gi, GenerateJob &job, SjdbLoci &loci, std::string &log);

// ---- Stats (source/Stats.{h,cpp}) ----
struct Stats {
    uint64_t readN = 0, readBases = 0, mappedMismatchesN = 0, mappedInsN = 0, mappedDelN = 0, mappedInsL = 0, mappedDelL = 0,
             mappedBases = 0, mappedReadsU = 0, mappedReadsM = 0, unmappedOther = 0, unmappedShort = 0, unmappedMismatch = 0,
             unmappedMulti = 0, unmappedAll = 0, chimericAll = 0, splicesNsjdb = 0, splicesN[7] = {0, 0, 0, 0, 0, 0, 0};
    double mappedPortion = 0;
    time_t timeStart = 0, timeStartMap = 0, timeFinish = 0;
    void add(const Stats &s);
    void reportFinal(const std::string &path);   // Stats::reportFinal, Stats.cpp:99-145
};

// ---- junction table (source/OutSJ.{h,cpp}, ReadAlign_outputTranscriptSJ.cpp, outputSJ.cpp) ----
struct Junction { uint64_t start; uint32_t gap; int8_t strand, motif, annot; uint32_t countUnique, countMultiple; uint16_t overhangLeft, overhangRight; };
struct OutSJ {
    std::vector<Junction> data;
    void collapse();                                 // OutSJ::collapseSJ, OutSJ.cpp:42-72
    void mergeFrom(const OutSJ &o) { data.insert(data.end(), o.data.begin(), o.data.end()); }
    std::vector<Junction> filtered(const RunParams &P, bool skipDistanceFilter);
    void novelWhitelist(const RunParams &P, std::vector<uint64_t> &start, std::vector<uint64_t> &end);
    // outputSJ filter + write (outputSJ.cpp:56-138); returns error text or ""
    std::string filterAndWrite(const RunParams &P, const GenomeIndex &gi, const std::string &path, bool skipDistanceFilter = false);
};

// ---- --quantMode GeneCounts (quant.cpp) ----
struct GeneAnnotation {                   // geneInfo.tab + exonGeTrInfo.tab of the genome (or of _STARgenome/ with --sjdbGTFfile)
    std::vector<std::string> geID;
    std::vector<uint64_t> s, e, eMax; std::vector<uint8_t> str; std::vector<uint32_t> g;
    std::string load(const std::string &dir);
};
struct GeneCounts {
    uint64_t cMulti = 0, cAmbig[3] = {0, 0, 0}, cNone[3] = {0, 0, 0};
    std::vector<uint64_t> gCount[3];
    GeneCounts() {}
    explicit GeneCounts(size_t nGe);
    void add(const GeneCounts &o);
    void addAlign(const GeneAnnotation &A, uint64_t nA, const staramd_transcript &a, const staramd_exon *ex);   // Transcriptome_geneCountsAddAlign.cpp:4-63
    std::string write(const std::string &path, const GeneAnnotation &A, const Stats &st) const;                  // Transcriptome.cpp:158-190
};

// ---- --quantMode TranscriptomeSAM (quant.cpp) ----
struct GenomicAlign { uint32_t nExons; uint32_t Str; uint64_t Lread; staramd_exon ex[STARAMD_MAX_N_EXONS]; };   // one alignment in genome coordinates (a working copy)
struct ProjectedAlign { uint32_t tr; uint32_t Str; uint32_t nExons; staramd_exon ex[STARAMD_MAX_N_EXONS]; };      // the same alignment on transcript `tr`
struct TranscriptAnnotation {             // transcriptInfo.tab + exonInfo.tab
    std::vector<std::string> trID;
    std::vector<uint64_t> trS, trE, trEmax; std::vector<uint8_t> trStr; std::vector<uint16_t> trExN; std::vector<uint32_t> trExI, trLen;
    std::vector<uint32_t> exSE, exLenCum;
    std::string load(const std::string &dir);
    uint32_t quantAlign(const GenomicAlign &aG, std::vector<ProjectedAlign> &out) const;
};
// primary flag of the transcriptomic alignments of one read: chosen with one random number per mapped read, in read order
// (ReadAlign_quantTranscriptome.cpp:69), so it is patched into the records after the threads of a batch have joined
struct QuantPatch { uint32_t ir; uint32_t nAlignT; std::vector<uint64_t> recOffset; std::vector<uint32_t> recAlign; };

// ---- BGZF framing of BAM output (bgzf.cpp) ----
bool bgzfCompress(const std::string &raw, int level, std::string &out);
void bgzfEof(std::string &out);

// sort key of one BAM record (BAMoutput::coordOneAlign, BAMbinSortByCoordinate.cpp:45-56): (refID << 32 | pos, read order, order of production)
struct BamKey { uint64_t g, r; uint64_t off; uint32_t len; uint32_t chunk; };

struct ReadBatch;
// chimeric.cpp: ReadAlign::chimericDetectionOld + the Chimeric.out.junction line; true = a chimeric alignment was recorded
// one segment of a chimeric alignment: a copy of the alignment whose block next to the chimeric junction was cut / extended to it
struct ChimTr { staramd_transcript t; staramd_exon ex[STARAMD_MAX_N_EXONS]; };
struct ChimPair { ChimTr a1, a2; bool best; VarOverlap var1, var2; };   // var1/2: the SNVs under the two alignments as they were BEFORE the junction shift (what the reference's copies carry)   // the two segments, in read order; best = the top-scoring chimera of the read (the primary one in the BAM)
// the recorded alignments of one read, window by window, best first in each: T[k].exonOffset indexes ex
// pre: the partner loop of chimeric detection was run by the engine (staramd_params::resultSelect 2) -- its outcome; partner = index into T, -1 none
struct ChimPre { int32_t partner; int scoreBest, scoreNext; uint32_t strBest; };
struct ReadAligns { const staramd_transcript *T; uint32_t nTr; const staramd_exon *ex; const ChimPre *pre = nullptr; };
bool chimericDetectionOld(const RunParams &P, const GenomeIndex &gi, const ReadBatch &b, uint32_t ir, const ReadAligns &ra,
                          const staramd_transcript *trBest, uint64_t nTr, const staramd_transcript *trMult0, const staramd_transcript *trMult1, std::string &out,
                          std::vector<ChimPair> *bamOut = nullptr, const ReadBatch *nameBatch = nullptr, uint32_t nameIr = 0, const uint32_t *mateStart = nullptr);

// --outMultimapperOrder Random: the swap partners of the two Fisher-Yates shuffles of every multimapping read of a batch, drawn in read
// order from the run's one random stream before the batch is formatted on threads (ReadAlign_multMapSelect.cpp:71-80)
struct MultOrder { std::vector<uint64_t> offset; std::vector<uint32_t> partner; std::vector<uint32_t> quantPick; };

// nameBatch / nameIr: where the read's name, read group and unclipped lengths come from when b holds merged mates (PEmerged_bool = 1)
bool chimericDetectionMult(const RunParams &P, const GenomeIndex &gi, const ReadBatch &b, uint32_t ir, const ReadAligns &ra, const staramd_transcript *trBest, std::string &out,
                           std::vector<ChimPair> *bamOut = nullptr, const ReadBatch *nameBatch = nullptr, uint32_t nameIr = 0, const uint32_t *mateStart = nullptr);
// Transcript::peOverlapSEtoPE: an alignment of merged mates cut into the blocks of the two mates (scores not set); false = too many blocks
bool mergedAlignToPair(ChimTr &o, const uint32_t mateStart[2], const staramd_transcript &t, const staramd_exon *tex, uint64_t tLread, const uint64_t readLength[2], uint64_t Lread);
// Transcript::alignScore: score and mismatches of an alignment recomputed from its blocks; Read1 = the read as mapped, Lread its length
int chimAlignScore(const staramd_params &D, const GenomeIndex &gi, const uint8_t *Read1, uint64_t Lread, ChimTr &c);

// signal.cpp: coverage tracks from BAM records in coordinate order; error text or ""
std::string writeSignal(const RunParams &P, const std::vector<std::string> &chrName, const std::vector<uint64_t> &chrLength, const std::string &sigFileName, const std::vector<const char *> &recs);
std::string signalFromBamFile(const RunParams &P, const std::string &bamPath, const std::string &sigFileName);   // --runMode inputAlignmentsFromBAM

// ---- post-map: multMapSelect, mappedFilter, outputAlignments ----
class PostMap {
public:
    PostMap(const RunParams &P, const GenomeIndex &gi) : P(P), gi(gi) {}
    // consumes the device (or oracle) results of one batch; appends SAM text to `sam`
    std::string process(const ReadBatch &b, const staramd_results &r, std::string &sam, OutSJ &sj, Stats &st);
    // bamKeys (with --outSAMtype BAM SortedByCoordinate): one entry per BAM record appended to `sam`
    // sj1 / held: 1st stage of --outFilterType BySJout (ReadAlign_outputAlignments.cpp:90-124): junctions of every read go to sj1,
    // reads with an unannotated junction are not output but listed in `held` for the 2nd stage
    // what one range of reads adds to (per-thread buffers of the caller; null = that output is off)
    struct RangeOut {
        std::string *sam = nullptr;                      // SAM text or raw BAM records of the alignments
        OutSJ *sj = nullptr; Stats *st = nullptr;
        OutSJ *sj1 = nullptr; std::vector<uint32_t> *held = nullptr;   // 1st stage of --outFilterType BySJout (ReadAlign_outputAlignments.cpp:90-124): junctions of every read, reads held for the 2nd stage
        GeneCounts *gc = nullptr;                        // --quantMode GeneCounts
        std::vector<BamKey> *bamKeys = nullptr;          // --outSAMtype BAM SortedByCoordinate: one key per record in *sam
        std::string *unmappedFastx = nullptr;            // [2]: --outReadsUnmapped Fastx text per mate
        std::string *chimJunction = nullptr;             // Chimeric.out.junction lines (--chimSegmentMin > 0); non-null switches the detection on
        std::string *chimSam = nullptr;                  // Chimeric.out.sam records (--chimOutType SeparateSAMold)
        std::string *quantBam = nullptr; std::vector<QuantPatch> *quantPatches = nullptr;   // --quantMode TranscriptomeSAM records
        int *waspEnd = nullptr;                          // vW the read after this range would inherit if it were a chimera in the BAM
    };
    // what the batch brings along besides its own alignments
    struct RangeIn {
        const MultOrder *order = nullptr;                // --outMultimapperOrder Random: the shuffles, drawn beforehand
        bool dry = false;                                // no alignment records, only the side outputs asked for
        const MergedBatch *merged = nullptr; const staramd_results *mergedRes = nullptr;   // --peOverlapNbasesMin: merged mates and their alignments
        const std::vector<int8_t> *waspType = nullptr;   // vW per read (--waspOutputMode SAMtag)
        bool *probeChimBam = nullptr;                    // internal: only decide whether the (one) read of the range is a chimera that goes to the BAM
    };
    std::string processRange(const ReadBatch &b, const staramd_results &r, uint32_t lo, uint32_t hi, const RangeOut &out, const RangeIn &in) const;
    std::string processRange(const ReadBatch &b, const staramd_results &r, uint32_t lo, uint32_t hi, const RangeOut &out) const;
    // nAlignT (with --quantMode TranscriptomeSAM): per read, the number of transcriptomic alignments + 1 where the read draws its primary one
    // right after its shuffles (ReadAlign_quantTranscriptome.cpp:69), 0 where it does not
    template <class Rng> void drawMultOrder(const ReadBatch &b, const staramd_results &r, Rng &&uniform01, MultOrder &o, const std::vector<uint32_t> *nAlignT = nullptr,
                                            const MergedBatch *merged = nullptr, const staramd_results *mergedRes = nullptr) const {
        o.offset.assign(b.n + 1, 0); o.partner.clear(); o.quantPick.assign(nAlignT ? b.n : 0, 0);
        for (uint32_t ir = 0; ir < b.n; ir++) {
            o.offset[ir] = o.partner.size();
            uint64_t nTr = 0, nbest = 0;
            multCounts(b, r, ir, merged, mergedRes, nTr, nbest);
            if (!(nTr > P.outFilterMultimapNmax || nTr < 2)) {
                for (int itr = (int)nbest - 1; itr >= 1; itr--) o.partner.push_back((uint32_t)int(uniform01() * itr + 0.5));
                for (int itr = (int)(nTr - nbest) - 1; itr >= 1; itr--) o.partner.push_back((uint32_t)int(uniform01() * itr + 0.5));
            }
            if (nAlignT && (*nAlignT)[ir]) o.quantPick[ir] = (uint32_t)(int)(uniform01() * ((*nAlignT)[ir] - 1));
        }
        o.offset[b.n] = o.partner.size();
    }
    // multMapSelect's counts for one read: alignments within the score range of the best, and how many of them tie with it (after mate merging, if any)
    void multCounts(const ReadBatch &b, const staramd_results &r, uint32_t ir, const MergedBatch *merged, const staramd_results *mergedRes, uint64_t &nTr, uint64_t &nbest) const;
    const GeneAnnotation *genes = nullptr;           // --quantMode GeneCounts
    const TranscriptAnnotation *transcripts = nullptr;   // --quantMode TranscriptomeSAM
    std::string quantBamHeader() const;              // samHeaders.cpp:8-20
    std::string samHeader() const;                   // samHeaders.cpp:27-106
    std::string bamHeader(bool sortedByCoordinate = false) const;                   // outBAMwriteHeader, BAMfunctions.cpp:83-98 (uncompressed bytes)
    int waspCarry = -1;                              // vW a chimeric first read of the batch inherits from the batch before (see processRange)
    bool samOff = false;                             // 1st pass of 2-pass mapping: no SAM text (twoPassRunPass1.cpp:18-22)
private:
    const RunParams &P;
    const GenomeIndex &gi;
};

// revComplementNucleotides, SequenceFuns.cpp:16-58
inline char rcNt(char c) {
    switch (c) {
        case 'A': return 'T'; case 'C': return 'G'; case 'G': return 'C'; case 'T': return 'A'; case 'N': return 'N';
        case 'R': return 'Y'; case 'Y': return 'R'; case 'K': return 'M'; case 'M': return 'K'; case 'S': return 'S'; case 'W': return 'W';
        case 'B': return 'V'; case 'D': return 'H'; case 'V': return 'B'; case 'H': return 'D';
        case 'a': return 't'; case 'c': return 'g'; case 'g': return 'c'; case 't': return 'a'; case 'n': return 'n';
        case 'r': return 'y'; case 'y': return 'r'; case 'k': return 'm'; case 'm': return 'k'; case 's': return 's'; case 'w': return 'w';
        case 'b': return 'v'; case 'd': return 'h'; case 'v': return 'b'; case 'h': return 'd';
        default: return c;
    }
}

} // namespace staramd
