// params.cpp -- the alignReads flags that reach the hot path or its outputs.
// Defaults are those of source/parametersDefault; derived values follow Parameters.cpp
// (alignEndsType :966-983) and Genome_genomeLoad.cpp:382-410 (window geometry).
// Every other STAR flag is rejected loudly rather than silently ignored.
#include "host.h"
#include <cstring>
#include <cmath>
#include <cstdlib>
#include <algorithm>
#include <map>
#include <functional>

namespace staramd {

RunParams::RunParams() {
    memset(&dev, 0, sizeof(dev));
    dev.readNmates = 1;
    dev.seedSearchStartLmax = 50; dev.seedSearchStartLmaxOverLread = 1.0; dev.seedSearchLmax = 0;
    dev.seedMultimapNmax = 10000; dev.seedPerReadNmax = 1000; dev.seedPerWindowNmax = 50;
    dev.seedSplitMin = 12; dev.seedMapMin = 5; dev.maxNsplit = 10;
    dev.winAnchorMultimapNmax = 50; dev.winBinNbits = 16; dev.winAnchorDistNbins = 9; dev.winFlankNbins = 4;
    dev.alignWindowsPerReadNmax = 10000; dev.alignTranscriptsPerWindowNmax = 100; dev.alignTranscriptsPerReadNmax = 10000;
    dev.alignIntronMin = 21; dev.alignIntronMax = 0; dev.alignMatesGapMax = 0;
    dev.alignSJoverhangMin = 5; dev.alignSJDBoverhangMin = 3;
    dev.alignSJstitchMismatchNmax[0] = 0; dev.alignSJstitchMismatchNmax[1] = -1; dev.alignSJstitchMismatchNmax[2] = 0; dev.alignSJstitchMismatchNmax[3] = 0;
    dev.alignSplicedMateMapLmin = 0; dev.alignSplicedMateMapLminOverLmate = 0.66;
    dev.alignEndsProtrudeNbasesMax = 0; dev.alignEndsProtrudeConcordantPair = 1;
    dev.alignSoftClipAtReferenceEnds = 1; dev.alignInsertionFlushRight = 0;
    dev.outFilterIntronStrandsRemoveInconsistent = 1; dev.outFilterIntronMotifs = 0; dev.outSAMstrandFieldIntronMotif = 0;
    dev.chimSegmentMinPositive = 0; dev.outFilterBySJoutStage = 0;
    dev.scoreGap = 0; dev.scoreGapNoncan = -8; dev.scoreGapGCAG = -4; dev.scoreGapATAC = -8;
    dev.scoreDelOpen = -2; dev.scoreDelBase = -2; dev.scoreInsOpen = -2; dev.scoreInsBase = -2;
    dev.scoreStitchSJshift = 1; dev.sjdbScore = 2; dev.scoreGenomicLengthLog2scale = -0.25;
    dev.outFilterMultimapScoreRange = 1; dev.outFilterMismatchNoverLmax = 0.3; dev.outFilterMatchNmin = 0; dev.resultSelect = 1;
}

std::string RunParams::parse(int argc, char **argv) {
    std::map<std::string, std::vector<std::string> > kv;
    std::string cur;
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        commandLine += (i > 1 ? " " : "") + a;
        if (a.size() > 2 && a[0] == '-' && a[1] == '-') {
            cur = a.substr(2);
            if (kv.count(cur)) return "EXITING: FATAL INPUT ERROR: duplicate parameter \"" + cur + "\" in input \"Command-Line\"\nSOLUTION: keep only one definition of input parameters in each input source\n";
            kv[cur];
        }
        else if (cur.empty()) return "EXITING: fatal input ERROR: unrecognized parameter name \"" + a + "\" in input \"Command-Line-Initial\"";
        else kv[cur].push_back(a);
    }
    commandLine = std::string(argc > 0 ? argv[0] : "star_amd") + " " + commandLine;
    std::string err;
    auto one = [&](const std::string &k, const std::vector<std::string> &v) -> const std::string & {
        static std::string empty;
        if (v.size() != 1) { err = "EXITING: fatal input ERROR: --" + k + " expects exactly one value"; return empty; }
        return v[0];
    };
    auto U = [&](const std::string &k, const std::vector<std::string> &v) { return (uint64_t)strtoull(one(k, v).c_str(), nullptr, 10); };
    auto I = [&](const std::string &k, const std::vector<std::string> &v) { return (int64_t)strtoll(one(k, v).c_str(), nullptr, 10); };
    auto D = [&](const std::string &k, const std::vector<std::string> &v) { return strtod(one(k, v).c_str(), nullptr); };
    auto I4 = [&](const std::string &k, const std::vector<std::string> &v, int32_t *out) {
        if (v.size() != 4) { err = "EXITING: fatal input ERROR: --" + k + " expects 4 values"; return; }
        for (int j = 0; j < 4; j++) out[j] = (int32_t)strtol(v[j].c_str(), nullptr, 10);
    };
    std::string alignEndsType = "Local";
    std::map<std::string, std::vector<std::string> > clipArgs;
    for (auto &e : kv) {
        const std::string &k = e.first; const std::vector<std::string> &v = e.second;
        if (k == "runMode") { if (one(k, v) != "alignReads") err = "EXITING: only --runMode alignReads is implemented by the MI355X engine (index generation: use reference STAR)"; }
        else if (k == "genomeDir") genomeDir = one(k, v);
        else if (k == "readFilesIn") readFilesIn = v;
        else if (k == "outFileNamePrefix") outFileNamePrefix = one(k, v);
        else if (k == "readFilesCommand") { readFilesCommand.clear(); if (!(v.size() == 1 && v[0] == "-")) for (auto &t : v) readFilesCommand += (readFilesCommand.empty() ? "" : " ") + t; }
        else if (k == "runThreadN") runThreadN = (int)I(k, v);
        else if (k == "readMapNumber") readMapNumber = I(k, v);
        else if (k == "gpuBatchReads") gpuBatchReads = U(k, v);
        else if (k == "twopassMode") { const std::string &m = one(k, v); if (m == "Basic") twopass = true; else if (m != "None") err = "EXITING because of fatal PARAMETERS error: unrecognized value of --twopassMode=" + m + "\nSOLUTION: for the 2-pass mode, use allowed values --twopassMode: Basic"; }
        else if (k == "twopass1readsN") { twopass1readsN = I(k, v); twopass1Set = true; }
        else if (k == "sjdbFileChrStartEnd") { if (!(v.size() == 1 && v[0] == "-")) sjdbFileChrStartEnd = v; }
        else if (k == "sjdbOverhang") { sjdbOverhang = (uint32_t)U(k, v); sjdbOverhangSet = true; }
        else if (k == "sjdbInsertSave") { const std::string &m = one(k, v); if (m == "All") sjdbInsertSaveAll = true; else if (m != "Basic") err = "EXITING: unsupported --sjdbInsertSave " + m; }
        else if (k == "limitSjdbInsertNsj") limitSjdbInsertNsj = U(k, v);
        else if (k == "sjdbGTFfile") { if (one(k, v) != "-") sjdbGTFfile = one(k, v); }
        else if (k == "sjdbGTFchrPrefix") sjdbGTFchrPrefix = one(k, v);
        else if (k == "sjdbGTFfeatureExon") sjdbGTFfeatureExon = one(k, v);
        else if (k == "sjdbGTFtagExonParentTranscript") sjdbGTFtagExonParentTranscript = one(k, v);
        else if (k == "sjdbGTFtagExonParentGene") sjdbGTFtagExonParentGene = one(k, v);
        else if (k == "sjdbGTFtagExonParentGeneName") sjdbGTFtagExonParentGeneName = v;
        else if (k == "sjdbGTFtagExonParentGeneType") sjdbGTFtagExonParentGeneType = v;
        else if (k == "gpuDevice") gpuDevice = (int)I(k, v);
        else if (k == "genomeLoad") { if (one(k, v) != "NoSharedMemory") err = "EXITING: --genomeLoad: the index lives in HBM; only NoSharedMemory is accepted"; }
        else if (k == "outSAMtype") {                   // Parameters.cpp:611-683
            if (v.empty()) err = "EXITING because of fatal input ERROR: --outSAMtype needs a value";
            else if (v[0] == "SAM") { if (v.size() > 1) err = "EXITING because of fatal PARAMETER error: --outSAMtype SAM can cannot be combined with " + v[1] + " or any other options\nSOLUTION: re-run STAR with with '--outSAMtype SAM' only, or with --outSAMtype BAM Unsorted|SortedByCoordinate\n"; }
            else if (v[0] == "None") outSAMnone = true;
            else if (v[0] == "BAM") {
                if (v.size() < 2) err = "EXITING because of fatal PARAMETER error: missing BAM option\nSOLUTION: re-run STAR with one of the allowed values of --outSAMtype BAM Unsorted OR SortedByCoordinate OR both\n";
                for (size_t i = 1; i < v.size() && err.empty(); i++) {
                    if (v[i] == "Unsorted") outBAMunsorted = true;
                    else if (v[i] == "SortedByCoordinate") outBAMcoord = true;
                    else err = "EXITING because of fatal input ERROR: unknown value for the word " + std::to_string(i + 1) + " of outSAMtype: " + v[i] + "\nSOLUTION: re-run STAR with one of the allowed values of --outSAMtype BAM Unsorted or SortedByCoordinate or both\n";
                }
            } else err = "EXITING because of fatal input ERROR: unknown value for the first word of outSAMtype: " + v[0] + "\nSOLUTION: re-run STAR with one of the allowed values of outSAMtype: BAM or SAM \n";
        }
        else if (k == "outBAMcompression") outBAMcompression = (int)I(k, v);
        else if (k == "outStd") { if (one(k, v) != "Log") err = "EXITING: only --outStd Log is implemented"; }
        else if (k == "outSAMmode") { const std::string &s = one(k, v); if (s == "NoQS") outSAMmodeNoQS = true; else if (s != "Full") err = "EXITING: unsupported --outSAMmode " + s; }
        else if (k == "outSAMunmapped") { if (v.size() >= 1 && v[0] == "Within") { outSAMunmappedWithin = true; if (v.size() > 1) err = "EXITING: --outSAMunmapped Within KeepPairs is not implemented"; } else if (!(v.size() == 1 && v[0] == "None")) err = "EXITING: unsupported --outSAMunmapped"; }
        else if (k == "outSAMattributes") {
            if (v.size() == 1 && v[0] == "Standard") outSAMattrOrder = {"NH", "HI", "AS", "nM"};
            else if (v.size() == 1 && v[0] == "None") outSAMattrOrder.clear();
            else if (v.size() >= 1 && v[0] == "All") { outSAMattrOrder = {"NH", "HI", "AS", "nM", "NM", "MD", "jM", "jI", "MC"}; attrHasCh = true; }   // + ch (Parameters_samAttributes.cpp:51-52)
            else { outSAMattrOrder.clear(); for (auto &t : v) { if (t == "NH" || t == "HI" || t == "AS" || t == "nM" || t == "jM" || t == "jI" || t == "XS" || t == "NM" || t == "MD" || t == "MC" || t == "RG") outSAMattrOrder.push_back(t); else if (t == "ch") attrHasCh = true; else err = "EXITING: unsupported SAM attribute " + t; } }
        }
        else if (k == "outSAMstrandField") { const std::string &s = one(k, v); if (s == "intronMotif") { dev.outSAMstrandFieldIntronMotif = 1; } else if (s != "None") err = "EXITING: unsupported --outSAMstrandField " + s; }
        else if (k == "outSAMprimaryFlag") { const std::string &s = one(k, v); if (s == "AllBestScore") outSAMprimaryAllBest = true; else if (s != "OneBestScore") err = "EXITING: unsupported --outSAMprimaryFlag " + s; }
        else if (k == "outSAMmapqUnique") outSAMmapqUnique = (int)I(k, v);
        else if (k == "outSAMattrIHstart") outSAMattrIHstart = (int)I(k, v);
        else if (k == "outSAMflagOR") outSAMflagOR = (uint32_t)U(k, v);
        else if (k == "outSAMflagAND") outSAMflagAND = (uint32_t)U(k, v);
        else if (k == "readNameSeparator") readNameSeparator = one(k, v);
        else if (k == "outSAMattrRGline") {                // Parameters_readFilesInit.cpp:64-82
            if (!(v.size() == 1 && v[0] == "-")) {
                for (size_t ii = 0; ii < v.size(); ii++) {
                    if (ii == 0 || v[ii] == ",") {
                        if (ii > 0) ++ii;
                        if (ii >= v.size()) break;
                        outSAMattrRGlineSplit.push_back(v[ii]);
                        if (v[ii].substr(0, 3) != "ID:") { err = "EXITING because of FATAL INPUT ERROR: the first word of a line from --outSAMattrRGline=" + v[ii] + " does not start with ID:xxx read group identifier\nSOLUTION: re-run STAR with all lines in --outSAMattrRGline starting with ID:xxx\n"; break; }
                        outSAMattrRG.push_back(v[ii].substr(3));
                    } else outSAMattrRGlineSplit.back() += "\t" + v[ii];
                }
            }
        }
        else if (k == "outReadsUnmapped") { const std::string &m = one(k, v); if (m == "Fastx") outReadsUnmappedFastx = true; else if (m != "None") err = "EXITING because of fatal input ERROR: unknown value of --outReadsUnmapped " + m; }
        else if (k == "outSAMreadID") { const std::string &m = one(k, v); if (m == "Number") outSAMreadIDnumber = true; else if (m != "Standard") err = "EXITING because of fatal input ERROR: unknown value of --outSAMreadID " + m; }
        else if (k == "outSAMtlen") { outSAMtlen = (int)I(k, v); if (outSAMtlen != 1 && outSAMtlen != 2) err = "EXITING because of fatal PARAMETERS error: --outSAMtlen can only be 1 or 2"; }
        else if (k == "outSAMmultNmax") outSAMmultNmax = I(k, v);
        else if (k == "chimSegmentMin") chim.segmentMin = U(k, v);
        else if (k == "chimScoreMin") chim.scoreMin = (int)I(k, v);
        else if (k == "chimScoreDropMax") chim.scoreDropMax = (int)I(k, v);
        else if (k == "chimScoreSeparation") chim.scoreSeparation = (int)I(k, v);
        else if (k == "chimScoreJunctionNonGTAG") chim.scoreJunctionNonGTAG = (int)I(k, v);
        else if (k == "chimJunctionOverhangMin") chim.junctionOverhangMin = U(k, v);
        else if (k == "chimSegmentReadGapMax") chim.segmentReadGapMax = U(k, v);
        else if (k == "chimMainSegmentMultNmax") chim.mainSegmentMultNmax = U(k, v);
        else if (k == "chimOutJunctionFormat") chim.outJunctionFormat = (int)I(k, v);
        else if (k == "chimMultimapNmax") { if (U(k, v) != 0) err = "EXITING: --chimMultimapNmax > 0 (the multimapping chimeric detection) is not implemented; only the default 0"; }
        else if (k == "chimOutType") { for (auto &t : v) if (t != "Junctions") err = "EXITING: only --chimOutType Junctions is implemented (Chimeric.out.junction)"; }
        else if (k == "chimFilter") {
            chim.filterGenomicN = false;
            for (auto &t : v) { if (t == "banGenomicN") chim.filterGenomicN = true; else if (t != "None") err = "EXITING because of fatal PARAMETERS error: unrecognized value of --chimFilter=" + t + "\nSOLUTION: use allowed values: banGenomicN || None"; }
        }
        else if (k == "quantMode") {
            for (auto &t : v) { if (t == "GeneCounts") quantGeneCounts = true; else if (t == "TranscriptomeSAM") quantTrSAM = true; else if (t != "-") err = "EXITING because of fatal INPUT error: unrecognized option in --quantMode=" + t + "\nSOLUTION: use one of the allowed values of --quantMode : TranscriptomeSAM or GeneCounts or - .\n"; }
        }
        else if (k == "quantTranscriptomeBAMcompression") quantTrBAMcompression = (int)I(k, v);
        else if (k == "quantTranscriptomeSAMoutput") {     // Parameters.cpp:912-924
            const std::string &m = one(k, v);
            if (m == "BanSingleEnd_BanIndels_ExtendSoftclip") { quantTrIndel = false; quantTrSoftClip = false; }
            else if (m == "BanSingleEnd") { quantTrIndel = true; quantTrSoftClip = true; }
            else if (m == "BanSingleEnd_ExtendSoftclip") { quantTrIndel = true; quantTrSoftClip = false; }
            else err = "EXITING because of fatal INPUT error: unrecognized option in --quantTranscriptomeSAMoutput=" + m;
        }
        else if (k == "runRNGseed") runRNGseed = (int)I(k, v);
        else if (k.compare(0, 5, "clip5") == 0 || k.compare(0, 5, "clip3") == 0) clipArgs[k] = v;
        else if (k == "clipAdapterType") { if (one(k, v) != "Hamming") err = "EXITING because of fatal PARAMETER error: --clipAdapterType = " + one(k, v) + " is not implemented here (Hamming only)\n"; }
        else if (k == "outFilterType") { const std::string &m = one(k, v); if (m == "BySJout") outFilterBySJout = true; else if (m != "Normal") err = "EXITING because of FATAL input ERROR: unknown value of parameter outFilterType: " + m + "\nSOLUTION: specify one of the allowed values: Normal | BySJout\n"; }
        else if (k == "outFilterMultimapScoreRange") dev.outFilterMultimapScoreRange = (int32_t)I(k, v);
        else if (k == "outFilterMultimapNmax") outFilterMultimapNmax = (uint32_t)U(k, v);
        else if (k == "outFilterMismatchNmax") outFilterMismatchNmax = (uint32_t)U(k, v);
        else if (k == "outFilterMismatchNoverLmax") dev.outFilterMismatchNoverLmax = D(k, v);
        else if (k == "outFilterMismatchNoverReadLmax") outFilterMismatchNoverReadLmax = D(k, v);
        else if (k == "outFilterScoreMin") outFilterScoreMin = (int32_t)I(k, v);
        else if (k == "outFilterScoreMinOverLread") outFilterScoreMinOverLread = D(k, v);
        else if (k == "outFilterMatchNmin") { outFilterMatchNmin = (uint32_t)U(k, v); dev.outFilterMatchNmin = outFilterMatchNmin; }
        else if (k == "outFilterMatchNminOverLread") outFilterMatchNminOverLread = D(k, v);
        else if (k == "gpuResultSelect") { const std::string &s = one(k, v); if (s == "All") dev.resultSelect = 0; else if (s == "Selected") dev.resultSelect = 1; else err = "EXITING: --gpuResultSelect takes All or Selected"; }
        else if (k == "outFilterIntronMotifs") { const std::string &s = one(k, v); if (s == "None") dev.outFilterIntronMotifs = 0; else if (s == "RemoveNoncanonical") dev.outFilterIntronMotifs = 1; else if (s == "RemoveNoncanonicalUnannotated") dev.outFilterIntronMotifs = 2; else err = "EXITING because of FATAL INPUT error: unrecognized value of --outFilterIntronMotifs=" + s; }
        else if (k == "outFilterIntronStrands") { const std::string &s = one(k, v); if (s == "RemoveInconsistentStrands") dev.outFilterIntronStrandsRemoveInconsistent = 1; else if (s == "None") dev.outFilterIntronStrandsRemoveInconsistent = 0; else err = "EXITING: unsupported --outFilterIntronStrands " + s; }
        else if (k == "outSJtype") { if (one(k, v) != "Standard") err = "EXITING: only --outSJtype Standard is implemented"; }
        else if (k == "outSJfilterReads") { const std::string &s = one(k, v); if (s == "Unique") outSJfilterReadsUnique = true; else if (s != "All") err = "EXITING: unsupported --outSJfilterReads " + s; }
        else if (k == "outSJfilterOverhangMin") I4(k, v, outSJfilterOverhangMin);
        else if (k == "outSJfilterCountUniqueMin") I4(k, v, outSJfilterCountUniqueMin);
        else if (k == "outSJfilterCountTotalMin") I4(k, v, outSJfilterCountTotalMin);
        else if (k == "outSJfilterDistToOtherSJmin") I4(k, v, outSJfilterDistToOtherSJmin);
        else if (k == "outSJfilterIntronMaxVsReadN") { outSJfilterIntronMaxVsReadN.clear(); for (auto &t : v) outSJfilterIntronMaxVsReadN.push_back(strtoull(t.c_str(), nullptr, 10)); }
        else if (k == "scoreGap") dev.scoreGap = (int32_t)I(k, v);
        else if (k == "scoreGapNoncan") dev.scoreGapNoncan = (int32_t)I(k, v);
        else if (k == "scoreGapGCAG") dev.scoreGapGCAG = (int32_t)I(k, v);
        else if (k == "scoreGapATAC") dev.scoreGapATAC = (int32_t)I(k, v);
        else if (k == "scoreGenomicLengthLog2scale") dev.scoreGenomicLengthLog2scale = D(k, v);
        else if (k == "scoreDelOpen") dev.scoreDelOpen = (int32_t)I(k, v);
        else if (k == "scoreDelBase") dev.scoreDelBase = (int32_t)I(k, v);
        else if (k == "scoreInsOpen") dev.scoreInsOpen = (int32_t)I(k, v);
        else if (k == "scoreInsBase") dev.scoreInsBase = (int32_t)I(k, v);
        else if (k == "scoreStitchSJshift") dev.scoreStitchSJshift = (int32_t)I(k, v);
        else if (k == "sjdbScore") dev.sjdbScore = (int32_t)I(k, v);
        else if (k == "seedSearchStartLmax") dev.seedSearchStartLmax = (uint32_t)U(k, v);
        else if (k == "seedSearchStartLmaxOverLread") dev.seedSearchStartLmaxOverLread = D(k, v);
        else if (k == "seedSearchLmax") dev.seedSearchLmax = (uint32_t)U(k, v);
        else if (k == "seedMultimapNmax") dev.seedMultimapNmax = (uint32_t)U(k, v);
        else if (k == "seedPerReadNmax") dev.seedPerReadNmax = (uint32_t)U(k, v);
        else if (k == "seedPerWindowNmax") dev.seedPerWindowNmax = (uint32_t)U(k, v);
        else if (k == "seedSplitMin") dev.seedSplitMin = (uint32_t)U(k, v);
        else if (k == "seedMapMin") dev.seedMapMin = (uint32_t)U(k, v);
        else if (k == "alignIntronMin") dev.alignIntronMin = U(k, v);
        else if (k == "alignIntronMax") dev.alignIntronMax = U(k, v);
        else if (k == "alignMatesGapMax") dev.alignMatesGapMax = U(k, v);
        else if (k == "alignSJoverhangMin") dev.alignSJoverhangMin = (uint32_t)U(k, v);
        else if (k == "alignSJDBoverhangMin") dev.alignSJDBoverhangMin = (uint32_t)U(k, v);
        else if (k == "alignSJstitchMismatchNmax") I4(k, v, dev.alignSJstitchMismatchNmax);
        else if (k == "alignSplicedMateMapLmin") dev.alignSplicedMateMapLmin = (uint32_t)U(k, v);
        else if (k == "alignSplicedMateMapLminOverLmate") dev.alignSplicedMateMapLminOverLmate = D(k, v);
        else if (k == "alignWindowsPerReadNmax") dev.alignWindowsPerReadNmax = (uint32_t)U(k, v);
        else if (k == "alignTranscriptsPerWindowNmax") dev.alignTranscriptsPerWindowNmax = (uint32_t)U(k, v);
        else if (k == "alignTranscriptsPerReadNmax") dev.alignTranscriptsPerReadNmax = (uint32_t)U(k, v);
        else if (k == "alignEndsType") alignEndsType = one(k, v);
        else if (k == "alignEndsProtrude") { if (v.size() != 2) err = "EXITING: --alignEndsProtrude expects 2 values"; else { dev.alignEndsProtrudeNbasesMax = (int32_t)strtol(v[0].c_str(), nullptr, 10); dev.alignEndsProtrudeConcordantPair = v[1] == "ConcordantPair"; } }
        else if (k == "alignSoftClipAtReferenceEnds") { const std::string &s = one(k, v); dev.alignSoftClipAtReferenceEnds = (s == "Yes"); if (s != "Yes" && s != "No") err = "EXITING: unsupported --alignSoftClipAtReferenceEnds " + s; }
        else if (k == "alignInsertionFlush") { const std::string &s = one(k, v); dev.alignInsertionFlushRight = (s == "Right"); if (s != "None" && s != "Right") err = "EXITING: unsupported --alignInsertionFlush " + s; }
        else if (k == "winAnchorMultimapNmax") dev.winAnchorMultimapNmax = (uint32_t)U(k, v);
        else if (k == "winBinNbits") dev.winBinNbits = (uint32_t)U(k, v);
        else if (k == "winAnchorDistNbins") dev.winAnchorDistNbins = (uint32_t)U(k, v);
        else if (k == "winFlankNbins") dev.winFlankNbins = (uint32_t)U(k, v);
        else err = "EXITING: fatal input ERROR: parameter \"" + k + "\" is outside the scope of the MI355X alignReads engine (SURVEY.md section 2) -- refusing to ignore it";
        if (!err.empty()) return err;
    }
    // Parameters.cpp:966-983
    memset(dev.alignEndsTypeExt, 0, sizeof(dev.alignEndsTypeExt));
    if (alignEndsType == "EndToEnd") { dev.alignEndsTypeExt[0][0] = dev.alignEndsTypeExt[0][1] = dev.alignEndsTypeExt[1][0] = dev.alignEndsTypeExt[1][1] = 1; }
    else if (alignEndsType == "Extend5pOfRead1") dev.alignEndsTypeExt[0][0] = 1;
    else if (alignEndsType == "Extend5pOfReads12") { dev.alignEndsTypeExt[0][0] = 1; dev.alignEndsTypeExt[1][0] = 1; }
    else if (alignEndsType == "Extend3pOfRead1") dev.alignEndsTypeExt[0][1] = 1;
    else if (alignEndsType != "Local") return "EXITING because of FATAL INPUT ERROR: unknown/unimplemented value for --alignEndsType: " + alignEndsType;
    {   // Parameters_samAttributes.cpp:172-178,213-216: XS <=> --outSAMstrandField intronMotif
        bool hasXS = std::find(outSAMattrOrder.begin(), outSAMattrOrder.end(), "XS") != outSAMattrOrder.end();
        if (hasXS) dev.outSAMstrandFieldIntronMotif = 1;
        else if (dev.outSAMstrandFieldIntronMotif) outSAMattrOrder.push_back("XS");
    }
    // Parameters.cpp:779-826
    if (twopass1Set && !twopass) return "EXITING because of fatal PARAMETERS error: --twopass1readsN is defined, but --twoPassMode is not defined\nSOLUTION: to activate the 2-pass mode, use --twopassMode Basic";
    if (twopass && twopass1readsN == 0) return "EXITING because of fatal PARAMETERS error: --twopass1readsN = 0 in the 2-pass mode\nSOLUTION: for the 2-pass mode, specify --twopass1readsN > 0. Use a very large number or -1 to map all reads in the 1st pass.\n";
    if (sjdbInsertYes() && sjdbOverhangSet && sjdbOverhang == 0) return "EXITING because of fatal PARAMETERS error: pGe.sjdbOverhang <=0 while junctions are inserted on the fly with --sjdbFileChrStartEnd or/and --sjdbGTFfile\nSOLUTION: specify pGe.sjdbOverhang>0, ideally readmateLength-1";
    {   // read groups: one for all input files or one per file; the RG attribute comes with them (Parameters_readFilesInit.cpp:84-93, Parameters_samAttributes.cpp:201-206)
        size_t nFiles = readFilesIn.empty() ? 0 : (size_t)std::count(readFilesIn[0].begin(), readFilesIn[0].end(), ',') + 1;
        if (outSAMattrRG.size() > 1 && outSAMattrRG.size() != nFiles)
            return "EXITING: because of fatal INPUT ERROR: number of input read files: " + std::to_string(nFiles) + " does not agree with number of read group RG entries: " + std::to_string(outSAMattrRG.size()) + "\nMake sure that the number of RG lines in --outSAMattrRGline is equal to either 1, or the number of input read files in --readFilesIn\n";
        if (outSAMattrRG.size() == 1) for (size_t i = 1; i < nFiles; i++) outSAMattrRG.push_back(outSAMattrRG[0]);
        bool hasRG = std::find(outSAMattrOrder.begin(), outSAMattrOrder.end(), "RG") != outSAMattrOrder.end();
        if (!outSAMattrRG.empty() && !hasRG) outSAMattrOrder.push_back("RG");
        if (outSAMattrRG.empty() && hasRG) return "EXITING because of fatal PARAMETER error: --outSAMattributes contains RG tag, but --outSAMattrRGline is not set\nSOLUTION: re-run STAR with a valid read group parameter --outSAMattrRGline.\n";
    }
    if (chim.segmentMin > 0) { dev.chimSegmentMinPositive = 1; dev.resultSelect = 0; }      // every transcript of every window is needed (stitchWindowAligns.cpp:247)
    // ch marks chimeric alignments (never produced here) but the reference insists on BAM output for it (Parameters_samAttributes.cpp)
    if (attrHasCh && !outBAMunsorted && !outBAMcoord) return "EXITING because of fatal PARAMETER error: --outSAMattributes contains ch tag, which requires BAM output.\nSOLUTION: re-run STAR with --outSAMtype BAM Unsorted (and/or) SortedByCoordinate option, or without ch tag in --outSAMattributes\n";
    outSAMattrOrderQuant = {"NH", "HI"};
    for (const std::string &a : outSAMattrOrder) if (a == "RG" || a == "MC") outSAMattrOrderQuant.push_back(a);
    attrNMorMD = std::find(outSAMattrOrder.begin(), outSAMattrOrder.end(), "NM") != outSAMattrOrder.end() || std::find(outSAMattrOrder.begin(), outSAMattrOrder.end(), "MD") != outSAMattrOrder.end();
    if (genomeDir.empty()) return "EXITING: --genomeDir is required";
    if (readFilesIn.empty() || readFilesIn.size() > 2) return "EXITING: --readFilesIn expects 1 or 2 FASTQ files";
    dev.readNmates = (uint32_t)readFilesIn.size();
    {   // ParametersClip::initialize (ParametersClip_initialize.cpp:33-82): a lone 0 / "-" is repeated for all mates, anything else needs one value per mate
        const std::string nm = std::to_string(dev.readNmates);
        const char *p53[2] = {"5", "3"};
        for (int ip = 0; ip < 2; ip++) {
            auto get = [&](const char *name, const char *def) { auto it = clipArgs.find(std::string("clip") + p53[ip] + "p" + name); return it == clipArgs.end() ? std::vector<std::string>{def} : it->second; };
            std::vector<std::string> adSeq = get("AdapterSeq", "-"), adMMp = get("AdapterMMp", "0.1"), N = get("Nbases", "0"), NafterAd = get("AfterAdapterNbases", "0");
            if (ip == 0) for (auto &x : adSeq) if (x != "-")
                return std::string("EXITING because of fatal PARAMETER error: --clip5pAdapterSeq is not supported yet, except for --clipAdapterType CellRanger4.                            \nSOLUTION: Do not use --clip5pAdapter* options without --clipAdapterType CellRanger4.\n");
            if (adSeq[0] == "-") { adSeq.resize(dev.readNmates, "-"); adMMp.resize(dev.readNmates, "0"); }
            if (strtoul(N[0].c_str(), nullptr, 10) == 0) N.resize(dev.readNmates, "0");
            if (strtoul(NafterAd[0].c_str(), nullptr, 10) == 0) NafterAd.resize(dev.readNmates, "0");
            auto bad = [&](const char *name, const char *tail) { return std::string("EXITING because of fatal PARAMETER error: --clip") + p53[ip] + "p" + name + " has to contain " + nm + " values to match the number of mates.\nSOLUTION: specify " + nm + "values in --clip" + p53[ip] + "p" + name + tail; };
            if (adSeq.size() != dev.readNmates) return bad("AdapterSeq", " , for no clipping use -");
            if (adMMp.size() != dev.readNmates) return bad("AdapterMMp", "");
            if (NafterAd.size() != dev.readNmates) return bad("AfterAdapterNbases", " , for no clipping use 0");
            if (N.size() != dev.readNmates) return bad("Nbases", " , for no clipping use 0");
            for (uint32_t m = 0; m < dev.readNmates; m++) {   // ClipMate::initialize (ClipMate_initialize.cpp:5-31): no N and no adapter = no clipping at this end at all
                ClipEnd &c = clip[m][ip];
                c.N = (uint32_t)strtoul(N[m].c_str(), nullptr, 10); c.NafterAd = (uint32_t)strtoul(NafterAd[m].c_str(), nullptr, 10); c.adMMp = strtod(adMMp[m].c_str(), nullptr);
                c.adSeq = adSeq[m] == "-" ? "" : adSeq[m] == "polyA" ? std::string(650, 'A') : adSeq[m];
                c.active = c.N > 0 || !c.adSeq.empty();
                if (c.active) clipYes = true;
            }
        }
    }
    return "";
}

void RunParams::finalize(const GenomeIndex &gi) {
    // Genome_genomeLoad.cpp:382-410
    uint64_t nGenome = gi.view.nGenome;
    if (!(dev.alignIntronMax == 0 && dev.alignMatesGapMax == 0)) {
        uint64_t a = std::max<uint64_t>(std::max<uint64_t>(4ull, dev.alignIntronMax), dev.alignMatesGapMax == 0 ? 1000ull : dev.alignMatesGapMax) / 4;
        dev.winBinNbits = (uint32_t)std::floor(std::log2((double)a) + 0.5);
        dev.winBinNbits = std::max<uint32_t>(dev.winBinNbits, (uint32_t)std::floor(std::log2((double)(nGenome / 40000 + 1)) + 0.5));
    }
    if (dev.winBinNbits > gi.view.gChrBinNbits) dev.winBinNbits = gi.view.gChrBinNbits;
    if (!(dev.alignIntronMax == 0 && dev.alignMatesGapMax == 0)) {
        dev.winFlankNbins = (uint32_t)(std::max(dev.alignIntronMax, dev.alignMatesGapMax) / (1ull << dev.winBinNbits) + 1);
        dev.winAnchorDistNbins = 2 * dev.winFlankNbins;
    }
    dev.winBinChrNbits = gi.view.gChrBinNbits - dev.winBinNbits;
    dev.winBinN = nGenome / (1ull << dev.winBinNbits) + 1;
}

} // namespace staramd
