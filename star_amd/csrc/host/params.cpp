// params.cpp -- the alignReads flags that reach the hot path or its outputs.
// Defaults are those of source/parametersDefault; derived values follow Parameters.cpp
// (alignEndsType :966-983) and Genome_genomeLoad.cpp:382-410 (window geometry).
// Every other STAR flag is rejected loudly rather than silently ignored.
#include "host.h"
#include <fstream>
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
    dev.alignEndsProtrudeNbasesMax = 0; dev.alignEndsProtrudeConcordantPair = 0;
    dev.alignSoftClipAtReferenceEnds = 1; dev.alignInsertionFlushRight = 0;
    dev.outFilterIntronStrandsRemoveInconsistent = 1; dev.outFilterIntronMotifs = 0; dev.outSAMstrandFieldIntronMotif = 0;
    dev.chimSegmentMinPositive = 0; dev.outFilterBySJoutStage = 0;
    dev.scoreGap = 0; dev.scoreGapNoncan = -8; dev.scoreGapGCAG = -4; dev.scoreGapATAC = -8;
    dev.scoreDelOpen = -2; dev.scoreDelBase = -2; dev.scoreInsOpen = -2; dev.scoreInsBase = -2;
    dev.scoreStitchSJshift = 1; dev.sjdbScore = 2; dev.scoreGenomicLengthLog2scale = -0.25;
    dev.outFilterMultimapScoreRange = 1; dev.outFilterMismatchNoverLmax = 0.3; dev.outFilterMatchNmin = 0; dev.resultSelect = 1;
}

std::string RunParams::parse(int argc, char **argv) {
    // Parameters::inputParameters (Parameters.cpp:311-441): command line first (--name value ... or --name=value), then the files of
    // --parametersFiles in turn, then the command line again on top; one definition per name and source
    std::map<std::string, std::vector<std::string> > kv, kvCommandLine;
    std::string cur;
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        commandLine += (i > 1 ? " " : "") + a;
        if (a.size() > 2 && a[0] == '-' && a[1] == '-') {
            size_t eq = a.find('=');
            cur = a.substr(2, eq == std::string::npos ? std::string::npos : eq - 2);
            if (kvCommandLine.count(cur)) return "EXITING: FATAL INPUT ERROR: duplicate parameter \"" + cur + "\" in input \"Command-Line\"\nSOLUTION: keep only one definition of input parameters in each input source\n";
            kvCommandLine[cur];
            if (eq != std::string::npos) kvCommandLine[cur].push_back(a.substr(eq + 1));
        }
        else if (cur.empty()) return "EXITING: fatal input ERROR: unrecognized parameter name \"" + a + "\" in input \"Command-Line-Initial\"";
        else kvCommandLine[cur].push_back(a);
    }
    for (auto &e : kvCommandLine) if (e.second.empty()) return "EXITING: FATAL INPUT ERROR: empty value for parameter \"" + e.first + "\" in input \"Command-Line\"\nSOLUTION: use non-empty value for this parameter\n";
    std::map<std::string, std::string> source;
    if (kvCommandLine.count("parametersFiles") && kvCommandLine["parametersFiles"][0] != "-") {
        for (const std::string &path : kvCommandLine["parametersFiles"]) {
            std::ifstream in(path.c_str());
            if (!in.good()) return "EXITING because of fatal input ERROR: could not open user-defined parameters file " + path + "\n";
            std::map<std::string, std::vector<std::string> > kvFile;
            std::string line;
            while (std::getline(in, line)) {                // Parameters::scanOneLine (:1205-1264), values as inputOneValue <string> reads them (ParameterInfo.h:31-42)
                size_t q = line.find_first_not_of(" \t\r");
                if (q == std::string::npos) continue;
                size_t q1 = line.find_first_of(" \t\r", q);
                std::string name = line.substr(q, q1 == std::string::npos ? std::string::npos : q1 - q);
                if (name.compare(0, 2, "//") == 0 || name[0] == '#') continue;
                std::vector<std::string> vals;
                while (q1 != std::string::npos) {
                    q = line.find_first_not_of(" \t\r", q1);
                    if (q == std::string::npos) break;
                    if (line[q] == '"') { q1 = line.find('"', q + 1); vals.push_back(line.substr(q + 1, q1 == std::string::npos ? std::string::npos : q1 - q - 1)); if (q1 != std::string::npos) q1++; }
                    else { q1 = line.find_first_of(" \t\r", q); vals.push_back(line.substr(q, q1 == std::string::npos ? std::string::npos : q1 - q)); }
                }
                if (vals.empty()) return "EXITING: FATAL INPUT ERROR: empty value for parameter \"" + name + "\" in input \"" + path + "\"\nSOLUTION: use non-empty value for this parameter\n";
                if (name == "parametersFiles" || name == "outFileNamePrefix" || name == "outTmpDir" || name == "outTmpKeep" || name == "outStd")
                    return "EXITING: FATAL INPUT ERROR: parameter \"" + name + "\" cannot be defined at the input level \"" + path + "\"\nSOLUTION: define parameter \"" + name + "\" in \"Command-Line\"\n";
                if (kvFile.count(name)) return "EXITING: FATAL INPUT ERROR: duplicate parameter \"" + name + "\" in input \"" + path + "\"\nSOLUTION: keep only one definition of input parameters in each input source\n";
                kvFile[name] = vals;
            }
            for (auto &e : kvFile) { kv[e.first] = e.second; source[e.first] = path; }
        }
    }
    for (auto &e : kvCommandLine) { kv[e.first] = e.second; source[e.first] = "Command-Line"; }
    kv.erase("parametersFiles");
    commandLine = std::string(argc > 0 ? argv[0] : "star_amd") + " " + commandLine;
    std::string err;
    auto one = [&](const std::string &k, const std::vector<std::string> &v) -> const std::string & {
        static std::string empty;
        if (v.size() != 1) { err = "EXITING: fatal input ERROR: --" + k + " expects exactly one value"; return empty; }
        return v[0];
    };
    // numbers: the reference reads them with operator>> and carries on with 0 after a failed conversion; a value that is not a number is
    // reported here instead (negative values of unsigned parameters wrap around exactly as they do there)
    auto notNumber = [&](const std::string &k, const std::string &t) { if (err.empty()) err = "EXITING: fatal input ERROR: --" + k + " expects a number, not \"" + t + "\"\nSOLUTION: check the value of --" + k + "\n"; };
    auto U = [&](const std::string &k, const std::vector<std::string> &v) { const std::string &t = one(k, v); char *e = nullptr; uint64_t x = strtoull(t.c_str(), &e, 10); if (!err.empty()) return (uint64_t)0; if (t.empty() || *e) notNumber(k, t); return x; };
    auto I = [&](const std::string &k, const std::vector<std::string> &v) { const std::string &t = one(k, v); char *e = nullptr; int64_t x = strtoll(t.c_str(), &e, 10); if (!err.empty()) return (int64_t)0; if (t.empty() || *e) notNumber(k, t); return x; };
    auto D = [&](const std::string &k, const std::vector<std::string> &v) { const std::string &t = one(k, v); char *e = nullptr; double x = strtod(t.c_str(), &e); if (!err.empty()) return 0.0; if (t.empty() || *e) notNumber(k, t); return x; };
    auto I4 = [&](const std::string &k, const std::vector<std::string> &v, int32_t *out) {
        if (v.size() != 4) { err = "EXITING: fatal input ERROR: --" + k + " expects 4 values"; return; }
        for (int j = 0; j < 4; j++) { char *e = nullptr; out[j] = (int32_t)strtol(v[j].c_str(), &e, 10); if (v[j].empty() || *e) notNumber(k, v[j]); }
    };
    std::string alignEndsType = "Local";
    std::map<std::string, std::vector<std::string> > clipArgs;
    for (auto &e : kv) {
        const std::string &k = e.first; const std::vector<std::string> &v = e.second;
        if (k == "runMode") { const std::string &m = one(k, v); if (m == "inputAlignmentsFromBAM") runModeFromBAM = true; else if (m == "genomeGenerate") runModeGenerate = true; else if (m != "alignReads") err = "EXITING: --runMode " + m + " is not implemented: alignReads, genomeGenerate and inputAlignmentsFromBAM are"; }
        else if (k == "genomeFastaFiles") { if (!(v.size() == 1 && v[0] == "-")) genomeFastaFiles = v; }
        else if (k == "genomeSAindexNbases") genomeSAindexNbases = (uint32_t)U(k, v);
        else if (k == "genomeChrBinNbits") genomeChrBinNbits = (uint32_t)U(k, v);
        else if (k == "genomeSAsparseD") genomeSAsparseD = (uint32_t)U(k, v);
        else if (k == "limitGenomeGenerateRAM") { (void)U(k, v); }      // the sort runs in HBM: nothing to size on the host
        else if (k == "inputBAMfile") { if (one(k, v) != "-") inputBAMfile = one(k, v); }
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
        else if (k == "genomeLoad") {           // the index lives in HBM, host shared memory does not come into it: the two values that only say how long to keep it are accepted
            const std::string &m = one(k, v);
            if (m == "LoadAndKeep" || m == "LoadAndRemove") genomeLoadShared = true;
            else if (m != "NoSharedMemory") err = "EXITING: --genomeLoad " + m + ": there is no shared-memory copy of the index to load or remove (it lives in HBM); use NoSharedMemory, LoadAndKeep or LoadAndRemove";
        }
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
        else if (k == "outStd") { outStd = one(k, v); if (outStd != "Log" && outStd != "SAM" && outStd != "BAM_Unsorted" && outStd != "BAM_SortedByCoordinate" && outStd != "BAM_Quant") err = "EXITING because of FATAL PARAMETER error: outStd=" + outStd + " is not a valid value of the parameter\nSOLUTION: provide a valid value fot outStd: Log / SAM / BAM_Unsorted / BAM_SortedByCoordinate"; }
        else if (k == "readFilesPrefix") { if (one(k, v) != "-") readFilesPrefix = one(k, v); }
        else if (k == "readFilesManifest") { if (one(k, v) != "-") readFilesManifest = one(k, v); }
        else if (k == "outSAMheaderHD") { if (v[0] != "-") outSAMheaderHD = v; }
        else if (k == "outSAMheaderPG") { if (v[0] != "-") outSAMheaderPG = v; }
        else if (k == "outSAMheaderCommentFile") { if (one(k, v) != "-") outSAMheaderCommentFile = one(k, v); }
        else if (k == "outSJtype") { const std::string &m = v[0]; if (m == "None") outSJnone = true; else if (m != "Standard") err = "EXITING because of FATAL input ERROR: unrecognized option in --outSJtype   " + m + "\nSOLUTION: use one of the allowed options: --outSJtype   Standard    OR    None\n"; }
        else if (k == "outQSconversionAdd") outQSconversionAdd = (int)I(k, v);
        else if (k == "varVCFfile") { if (one(k, v) != "-") varVCFfile = one(k, v); }
        else if (k == "waspOutputMode") { const std::string &m = one(k, v); if (m == "SAMtag") { wasp = true; varHeteroOnly = true; } else if (m != "None") err = "EXITING because of FATAL INPUT ERROR: unknown/unimplemented --waspOutputMode option: " + m + "\nSOLUTION: re-run STAR with allowed --waspOutputMode options: None or SAMtag\n"; }
        else if (k == "readFilesType") {       // Parameters_readFilesInit.cpp:11-39,152-166
            if (v[0] == "Fastx") readFilesSAMmates = 0;
            else if (v[0] == "SAM") {
                if (v.size() == 2 && v[1] == "SE") readFilesSAMmates = 1; else if (v.size() == 2 && v[1] == "PE") readFilesSAMmates = 2;
                else err = "EXITING because of FATAL INPUT ERROR: --readFilesType SAM requires specifying SE or PE reads\nSOLUTION: specify --readFilesType SAM SE for single-end reads or --readFilesType SAM PE for paired-end reads\n";
            } else err = "EXITING because of FATAL INPUT ERROR: unknown/unimplemented value for --readFilesType: " + v[0] + "\nSOLUTION: specify one of the allowed values: Fastx or SAM\n";
        }
        else if (k == "readFilesSAMattrKeep") {
            samAttrKeepAll = false; samAttrKeepNone = false; samAttrKeep.clear();
            if (v[0] == "All") samAttrKeepAll = true; else if (v[0] == "None") samAttrKeepNone = true;
            else for (auto &t : v) { if (t.size() != 2) err = "EXITING because of FATAL PARAMETER ERROR: each SAM tags in --readFilesSAMtagsKeep should contain two letters\n                                  SOLUTION: specify only two-letter tags in --readFilesSAMtagsKeep."; samAttrKeep.push_back(t); }
        }
        else if (k == "outWigType") {          // Parameters.cpp:511-552
            if (v[0] == "None") wig.yes = false; else if (v[0] == "bedGraph") { wig.yes = true; wig.format = 0; } else if (v[0] == "wiggle") { wig.yes = true; wig.format = 1; }
            else err = "EXITING because of FATAL INPUT ERROR: unrecognized option in --outWigType=" + v[0] + "\nSOLUTION: use one of the allowed values of --outWigType : 'None' or 'bedGraph' \n";
            if (v.size() > 1) { if (v[1] == "read1_5p") wig.type = 1; else if (v[1] == "read2") wig.type = 2; else err = "EXITING because of FATAL INPUT ERROR: unrecognized second option in --outWigType=" + v[1] + "\nSOLUTION: use one of the allowed values of --outWigType : 'read1_5p' \n"; }
        }
        else if (k == "outWigStrand") { const std::string &m = v[0]; if (m == "Stranded") wig.strand = true; else if (m == "Unstranded") wig.strand = false; else err = "EXITING because of FATAL INPUT ERROR: unrecognized option in --outWigStrand=" + m + "\nSOLUTION: use one of the allowed values of --outWigStrand : 'Stranded' or 'Unstranded' \n"; }
        else if (k == "outWigNorm") { const std::string &m = v[0]; if (m == "None") wig.norm = 0; else if (m == "RPM") wig.norm = 1; else err = "EXITING because of fatal parameter ERROR: unrecognized option in --outWigNorm=" + m + "\nSOLUTION: use one of the allowed values of --outWigNorm : 'None' or 'RPM' \n"; }
        else if (k == "outWigReferencesPrefix") { if (one(k, v) != "-") wig.referencesPrefix = one(k, v); }
        else if (k == "seedNoneLociPerWindow" || k == "winReadCoverageRelativeMin" || k == "winReadCoverageBasesMin" || k == "limitGenomeGenerateRAM") { (void)D(k, v); }   // long-read build / index generation only (ReadAlign_stitchPieces.cpp:202-230)
        else if (k == "peOverlapNbasesMin") peOverlapNbasesMin = (uint32_t)U(k, v);
        else if (k == "peOverlapMMp") peOverlapMMp = D(k, v);
        else if (k == "outMultimapperOrder") { const std::string &m = one(k, v); if (m == "Random") outMultimapperRandom = true; else if (m != "Old_2.4") err = "EXITING because of FATAL INPUT ERROR: unknown/unimplemented value for --outMultimapperOrder: " + m + "\nSOLUTION: specify one of the allowed values: Old_2.4 or Random\n"; }
        else if (k == "outSAMorder") { const std::string &m = one(k, v); if (m == "PairedKeepInputOrder") outSAMorderKeep = true; else if (m != "Paired") err = "EXITING because of FATAL INPUT ERROR: unknown value for --outSAMorder: " + m; }   // batches are always written in input order here
        else if (k == "readMatesLengthsIn") { const std::string &m = one(k, v); if (m != "NotEqual" && m != "Equal") err = "EXITING: unknown value for --readMatesLengthsIn: " + m; }
        else if (k == "readQualityScoreBase") { (void)I(k, v); }       // only STARsolo's statistics look at it (SoloFeature_statsOutput.cpp:18)
        else if (k == "runDirPerm") { const std::string &m = one(k, v); if (m == "All_RWX") runDirPermAll = true; else if (m != "User_RWX") err = "EXITING because of FATAL INPUT ERROR: unrecognized option in --runDirPerm=" + m + "\nSOLUTION: use one of the allowed values of --runDirPerm : 'User_RWX' or 'All_RWX' \n"; }
        // limits and knobs of the reference's own buffers, temporary files and sorting threads: nothing here is sized by them
        else if (k == "limitBAMsortRAM" || k == "limitIObufferSize" || k == "limitOutSAMoneReadBytes" || k == "limitOutSJcollapsed" || k == "limitOutSJoneRead" || k == "limitNreadsSoft"
                 || k == "outBAMsortingThreadN" || k == "outBAMsortingBinsN" || k == "outTmpDir" || k == "outTmpKeep" || k == "sysShell") { if (v.empty()) err = "EXITING: --" + k + " needs a value"; }
        else if (k == "outSAMmode") { const std::string &s = one(k, v); if (s == "NoQS") outSAMmodeNoQS = true; else if (s != "Full") err = "EXITING: unsupported --outSAMmode " + s; }
        else if (k == "outSAMunmapped") {       // Parameters.cpp:1062-1082
            if (v.size() == 1 && v[0] == "None") {}
            else if (v.size() == 1 && v[0] == "Within") outSAMunmappedWithin = true;
            else if (v.size() >= 2 && v[0] == "Within" && v[1] == "KeepPairs") { outSAMunmappedWithin = true; outSAMunmappedKeepPairs = true; }
            else { err = "EXITING because of fatal PARAMETERS error: unrecognized option for --outSAMunmapped="; for (auto &t : v) err += " " + t; err += "\nSOLUTION: use allowed options: None OR Within OR Within KeepPairs"; }
        }
        else if (k == "outSAMattributes") {
            if (v.size() == 1 && v[0] == "Standard") outSAMattrOrder = {"NH", "HI", "AS", "nM"};
            else if (v.size() == 1 && v[0] == "None") outSAMattrOrder.clear();
            else if (v.size() >= 1 && v[0] == "All") { outSAMattrOrder = {"NH", "HI", "AS", "nM", "NM", "MD", "jM", "jI", "MC", "ch"}; }   // + ch (Parameters_samAttributes.cpp:51-52)
            else { outSAMattrOrder.clear(); for (auto &t : v) { if (t == "NH" || t == "HI" || t == "AS" || t == "nM" || t == "jM" || t == "jI" || t == "XS" || t == "NM" || t == "MD" || t == "MC" || t == "RG" || t == "ch" || t == "vA" || t == "vG" || t == "vW" || t == "rB" || t == "cN") outSAMattrOrder.push_back(t); else err = "EXITING: unsupported SAM attribute " + t; } }
            if (outSAMattrOrder.size() > 26) err = "EXITING because of fatal PARAMETERS error: --outSAMattributes lists more than 26 attributes";      // (+ up to 4 added below; the formatter holds 32)
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
        else if (k == "chimMultimapNmax") chim.multimapNmax = U(k, v);
        else if (k == "chimMultimapScoreRange") chim.multimapScoreRange = U(k, v);
        else if (k == "chimNonchimScoreDropMin") chim.nonchimScoreDropMin = U(k, v);
        else if (k == "chimOutType") {          // ParametersChimeric_initialize.cpp:20-37
            chim.outJunctions = false;
            for (auto &t : v) {
                if (t == "Junctions") chim.outJunctions = true; else if (t == "WithinBAM") chim.outBam = true; else if (t == "HardClip") chim.bamHardClip = true; else if (t == "SoftClip") chim.bamHardClip = false;
                else if (t == "SeparateSAMold") chim.outSamOld = true;
                else err = "EXITING because of FATAL INPUT ERROR: unknown/unimplemented value for --chimOutType: " + t + "\nSOLUTION: re-run STAR with --chimOutType Junctions , SeparateSAMold  , WithinBAM , HardClip \n";
            }
        }
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
        else if (k == "alignEndsProtrude") { if (v.size() != 2) err = "EXITING: --alignEndsProtrude expects 2 values"; else {   // Parameters.cpp:1084-1097: the pair type only counts when protrusion is allowed at all
            dev.alignEndsProtrudeNbasesMax = (int32_t)strtol(v[0].c_str(), nullptr, 10); dev.alignEndsProtrudeConcordantPair = 0;
            if (dev.alignEndsProtrudeNbasesMax > 0) { if (v[1] == "ConcordantPair") dev.alignEndsProtrudeConcordantPair = 1; else if (v[1] != "DiscordantPair") err = "EXITING because of fatal PARAMETERS error: unrecognized option in of --alignEndsProtrude=" + v[1] + "\nSOLUTION: use allowed option: ConcordantPair or DiscordantPair"; } } }
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
    bool addXSlater = false, vWquant = false;
    {   // Parameters_samAttributes.cpp:172-178,213-216: XS <=> --outSAMstrandField intronMotif
        bool hasXS = std::find(outSAMattrOrder.begin(), outSAMattrOrder.end(), "XS") != outSAMattrOrder.end();
        if (hasXS) dev.outSAMstrandFieldIntronMotif = 1;
        else if (dev.outSAMstrandFieldIntronMotif) addXSlater = true;      // appended after the RG that --outSAMattrRGline brings (:201-216)
    }
    // Parameters.cpp:779-826
    if (twopass1Set && !twopass) return "EXITING because of fatal PARAMETERS error: --twopass1readsN is defined, but --twoPassMode is not defined\nSOLUTION: to activate the 2-pass mode, use --twopassMode Basic";
    if (twopass && twopass1readsN == 0) return "EXITING because of fatal PARAMETERS error: --twopass1readsN = 0 in the 2-pass mode\nSOLUTION: for the 2-pass mode, specify --twopass1readsN > 0. Use a very large number or -1 to map all reads in the 1st pass.\n";
    if (sjdbInsertYes() && sjdbOverhangSet && sjdbOverhang == 0) return "EXITING because of fatal PARAMETERS error: pGe.sjdbOverhang <=0 while junctions are inserted on the fly with --sjdbFileChrStartEnd or/and --sjdbGTFfile\nSOLUTION: specify pGe.sjdbOverhang>0, ideally readmateLength-1";
    if (runModeGenerate) {                   // Genome_genomeGenerate.cpp:100-129
        if (genomeDir.empty()) return "EXITING because of fatal INPUT error: --runMode genomeGenerate needs --genomeDir";
        if (genomeFastaFiles.empty()) return "EXITING because of fatal INPUT error: --runMode genomeGenerate needs --genomeFastaFiles";
        if (genomeSAindexNbases < 1 || genomeSAindexNbases > 16) return "EXITING because of fatal PARAMETERS error: --genomeSAindexNbases must be in 1..16";
        if (genomeChrBinNbits < 1 || genomeChrBinNbits > 40) return "EXITING because of fatal PARAMETERS error: bad --genomeChrBinNbits";
        if (!sjdbOverhangSet) sjdbOverhang = 100;
        if (sjdbOverhang == 0 && sjdbInsertPass1())
            return "EXITING because of FATAL INPUT PARAMETER ERROR: for generating genome with annotations (--sjdbFileChrStartEnd or --sjdbGTFfile options)\nyou need to specify >0 --sjdbOverhang\nSOLUTION: re-run genome generation specifying non-zero --sjdbOverhang, which ideally should be equal to OneMateLength-1, or could be chosen generically as ~100\n";
        if (!sjdbInsertPass1() && sjdbOverhangSet && sjdbOverhang > 0)
            return "EXITING because of FATAL INPUT PARAMETER ERROR: when generating genome without annotations (--sjdbFileChrStartEnd or --sjdbGTFfile options)\ndo not specify >0 --sjdbOverhang\nSOLUTION: re-run genome generation without --sjdbOverhang option\n";
        return "";
    }
    if (runModeFromBAM) {                    // Parameters.cpp:585-605
        if (!wig.yes) return "EXITING because of fatal INPUT error: at the moment --runMode inputFromBAM only works with --outWigType bedGraph OR --bamRemoveDuplicatesType Identical\nSOLUTION: re-run STAR with --outWigType bedGraph (duplicate removal is not implemented here)\n";
        if (inputBAMfile.empty()) return "EXITING because of fatal INPUT error: --runMode inputAlignmentsFromBAM needs --inputBAMfile";
        return "";
    }
    if (!readFilesManifest.empty()) {        // Parameters_readFilesInit.cpp:100-139: Read1 <tab> Read2 (or -) <tab> read group line
        std::ifstream rfM(readFilesManifest.c_str());
        if (!rfM.good()) return "EXITING because of fatal INPUT error: could not open input file " + readFilesManifest + "\nSOLUTION: check the path and permissions for readFilesManifest = " + readFilesManifest + "\n";
        std::vector<std::string> names[2]; std::string line;
        outSAMattrRG.clear(); outSAMattrRGlineSplit.clear();
        while (std::getline(rfM, line)) {
            if (line.find_first_not_of(" \t") >= line.size()) continue;
            size_t itab1 = 0, itab2 = 0;
            for (int im = 0; im < 2; im++) {
                itab2 = line.find('\t', itab1);
                if (itab2 >= line.size()) return "EXITING because of FATAL INPUT FILE error: readFileManifest file " + readFilesManifest + " has to contain at least 3 tab separated columns\nSOLUTION: fix the formatting of the readFileManifest file: Read1 <tab> Read2 <tab> ReadGroup. For single-end reads, use - in the 2nd column.\n";
                names[im].push_back(line.substr(itab1, itab2 - itab1));
                itab1 = itab2 + 1;
            }
            std::string rg = line.substr(itab2 + 1);
            if (rg.substr(0, 3) != "ID:") rg.insert(0, "ID:");
            outSAMattrRGlineSplit.push_back(rg);
            size_t t = rg.find('\t');
            outSAMattrRG.push_back(rg.substr(3, t == std::string::npos ? std::string::npos : t - 3));
        }
        if (names[0].empty()) return "EXITING because of FATAL INPUT FILE error: readFileManifest file " + readFilesManifest + " has no lines";
        const int nEnds = (!names[1][0].empty() && names[1][0].back() == '-') ? 1 : 2;      // (an empty 2nd column is a missing file name, reported when it is opened)
        readFilesIn.assign(nEnds, "");
        for (int im = 0; im < nEnds; im++) for (size_t i = 0; i < names[im].size(); i++) readFilesIn[im] += (i ? "," : "") + names[im][i];
    }
    if (readFilesIn.empty() || readFilesIn.size() > 2) return "EXITING: --readFilesIn expects 1 or 2 FASTQ files";
    for (std::string &list : readFilesIn) {                // a trailing comma is dropped, the prefix goes in front of every name (:46-61)
        if (!list.empty() && list.back() == ',') list.pop_back();
        if (!readFilesPrefix.empty()) { std::string o = readFilesPrefix; for (char c : list) { o.push_back(c); if (c == ',') o += readFilesPrefix; } list = o; }
    }
    if (readFilesIn.size() == 2 && std::count(readFilesIn[0].begin(), readFilesIn[0].end(), ',') != std::count(readFilesIn[1].begin(), readFilesIn[1].end(), ','))
        return "EXITING: because of fatal INPUT ERROR: number of input files for mate2=" + std::to_string(std::count(readFilesIn[1].begin(), readFilesIn[1].end(), ',') + 1) + " is not equal to that for mate0="
               + std::to_string(std::count(readFilesIn[0].begin(), readFilesIn[0].end(), ',') + 1) + "\nMake sure that the number of files in --readFilesIn is the same for both mates\n";
    // Parameters.cpp:710-721 (nothing here depends on the option, but the combinations the reference refuses are refused)
    if (outFilterBySJout && outSAMorderKeep) return "EXITING: fatal input ERROR: --outFilterType=BySJout is not presently compatible with --outSAMorder=PairedKeepInputOrder\nSOLUTION: re-run STAR without setting one of those parameters.\n";
    if (outSAMorderKeep && (outBAMunsorted || outBAMcoord || outSAMnone)) return "EXITING: fatal input ERROR: --outSAMorder=PairedKeepInputOrder is presently only compatible with SAM output, i.e. default --outSMAtype SAM\nSOLUTION: re-run STAR without --outSAMorder=PairedKeepInputOrder, or with --outSAMorder=PairedKeepInputOrder --outSMAtype SAM .\n";
    if (outSJnone && outFilterBySJout) return "EXITING because of FATAL input ERROR: --outFilterType BySJout requires --outSJtype Standard\nSOLUTION: --outFilterType Normal    OR   --outFilterType BySJout --outSJtype Standard\n";
    if (outSJnone && twopass) return "EXITING because of FATAL input ERROR: --twopassMode Basic needs the junctions of the 1st pass, i.e. --outSJtype Standard\n";
    if (twopass && genomeLoadShared) return "EXITING because of fatal PARAMETERS error: 2-pass method is not compatible with genomeLoad shared memory options\nSOLUTION: re-run STAR with --genomeLoad NoSharedMemory ; this is the only option compatible with --twopassMode Basic .\n";
    {   // read groups: one for all input files or one per file; the RG attribute comes with them (Parameters_readFilesInit.cpp:84-93, Parameters_samAttributes.cpp:201-206)
        size_t nFiles = readFilesIn.empty() ? 0 : (size_t)std::count(readFilesIn[0].begin(), readFilesIn[0].end(), ',') + 1;
        if (outSAMattrRG.size() > 1 && outSAMattrRG.size() != nFiles)
            return "EXITING: because of fatal INPUT ERROR: number of input read files: " + std::to_string(nFiles) + " does not agree with number of read group RG entries: " + std::to_string(outSAMattrRG.size()) + "\nMake sure that the number of RG lines in --outSAMattrRGline is equal to either 1, or the number of input read files in --readFilesIn\n";
        if (outSAMattrRG.size() == 1) for (size_t i = 1; i < nFiles; i++) outSAMattrRG.push_back(outSAMattrRG[0]);
        bool hasRG = std::find(outSAMattrOrder.begin(), outSAMattrOrder.end(), "RG") != outSAMattrOrder.end();
        if (!outSAMattrRG.empty() && !hasRG && readFilesManifest.empty()) outSAMattrOrder.push_back("RG");   // only --outSAMattrRGline adds the attribute by itself (Parameters_samAttributes.cpp:201)
        if (addXSlater) outSAMattrOrder.push_back("XS");
        {   // Parameters.cpp:876-890, Parameters_samAttributes.cpp:187-206
            auto has = [&](const char *a) { return std::find(outSAMattrOrder.begin(), outSAMattrOrder.end(), a) != outSAMattrOrder.end(); };
            if (wasp && varVCFfile.empty()) return "EXITING because of FATAL INPUT ERROR: --waspOutputMode option requires VCF file: SAMtag\nSOLUTION: re-run STAR with --waspOutputMode ... and --varVCFfile /path/to/file.vcf\n";
            if (wasp && !outBAMunsorted && !outBAMcoord) return "EXITING because of FATAL INPUT ERROR: --waspOutputMode requires output to BAM file\nSOLUTION: re-run STAR with --waspOutputMode ... and --outSAMtype BAM ... \n";
            if (varVCFfile.empty() && (has("vA") || has("vG"))) return "EXITING because of fatal PARAMETER error: --outSAMattributes contains vA and/or vG tag(s), but --varVCFfile is not set\nSOLUTION: re-run STAR with a --varVCFfile option, or without vA/vG tags in --outSAMattributes\n";
            if (!wasp && has("vW")) return "EXITING because of fatal PARAMETER error: --outSAMattributes contains vW tag, but --waspOutputMode is not set\nSOLUTION: re-run STAR with a --waspOutputMode option, or without vW tags in --outSAMattributes\n";
            if (wasp && !has("vW")) { outSAMattrOrder.push_back("vW"); vWquant = true; }      // only the vW that is added here goes into the transcriptome BAM as well (:201-206 vs :90-92)
            if (has("cN") && !outBAMunsorted && !outBAMcoord) return "EXITING: --outSAMattributes cN is only written to BAM output";
            for (const char *a : {"rB", "vG", "vA", "vW"})        // samAttrRequiresBAM (:236-240, 250-260), in this order
                if (has(a) && !outBAMunsorted && !outBAMcoord) return std::string("EXITING because of fatal PARAMETER error: --outSAMattributes contains ") + a + " tag, which requires BAM output.\nSOLUTION: re-run STAR with --outSAMtype BAM Unsorted (and/or) SortedByCoordinate option, or without " + a + " tag in --outSAMattributes\n";
            if (wasp && peOverlapNbasesMin > 0) return "EXITING: --waspOutputMode together with --peOverlapNbasesMin is not implemented";
        }
        if (outSAMattrRG.empty() && hasRG) return "EXITING because of fatal PARAMETER error: --outSAMattributes contains RG tag, but --outSAMattrRGline is not set\nSOLUTION: re-run STAR with a valid read group parameter --outSAMattrRGline.\n";
    }
    if (peOverlapNbasesMin > 0 && (readFilesIn.size() == 2 || readFilesSAMmates == 2)) dev.resultSelect = 0;          // every alignment of the merged mates is cut back into a pair and re-scored (ReadAlign_peOverlapMergeMap.cpp:279-296)
    dev.chimSegmentMin = (uint32_t)std::min<uint64_t>(chim.segmentMin, 0xFFFFFFFFull); dev.chimSegmentReadGapMax = (uint32_t)std::min<uint64_t>(chim.segmentReadGapMax, 0xFFFFFFFFull);
    if (chim.segmentMin > 0) { dev.chimSegmentMinPositive = 1; dev.resultSelect = 0; }      // every transcript of every window is needed (stitchWindowAligns.cpp:247)
    // ch marks chimeric alignments (never produced here) but the reference insists on BAM output for it (Parameters_samAttributes.cpp)
    attrHasCh = std::find(outSAMattrOrder.begin(), outSAMattrOrder.end(), "ch") != outSAMattrOrder.end();
    if (chim.segmentMin > 0 && chim.outBam) {               // ParametersChimeric_initialize.cpp:76-101
        if (!outBAMunsorted && !outBAMcoord) return "EXITING because of fatal PARAMETERS error: --chimOutType WithinBAM requires BAM output\nSOLUTION: re-run with --outSAMtype BAM Unsorted/SortedByCoordinate\n";
        if (std::find(outSAMattrOrder.begin(), outSAMattrOrder.end(), "NM") == outSAMattrOrder.end()) outSAMattrOrder.push_back("NM");
    }
    if (chim.segmentMin == 0) { chim.outBam = false; chim.outJunctions = false; chim.outSamOld = false; }
    if (chim.multimapNmax > 0 && chim.outSamOld) return "EXITING because of fatal PARAMETERS error: --chimMultimapNmax > 0 (new chimeric detection) presently only works with --chimOutType Junctions/WithinBAM\nSOLUTION: re-run with --chimOutType Junctions/WithinBAM\n";
    if (wig.yes && !outBAMcoord) return "EXITING because of fatal PARAMETER error: generating signal with --outWigType requires sorted BAM\nSOLUTION: re-run STAR with with --outSAMtype BAM SortedByCoordinate, or, id you also need unsroted BAM, with --outSAMtype BAM SortedByCoordinate Unsorted\n";
    if (peOverlapNbasesMin > 0 && chim.segmentMin > 0) {
        if (chim.multimapNmax == 0 && (chim.outJunctions || chim.outSamOld)) return "EXITING because of fatal PARAMETERS error: --chimMultimapNmax 0 (default old chimeric detection) and --peOverlapNbasesMin > 0 (merging ovelrapping mates) presently only works with --chimOutType WithinBAM\nSOLUTION: re-run with --chimOutType WithinBAM\n";
    }
    if (attrHasCh && !outBAMunsorted && !outBAMcoord) return "EXITING because of fatal PARAMETER error: --outSAMattributes contains ch tag, which requires BAM output.\nSOLUTION: re-run STAR with --outSAMtype BAM Unsorted (and/or) SortedByCoordinate option, or without ch tag in --outSAMattributes\n";
    outSAMattrOrderQuant = {"NH", "HI"};
    for (const std::string &a : outSAMattrOrder) if (a == "RG" || a == "MC" || a == "rB" || a == "cN" || (a == "vW" && vWquant)) outSAMattrOrderQuant.push_back(a);
    attrNMorMD = std::find(outSAMattrOrder.begin(), outSAMattrOrder.end(), "NM") != outSAMattrOrder.end() || std::find(outSAMattrOrder.begin(), outSAMattrOrder.end(), "MD") != outSAMattrOrder.end();
    if (genomeDir.empty()) return "EXITING: --genomeDir is required";
    dev.readNmates = (uint32_t)readFilesIn.size();
    if (readFilesSAMmates > 0) {
        if (readFilesIn.size() != 1) return "EXITING: --readFilesType SAM reads all mates from one input (one file or list in --readFilesIn)";
        if (readFilesIn[0].find(',') != std::string::npos) return "EXITING: --readFilesType SAM with several input files is not implemented; concatenate them in --readFilesCommand";
        dev.readNmates = (uint32_t)readFilesSAMmates;
    }
    if (dev.readNmates != 2) outSAMunmappedKeepPairs = false;     // Parameters.cpp:1074
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
