/* star_amd_index.h -- C ABI of the on-device index stages (MI355X, gfx950).
 *
 * Two things in the reference turn a genome into the arrays the hot path walks:
 *   (1) Genome::genomeGenerate: suffix array sort + SAindex        source/Genome_genomeGenerate.cpp:191-316,
 *                                                                    source/genomeSAindex.cpp:6-217
 *   (2) sjdbBuildIndex: junction insertion into SA / SAindex         source/sjdbBuildIndex.cpp:15-333
 *       (genomeGenerate with annotations, --sjdbGTFfile / --sjdbFileChrStartEnd at the mapping stage, and between the
 *        passes of --twopassMode Basic: source/sjdbInsertJunctions.cpp:12-102)
 * Both are rebuilt here as radix sorts, scans and streaming merges over arrays in HBM; the outputs are byte-identical
 * to the files the reference writes (`SA`, `SAindex`).  (1) is what lets bench.py run on a human-size index inside its
 * time budget (the reference needs tens of minutes for it); (2) is SURVEY.md section 8(f) row 2.
 *
 * Plain C: host pointers + sizes, caller-owned buffers, 0 / negative return codes (STARAMD_ERR_* of star_amd.h).
 */
#ifndef STAR_AMD_INDEX_H
#define STAR_AMD_INDEX_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct staramd_index_params {
    uint64_t nGenome;          /* bytes of G: chromosomes padded to 2^genomeChrBinNbits (genomeScanFastaFiles.cpp)        */
    uint32_t GstrandBit;       /* floor(log2(nGenome + limitSjdbInsertNsj*sjdbLength))+1, >=32  Genome_genomeGenerate.cpp:177 */
    uint32_t gSAindexNbases;   /* --genomeSAindexNbases                                                                   */
    uint32_t gSAsparseD;       /* --genomeSAsparseD: only 1 is built on the device                                        */
    uint32_t reserved;
} staramd_index_params;

typedef struct staramd_index_result {
    uint64_t nSA, nSAbyte;     /* entries / bytes of the packed suffix array ((GstrandBit+1) bits per entry)              */
    uint64_t nSAi, nSAibyte;   /* entries / bytes of the packed SAindex ((GstrandBit+3) bits per entry), without header   */
    uint64_t genomeSAindexStart[17];
    uint32_t doublingRounds;
    float    msText, msSort, msPack, msSAindex, msTotal;   /* HIP-event times on the build stream                          */
} staramd_index_result;

/* Suffix array + SAindex of the genome G (codes 0..3 ACGT, 4 N, 5 padding, exactly the `Genome` file).
 * SA / SAi: host buffers of saCapacity / saiCapacity bytes (nSAbyte / nSAibyte are reported; call with null buffers to
 * learn the sizes: nSA = 2 * number of bases < 4).  */
int staramd_index_build(int device, const uint8_t *G, const staramd_index_params *p,
                        uint8_t *SA, uint64_t saCapacity, uint8_t *SAi, uint64_t saiCapacity, staramd_index_result *res);

/* Junction insertion (sjdbBuildIndex, source/sjdbBuildIndex.cpp:15-333) on the device, host buffers in and out: the search of every new
 * junction suffix in the old suffix array, the sort of the new suffixes, the merge into the new packed array and the SAindex of the result.
 * What sjdbPrepare decides (which junctions, their sequences, motifs, shifts) is the caller's: it hands in
 *   Gsj       sjdbN blocks of sjdbLength codes (donor flank, acceptor flank, spacer 5): the forward half of the reference's Gsj
 *   isOld     per junction of the NEW table: 1 = already in the old index (binarySearch2 hit, :47-55): no new suffixes
 *   oldSJind  per junction of the OLD table: its number in the new table (:54)
 * SAout / SAiOut are sized by the caller: nSAnew = nSAold + (number of offsets in Gsj+revcomp(Gsj) that start with ACGT inside new junctions). */
typedef struct staramd_sjdb_args {
    const uint8_t *G; uint64_t nGenomeOld, nGenomeReal;      /* old genome text; nGenomeReal = chrStart[nChrReal] = start of the junction block */
    const uint8_t *SA; uint64_t nSAold, nSAbyteOld; uint32_t GstrandBit, gSAindexNbases;
    const uint8_t *Gsj; uint32_t sjdbN, sjdbLength;
    const uint8_t *isOld; const uint32_t *oldSJind; uint32_t oldSjdbN; uint32_t reserved; uint64_t sjNew;
    uint8_t *SAout; uint64_t saOutCapacity; uint8_t *SAiOut; uint64_t saiOutCapacity;
} staramd_sjdb_args;
typedef struct staramd_sjdb_result { uint64_t nInd, nSAnew, nSAbyteNew, nSAibyte; float msTotal; uint32_t reserved; } staramd_sjdb_result;
int staramd_sjdb_insert(int device, const staramd_sjdb_args *a, staramd_sjdb_result *res);

const char *staramd_index_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
