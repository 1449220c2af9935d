// k_gather.hip -- kernel 4: order the per-window transcript blocks by read and produce the C-ABI result arrays.
// (No counterpart in the reference: there the per-thread ReadAlign object owns trAll[][] directly.)
// Output order is deterministic -- reads in batch order, windows in window order, transcripts best-first --
// so result buffers can be compared byte for byte between runs and against the oracle.
#include "dev.h"

// exclusive scan of (nTr, nEx) over reads: one block, each thread scans a contiguous chunk
extern "C" __global__ void __launch_bounds__(1024) k_scan_offsets(DevBatch B, u32 *trBase, u32 *exBase, u32 *totals) {
    if (B.cursors[CUR_FLAGS] != 0) return;          // a pool overflowed in an earlier kernel: the host grows it and re-runs the batch
    __shared__ u32 sT[1024], sE[1024];
    u32 t = threadIdx.x, n = B.nReads;
    u32 chunk = (n + 1023) / 1024;
    u32 lo = min(n, t * chunk), hi = min(n, lo + chunk);
    u32 aT = 0, aE = 0;
    for (u32 i = lo; i < hi; i++) { aT += B.reads[i].nTr; aE += B.reads[i].nEx; }
    sT[t] = aT; sE[t] = aE;
    __syncthreads();
    for (u32 off = 1; off < 1024; off <<= 1) {
        u32 vT = t >= off ? sT[t - off] : 0, vE = t >= off ? sE[t - off] : 0;
        __syncthreads();
        sT[t] += vT; sE[t] += vE;
        __syncthreads();
    }
    u32 bT = sT[t] - aT, bE = sE[t] - aE;
    for (u32 i = lo; i < hi; i++) { trBase[i] = bT; exBase[i] = bE; bT += B.reads[i].nTr; bE += B.reads[i].nEx; }
    if (t == 1023) { totals[0] = sT[1023]; totals[1] = sE[1023]; }
}

extern "C" __global__ void __launch_bounds__(256) k_gather(DevBatch B, const u32 *trBase, const u32 *exBase,
                                                          staramd_read_result *outReads, staramd_transcript *outTr, u32 outTrCap,
                                                          staramd_exon *outEx, u32 outExCap) {
    if (B.cursors[CUR_FLAGS] != 0) return;          // a pool overflowed in an earlier kernel: the host grows it and re-runs the batch
    u32 ir = blockIdx.x * blockDim.x + threadIdx.x;
    if (ir >= B.nReads) return;
    const DRead rd = B.reads[ir];
    staramd_read_result rr;
    rr.status = rd.status; rr.nW = rd.nWt; rr.nTr = rd.nTr; rr.trOffset = trBase[ir]; rr.trBest = -1;
    rr.maxScoreMate[0] = rd.maxScoreMate[0]; rr.maxScoreMate[1] = rd.maxScoreMate[1]; rr.unmappedLength = rd.unmappedLength;
    if (rd.nWt > 0) rr.status |= STARAMD_ST_MAPPED_WINDOWS;
    u32 to = trBase[ir], eo = exBase[ir];
    if (rd.nTr > 0 && (u64)to + rd.nTr <= outTrCap && (u64)eo + rd.nEx <= outExCap) {
        u32 ord = 0;
        for (u32 w = 0; w < rd.nWin; w++) {
            const DWinOut d = B.wout[rd.winOffset + w];
            if (d.nTr == 0) continue;
            if ((i32)ord == rd.bestW) rr.trBest = (i32)(to - trBase[ir]);
            for (u32 k = 0; k < d.nTr; k++) {
                staramd_transcript t = B.trPool[d.trOffset + k];
                t.iW = ord; t.exonOffset += eo;
                outTr[to + k] = t;
            }
            for (u32 k = 0; k < d.nEx; k++) outEx[eo + k] = B.exPool[d.exOffset + k];
            to += d.nTr; eo += d.nEx; ord++;
        }
    }
    outReads[ir] = rr;
}
