// k_gather.hip -- kernel 4: order the per-window transcript blocks by read and produce the C-ABI result arrays.
// (No counterpart in the reference: there the per-thread ReadAlign object owns trAll[][] directly.)
// Output order is deterministic -- reads in batch order, windows in window order, transcripts best-first --
// so result buffers can be compared byte for byte between runs and against the oracle.
#include "dev.h"

// exclusive scan of (nTr, nEx) over reads, two levels: (1) every block of 256 reads scans itself (coalesced over the batch) and
// leaves its totals, (2) one block scans the block totals, (3) the gather adds its block's base.
extern "C" __global__ void __launch_bounds__(256) k_scan_local(DevBatch B, u32 *trBase, u32 *exBase, u32 *blockTot) {
    if (B.cursors[CUR_FLAGS] != 0) return;          // a pool overflowed in an earlier kernel: the host grows it and re-runs the batch
    __shared__ u32 sT[256], sE[256];
    const u32 t = threadIdx.x, ir = blockIdx.x * 256u + t;
    u32 vT = 0, vE = 0;
    if (ir < B.nReads) { vT = B.reads[ir].nTr; vE = B.reads[ir].nEx; }
    sT[t] = vT; sE[t] = vE;
    __syncthreads();
    for (u32 off = 1; off < 256; off <<= 1) {
        u32 aT = t >= off ? sT[t - off] : 0, aE = t >= off ? sE[t - off] : 0;
        __syncthreads();
        sT[t] += aT; sE[t] += aE;
        __syncthreads();
    }
    if (ir < B.nReads) { trBase[ir] = sT[t] - vT; exBase[ir] = sE[t] - vE; }
    if (t == 255) { blockTot[2 * blockIdx.x] = sT[255]; blockTot[2 * blockIdx.x + 1] = sE[255]; }
}

extern "C" __global__ void __launch_bounds__(1024) k_scan_offsets(DevBatch B, u32 *blockTot, u32 nBlocks, u32 *totals) {
    if (B.cursors[CUR_FLAGS] != 0) return;
    __shared__ u32 sT[1024], sE[1024];
    u32 t = threadIdx.x, n = nBlocks;
    u32 chunk = (n + 1023) / 1024;
    u32 lo = min(n, t * chunk), hi = min(n, lo + chunk);
    u32 aT = 0, aE = 0;
    for (u32 i = lo; i < hi; i++) { aT += blockTot[2 * i]; aE += blockTot[2 * i + 1]; }
    sT[t] = aT; sE[t] = aE;
    __syncthreads();
    for (u32 off = 1; off < 1024; off <<= 1) {
        u32 vT = t >= off ? sT[t - off] : 0, vE = t >= off ? sE[t - off] : 0;
        __syncthreads();
        sT[t] += vT; sE[t] += vE;
        __syncthreads();
    }
    u32 bT = sT[t] - aT, bE = sE[t] - aE;
    for (u32 i = lo; i < hi; i++) { u32 xT = blockTot[2 * i], xE = blockTot[2 * i + 1]; blockTot[2 * i] = bT; blockTot[2 * i + 1] = bE; bT += xT; bE += xE; }
    if (t == 1023) { totals[0] = sT[1023]; totals[1] = sE[1023]; }
}

extern "C" __global__ void __launch_bounds__(256) k_gather(DevBatch B, const u32 *trBaseLocal, const u32 *exBaseLocal, const u32 *blockBase,
                                                          staramd_read_result *outReads, staramd_transcript *outTr, u32 outTrCap,
                                                          staramd_exon *outEx, u32 outExCap) {
    if (B.cursors[CUR_FLAGS] != 0) return;          // a pool overflowed in an earlier kernel: the host grows it and re-runs the batch
    u32 ir = blockIdx.x * blockDim.x + threadIdx.x;
    if (ir >= B.nReads) return;
    const DRead rd = B.reads[ir];
    staramd_read_result rr;
    const u32 trBase0 = trBaseLocal[ir] + blockBase[2 * (ir >> 8)], exBase0 = exBaseLocal[ir] + blockBase[2 * (ir >> 8) + 1];
    rr.status = rd.status; rr.nW = rd.nWt; rr.nTr = rd.nTr; rr.trOffset = trBase0; rr.trBest = -1;
    rr.maxScoreMate[0] = rd.maxScoreMate[0]; rr.maxScoreMate[1] = rd.maxScoreMate[1]; rr.unmappedLength = rd.unmappedLength;
    if (rd.nWt > 0) rr.status |= STARAMD_ST_MAPPED_WINDOWS;
    u32 to = trBase0, eo = exBase0;
    if (rd.nTr > 0 && (u64)to + rd.nTr <= outTrCap && (u64)eo + rd.nEx <= outExCap) {
        u32 ord = 0;
        for (u32 w = 0; w < rd.nWin; w++) {
            const DWinOut d = B.wout[rd.winOffset + w];
            if (d.nTr == 0) continue;
            if ((i32)ord == rd.bestW) rr.trBest = (i32)(to - trBase0);
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
