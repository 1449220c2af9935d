// selftest.cpp -- the cross-lane primitives of the wavefront emulator (oracle/wave_emul/emu.h; test infrastructure) against their definition:
// DPP reductions (row_shr / row_bcast / wave_shl / wave_shr with row and bank masks), ballot, mbcnt prefix counts, readlane / readfirstlane,
// shuffles, __syncthreads -- through the helpers of star_amd/csrc/engine/dev.h, which is how the kernels use them.  Prints "primitives OK".
#include <hip/hip_runtime.h>
#include "dev.h"
#include <cstdio>
__global__ void k(u32 *out, u32 seed) {
    u32 lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    u32 v = (lane * 2654435761u + seed + w * 977u) >> 7;
    u32 mx = waveMaxU32(v), sm = waveSumU32(v & 0xffff);
    u64 b = __ballot((v & 3) == 1);
    u32 cb = cntBelow(b);
    u64 g = laneGet64(((u64)v << 32) | lane, (seed + w) & 63);
    u32 f = first32(v + 5);
    int prev = __builtin_amdgcn_update_dpp(-9, (int)v, 0x138, 0xf, 0xf, false), next = __builtin_amdgcn_update_dpp(-7, (int)v, 0x130, 0xf, 0xf, false);
    u32 x = (u32)__shfl_xor((int)v, 5, 64);
    u32 *o = out + threadIdx.x * 16;
    o[0] = v; o[1] = mx; o[2] = sm; o[3] = (u32)b; o[4] = (u32)(b >> 32); o[5] = cb; o[6] = (u32)g; o[7] = (u32)(g >> 32); o[8] = f; o[9] = (u32)prev; o[10] = (u32)next; o[11] = x; o[12] = laneId();
    __shared__ u32 sh[256];
    sh[threadIdx.x] = v; __syncthreads();
    o[13] = sh[(threadIdx.x + 100) & 255];
}
int main() {
    static u32 out[256 * 16];
    int bad = 0;
    for (u32 seed = 1; seed < 40; seed++) {
        hipLaunchKernelGGL(k, dim3(2), dim3(256), 0, 0, out, seed);
        for (int w = 0; w < 4; w++) {
            u32 v[64]; u32 mx = 0, sm = 0; u64 b = 0;
            for (int l = 0; l < 64; l++) { v[l] = out[(w * 64 + l) * 16]; mx = v[l] > mx ? v[l] : mx; sm += v[l] & 0xffff; if ((v[l] & 3) == 1) b |= 1ull << l; }
            for (int l = 0; l < 64; l++) {
                u32 *o = out + (w * 64 + l) * 16; u32 src = (seed + w) & 63;
                bool ok = o[1] == mx && o[2] == sm && o[3] == (u32)b && o[4] == (u32)(b >> 32) && o[5] == (u32)__builtin_popcountll(b & ((1ull << l) - 1)) && o[6] == src && o[7] == v[src] && o[8] == v[0] + 5
                          && o[9] == (l ? v[l - 1] : (u32)-9) && o[10] == (l < 63 ? v[l + 1] : (u32)-7) && o[11] == v[l ^ 5] && o[12] == (u32)l && o[13] == out[((w * 64 + l + 100) & 255) * 16];
                if (!ok && bad++ < 5) printf("seed %u wave %d lane %d: mx %u/%u sm %u/%u cb %u g %u,%u f %u prev %d next %d\n", seed, w, l, o[1], mx, o[2], sm, o[5], o[6], o[7], o[8], (int)o[9], (int)o[10]);
            }
        }
    }
    printf("%s (%d bad)\n", bad ? "FAIL" : "primitives OK", bad);
    return bad != 0;
}
