// gather_ceiling.hip -- measured ceiling for DEPENDENT random 8-byte gathers on MI355X (the access pattern of the seed
// search: SAindex -> packed SA -> genome, every address depends on the previous load).  Each lane walks a chain
// idx = mix(table[idx]) over a table of `mb` MiB; reports gathers/s and the implied 64-byte-sector bandwidth.
// Build: hipcc --offload-arch=gfx950 -O3 tools/gather_ceiling.hip -o tools/gather_ceiling     Run: tools/gather_ceiling [MiB] [steps]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
__global__ void __launch_bounds__(256) chase(const uint64_t *t, uint64_t mask, int steps, uint64_t *out) {
    uint64_t i = (blockIdx.x * 256ull + threadIdx.x) * 0x9E3779B97F4A7C15ull;
    uint64_t acc = 0;
    for (int s = 0; s < steps; s++) { uint64_t v = t[i & mask]; acc += v; i = (v ^ (i >> 7)) * 0xD6E8FEB86659FD93ull + s; }
    out[blockIdx.x * 256ull + threadIdx.x] = acc;
}
int main(int argc, char **argv) {
    size_t mb = argc > 1 ? strtoull(argv[1], 0, 10) : 2048; int steps = argc > 2 ? atoi(argv[2]) : 256;
    size_t n = mb * 1024 * 1024 / 8; size_t p2 = 1; while (p2 * 2 <= n) p2 *= 2; n = p2;
    uint64_t *t, *out; hipMalloc(&t, n * 8); 
    uint64_t *h = (uint64_t *)malloc(n * 8); uint64_t x = 88172645463325252ull;
    for (size_t k = 0; k < n; k++) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; h[k] = x; }
    hipMemcpy(t, h, n * 8, hipMemcpyHostToDevice);
    for (int blocksPerCU : {1, 2, 4, 8}) {
        int blocks = 256 * blocksPerCU; hipMalloc(&out, blocks * 256ull * 8);
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        hipLaunchKernelGGL(chase, dim3(blocks), dim3(256), 0, 0, t, n - 1, 16, out); hipDeviceSynchronize();
        hipEventRecord(a); hipLaunchKernelGGL(chase, dim3(blocks), dim3(256), 0, 0, t, n - 1, steps, out); hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        double g = (double)blocks * 256 * steps / (ms * 1e-3);
        printf("table %zu MiB, %d lanes in flight: %.2f G gathers/s = %.0f GB/s of 64-B sectors, %.2f us per dependent step\n", n * 8 >> 20, blocks * 256, g / 1e9, g * 64 / 1e9, ms * 1e3 / steps);
        hipFree(out);
    }
    return 0;
}
