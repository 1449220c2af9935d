// dynamic LDS of the emulated kernels (`extern __shared__` in k_window.hip / stitch_common.h): every block starts at offset 0 of the
// one array, as on the device; zeroed by the launcher for the size the launch asks for.  Test infrastructure (see emu.h).
#include <cstdint>
thread_local uint32_t ldsTab[160 * 1024 / 4];
thread_local uint32_t ldsReads[160 * 1024 / 4];
