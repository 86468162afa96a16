// Diagnostic: occupy a chosen number of compute units for a chosen time on a stream of the caller's choice, the way a
// communication library's ring kernels do during the backward of a data-parallel step (RCCL all-reduce: a few dozen long-lived
// workgroups of 256-512 threads).  Used by tools/cu_contention.py to measure what such co-resident kernels cost the
// one-resident-round grids of this library, and what hs_set_reserved_cus() buys back (profiles/r03_cu_contention.json).
#include "hs_device.h"

namespace hs {
namespace {

// each workgroup spins until `cycles` ticks of the 100 MHz wall clock have passed; `lds_bytes` of dynamic LDS and the launch
// bounds make the workgroup as heavy as the kernel it stands in for
__global__ void __launch_bounds__(512) occupy_kernel(int64_t ticks, unsigned* sink) {
    extern __shared__ unsigned char lds[];
    const int64_t t0 = wall_clock64();
    unsigned acc = 0;
    // claim a communication kernel's register footprint (128 VGPRs: four such waves leave a SIMD 3/4 of its register file)
    asm volatile("v_mov_b32 v127, 0" ::: "v127");
    while (wall_clock64() - t0 < ticks) {
        acc += (unsigned)lds[(threadIdx.x * 4) & 1023];  // (keeps the LDS allocation live)
        __builtin_amdgcn_s_sleep(8);
    }
    if (acc == 0xffffffffu && sink) *sink = acc;
}

}  // namespace
}  // namespace hs

extern "C" {

int hs_debug_occupy_cus(int n_workgroups, int threads, int lds_bytes, double microseconds, void* stream) {
    using namespace hs;
    HS_CHECK_ARG(n_workgroups >= 1 && n_workgroups <= 4096, "hs_debug_occupy_cus: 1..4096 workgroups");
    HS_CHECK_ARG(threads >= 64 && threads <= 512 && threads % 64 == 0, "hs_debug_occupy_cus: 64..512 threads, a multiple of 64");
    HS_CHECK_ARG(lds_bytes >= 1024 && lds_bytes <= 160 * 1024, "hs_debug_occupy_cus: 1 KB .. 160 KB of LDS");
    HS_CHECK_ARG(microseconds > 0 && microseconds <= 2e6, "hs_debug_occupy_cus: at most 2 s");
    static bool configured = false;
    if (!configured) {
        HS_HIP_CHECK(hipFuncSetAttribute((const void*)occupy_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        configured = true;
    }
    const int64_t ticks = (int64_t)(microseconds * 100.0);  // wall_clock64: 100 MHz
    hipLaunchKernelGGL(occupy_kernel, dim3(n_workgroups), dim3(threads), lds_bytes, (hipStream_t)stream, ticks, (unsigned*)nullptr);
    HS_LAUNCH_CHECK("occupy_kernel");
    return HS_OK;
}

}  // extern "C"
