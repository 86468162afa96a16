// Diagnostic: occupy a chosen number of compute units for a chosen time on a stream of the caller's choice, the way a
// communication library's ring kernels do during the backward of a data-parallel step (RCCL all-reduce: a few dozen long-lived
// workgroups of 256-512 threads).  Used by tools/cu_contention.py to measure what such co-resident kernels cost the
// one-resident-round grids of this library, and what hs_set_reserved_cus() buys back (profiles/archive_r01_r04/r03_cu_contention.json).
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

// The range-check rule the role-separated DMA of hs_gemm_nt (FAST) relies on: with a raw buffer descriptor the SCALAR offset of
// buffer_load (... lds) takes part in the bounds check like the vector offset, so an operand row beyond num_records reads zeros
// instead of memory behind the operand.  One wave: lane i loads a dword at voffset = 4 i with soffset = soff from a descriptor
// over `bytes` bytes, once into registers and once by LDS-DMA.
__global__ void __launch_bounds__(64) soffset_probe_kernel(const uint32_t* src, int bytes, int soff, uint32_t* out) {
#if defined(__HIP_DEVICE_COMPILE__)
    __shared__ uint32_t patch[64];
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, bytes, 0x00020000);
    const int lane = threadIdx.x;
    out[lane] = __builtin_amdgcn_raw_buffer_load_b32(r, lane * 4, soff, 0);
    patch[lane] = 0xdeadbeefu;
    __syncthreads();
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)patch, 4, lane * 4, soff, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    out[64 + lane] = patch[lane];
#endif
}

}  // namespace
}  // namespace hs

extern "C" {

int hs_debug_buffer_soffset_probe(const void* src, int bytes, int soffset, void* out128, void* stream) {
    using namespace hs;
    HS_CHECK_ARG(src && out128 && bytes >= 0 && soffset >= 0, "hs_debug_buffer_soffset_probe: bad argument");
    hipLaunchKernelGGL(soffset_probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (const uint32_t*)src, bytes, soffset, (uint32_t*)out128);
    HS_LAUNCH_CHECK("soffset_probe_kernel");
    return HS_OK;
}

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
