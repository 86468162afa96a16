// LDS read rate per CU on gfx950: ds_read_b64_tr_b16 (the transposing fragment read of hs_linear_wgrad / the attention backward)
// against ds_read_b64 and ds_read_b128, 8 waves per CU, conflict-free addresses, optionally with an LDS-DMA stream
// (buffer_load_dwordx4 ... lds from cache-resident data) writing into another part of the LDS at the same time.
#include <hip/hip_runtime.h>

#include <cstdio>

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

template <int KIND, bool DMA>  // 0: ds_read_b64_tr_b16, 1: ds_read_b64, 2: ds_read_b128
__global__ void __launch_bounds__(512, 1) lds_read(const unsigned char* src, int iters, unsigned* sink) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[128 * 1024];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 32 * 1024 / 4; i += 512) ((unsigned*)smem)[i] = i;
    __syncthreads();
    const unsigned base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
    // each wave walks its own 4 KB; lane addresses: consecutive 8 (or 16) bytes -> conflict-free
    const unsigned a = base + wave * 4096 + lane * (KIND == 2 ? 16 : 8);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(src + (size_t)blockIdx.x * 65536), 0, 65536, 0x00020000);
    u32x4 acc = {0, 0, 0, 0};
    int off = 0;
    for (int t = 0; t < iters; ++t) {
        if (DMA) {  // 32 KB per iteration and workgroup = 4 x 1 KB per wave, into the upper 64 KB
#pragma unroll
            for (int j = 0; j < 4; ++j)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(smem + 65536 + (t & 1) * 32768 + (wave * 4 + j) * 1024), 16,
                                                         off + (wave * 4 + j) * 1024 + lane * 16, 0, 0, 0);
            off = off ? 0 : 32768;
        }
        // 24 reads of 512 B (or 12 of 1 KB) per wave = 96 KB per workgroup and iteration: one 32-token stage of the 256 x 256 tile
        if (KIND == 2) {
            u32x4 v[12];
#pragma unroll
            for (int j = 0; j < 12; ++j) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v[j]) : "v"(a), "n"((j % 4) * 1024));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int j = 0; j < 12; ++j) acc ^= v[j];
        } else {
            u32x2 v[24];
#pragma unroll
            for (int j = 0; j < 24; ++j) {
                if (KIND == 0) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v[j]) : "v"(a), "n"((j % 8) * 512));
                else asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v[j]) : "v"(a), "n"((j % 8) * 512));
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int j = 0; j < 24; ++j) {
                acc[0] ^= v[j][0];
                acc[1] ^= v[j][1];
            }
        }
        if (DMA && (t & 1)) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345u || tid == 0) sink[blockIdx.x] = acc[0];
}

template <typename K>
void run(const char* name, K kern, const unsigned char* src, unsigned* sink) {
    const int iters = 4000, grid = 256;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), 0, 0, src, 100, sink);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), 0, 0, src, iters, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)iters * 96 * 1024;
    printf("%-44s %8.1f us  %6.1f ns per 96 KB 'stage'  LDS reads %6.1f GB/s per CU  ~%5.1f B/clk/CU\n", name, ms * 1e3, ms * 1e6 / iters,
           bytes / (ms * 1e-3) / 1e9, bytes / (ms * 1e-3) / 2.1e9);
}

int main() {
    unsigned char* src;
    unsigned* sink;
    hipMalloc(&src, 256 * 65536);
    hipMemset(src, 1, 256 * 65536);
    hipMalloc(&sink, 4096);
    run("ds_read_b64_tr_b16", lds_read<0, false>, src, sink);
    run("ds_read_b64", lds_read<1, false>, src, sink);
    run("ds_read_b128", lds_read<2, false>, src, sink);
    run("ds_read_b64_tr_b16 + 32 KB LDS-DMA per stage", lds_read<0, true>, src, sink);
    run("ds_read_b64 + 32 KB LDS-DMA per stage", lds_read<1, true>, src, sink);
    run("ds_read_b128 + 32 KB LDS-DMA per stage", lds_read<2, true>, src, sink);
    return 0;
}
