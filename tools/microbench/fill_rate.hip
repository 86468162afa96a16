// Per-CU global -> LDS fill rate on gfx950: LDS-DMA (buffer_load_dwordx4 ... lds) against register staging
// (global_load_dwordx4 -> VGPR -> ds_write_b128), from cache-resident data, 1 workgroup of 8 waves per CU, 32 KB stages,
// NS stages in flight.  No consumers: this is the ceiling the GEMM / weight-gradient main loops fill their tiles at.
//   hipcc --offload-arch=gfx950 -O3 -o fill_rate fill_rate.hip && ./fill_rate
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

constexpr int kStage = 32768;  // bytes per stage and workgroup

template <int NS, int WAVES>
__global__ void __launch_bounds__(WAVES * 64, 1) fill_dma(const unsigned char* src, int region, int iters, unsigned* sink) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[(NS + 1) * kStage];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    constexpr int PER = kStage / 1024 / WAVES;  // 1-KB DMA instructions per wave and stage
    const unsigned char* mine = src + (size_t)blockIdx.x * region;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)mine, 0, region, 0x00020000);
    int off = 0;
    auto issue = [&](int b) {
#pragma unroll
        for (int j = 0; j < PER; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(smem + b * kStage + (wave * PER + j) * 1024), 16,
                                                     off + (wave * PER + j) * 1024 + lane * 16, 0, 0, 0);
        off += kStage;
        if (off >= region) off = 0;
    };
    for (int s = 0; s < NS; ++s) issue(s);
    int b = 0, nb = NS;
    for (int t = 0; t < iters; ++t) {
        if constexpr (NS == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if constexpr (NS == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PER) : "memory");
        __builtin_amdgcn_s_barrier();
        issue(nb);
        b = b == NS ? 0 : b + 1;
        nb = nb == NS ? 0 : nb + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) sink[blockIdx.x] = *(unsigned*)(smem + 64);
}

template <int NS, int WAVES>
__global__ void __launch_bounds__(WAVES * 64, 1) fill_reg(const unsigned char* src, int region, int iters, unsigned* sink) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * kStage];
    const int tid = threadIdx.x;
    constexpr int PER = kStage / 16 / (WAVES * 64);  // 16-byte loads per thread and stage
    const unsigned char* mine = src + (size_t)blockIdx.x * region;
    u32x4 r[NS][PER];
    int off = 0;
    auto issue = [&](auto sc) {
        constexpr int s = decltype(sc)::value;
#pragma unroll
        for (int j = 0; j < PER; ++j) r[s][j] = *(const u32x4*)(mine + off + (j * WAVES * 64 + tid) * 16);
        off += kStage;
        if (off >= region) off = 0;
    };
    auto drain = [&](auto sc, int buf) {
        constexpr int s = decltype(sc)::value;
#pragma unroll
        for (int j = 0; j < PER; ++j) *(u32x4*)(smem + buf * kStage + (j * WAVES * 64 + tid) * 16) = r[s][j];
    };
    using std::integral_constant;
    issue(integral_constant<int, 0>{});
    if constexpr (NS > 1) issue(integral_constant<int, 1>{});
    if constexpr (NS > 2) issue(integral_constant<int, 2>{});
    for (int t = 0; t < iters; t += NS) {
        drain(integral_constant<int, 0>{}, 0);
        issue(integral_constant<int, 0>{});
        __builtin_amdgcn_s_barrier();
        if constexpr (NS > 1) {
            drain(integral_constant<int, 1>{}, 1);
            issue(integral_constant<int, 1>{});
            __builtin_amdgcn_s_barrier();
        }
        if constexpr (NS > 2) {
            drain(integral_constant<int, 2>{}, 0);
            issue(integral_constant<int, 2>{});
            __builtin_amdgcn_s_barrier();
        }
    }
    __syncthreads();
    unsigned acc = 0;
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int j = 0; j < PER; ++j) acc ^= r[s][j][0];
    if (acc == 0x12345678u || tid == 0) sink[blockIdx.x] = acc ^ *(unsigned*)(smem + 64);
}

#define CHECK(x)                                                                  \
    do {                                                                          \
        hipError_t e = (x);                                                       \
        if (e != hipSuccess) {                                                    \
            printf("%s failed: %s\n", #x, hipGetErrorString(e));                  \
            return 1;                                                             \
        }                                                                         \
    } while (0)

template <typename K>
int run(const char* name, K kern, int waves, const unsigned char* src, int region, unsigned* sink, int grid) {
    const int iters = 3000;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(waves * 64), 0, 0, src, region, 300, sink);
    CHECK(hipDeviceSynchronize());
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(waves * 64), 0, 0, src, region, iters, sink);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best;
    }
    const double bytes = (double)grid * iters * kStage;
    // ~2.1 GHz assumed for the per-clock figure
    printf("%-34s region %7d B/WG  %8.1f us  %7.2f TB/s aggregate  %6.1f GB/s per CU  ~%5.1f B/clk/CU\n", name, region, best * 1e3,
           bytes / (best * 1e-3) / 1e12, bytes / grid / (best * 1e-3) / 1e9, bytes / grid / (best * 1e-3) / 2.1e9);
    return 0;
}

int main() {
    const int grid = 256;
    const size_t total = (size_t)grid * (4 << 20);
    unsigned char* src;
    unsigned* sink;
    CHECK(hipMalloc(&src, total));
    CHECK(hipMemset(src, 1, total));
    CHECK(hipMalloc(&sink, grid * 4));
    for (int region : {65536, 1 << 20, 4 << 20}) {  // per workgroup: L2-resident (2 MB per XCD) ... MALL-resident ... 1 GB total
        run("dma  8 waves, 1 stage in flight", fill_dma<1, 8>, 8, src, region, sink, grid);
        run("dma  8 waves, 2 stages in flight", fill_dma<2, 8>, 8, src, region, sink, grid);
        run("dma  8 waves, 3 stages in flight", fill_dma<3, 8>, 8, src, region, sink, grid);
        run("dma  4 waves, 2 stages in flight", fill_dma<2, 4>, 4, src, region, sink, grid);
        run("reg  8 waves, 1 stage in flight", fill_reg<1, 8>, 8, src, region, sink, grid);
        run("reg  8 waves, 2 stages in flight", fill_reg<2, 8>, 8, src, region, sink, grid);
        run("reg  8 waves, 3 stages in flight", fill_reg<3, 8>, 8, src, region, sink, grid);
        run("reg  4 waves, 2 stages in flight", fill_reg<2, 4>, 4, src, region, sink, grid);
    }
    return 0;
}
