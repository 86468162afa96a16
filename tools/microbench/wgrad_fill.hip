// The global -> LDS fill of hs_linear_wgrad's 256 x 256 tile kernel WITHOUT the MFMAs, with the number of 32-token stages in
// flight as a knob: what the operand delivery alone costs for one weight-gradient launch (default: HEAL-SWIN-B stage-2 fc1,
// 98 304 rows, dY [rows, 2048], X [rows, 512]: 16 tiles x 16 token slices = 256 workgroups, the tiles of a slice on one XCD).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o wgrad_fill wgrad_fill.hip && ./wgrad_fill [rows n_out k_in]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <type_traits>

typedef __attribute__((address_space(3))) void lds_void;
constexpr int kTok = 32, TN = 256, TK = 256, NW = 8;
constexpr int YB = kTok * TN * 2, XB = kTok * TK * 2, STAGE = YB + XB;  // 16 + 16 KB
constexpr int YI = YB / 1024 / NW, XI = XB / 1024 / NW;

struct Geo {
    int tiles_k, tiles, slices, per_xcd, rows_per_slice, stagger;
    int mode;  // 0 = the kernel's pattern; 1 = without the chunk swizzle; 2 = dY tile only; 3 = X tile only;
               // 4 = every workgroup streams a PRIVATE contiguous region (no sharing); 5 = the 16 tiles of a slice all read the
               // SAME contiguous 32 KB per stage (pure sharing); 6 = as 0 but rows of the tile contiguous (tile-major copy)
};

template <int AH>  // stages in flight
__global__ void __launch_bounds__(512, 1) fill(const unsigned short* dy, const unsigned short* x, int rows, int n_out, int k_in, Geo g,
                                              unsigned* sink) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[(AH + 1) * STAGE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int v = (blockIdx.x & 7) * g.per_xcd + (blockIdx.x >> 3);
    if ((int)(blockIdx.x >> 3) >= g.per_xcd || v >= g.slices * g.tiles) return;
    const int slice = v / g.tiles, tile = v % g.tiles;
    const int tn = tile / g.tiles_k, tk = tile % g.tiles_k;
    const int n0 = tn * TN, k0 = tk * TK;
    const long m_begin = (long)slice * g.rows_per_slice;
    long m_end = m_begin + g.rows_per_slice;
    if (m_end > rows) m_end = rows;
    const int m_len = (int)(m_end - m_begin);
    const int nst = (m_len + kTok - 1) / kTok;
    const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc((void*)(dy + m_begin * n_out), 0, m_len * n_out * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)(x + m_begin * k_in), 0, m_len * k_in * 2, 0x00020000);
    constexpr int YCH = TN * 2 / 16, XCH = TK * 2 / 16;
    int voff_y[YI], voff_x[XI];
    // stagger: tile t of a slice starts `stagger * t` stages into the slice (wrapping), so that the workgroups sharing a
    // row panel are NOT in lock step
    // stagger >= 100: offset = (tile % (stagger / 100)) * (stagger % 100) stages (small phase groups that stay inside L2)
    const int st0 = g.stagger >= 100 ? ((tile % (g.stagger / 100)) * (g.stagger % 100)) % nst : (g.stagger ? (tile * g.stagger) % nst : 0);
    int ystep = kTok * n_out * 2, xstep = kTok * k_in * 2;
#pragma unroll
    for (int j = 0; j < YI; ++j) {
        const int p = (wave * YI + j) * 64 + lane, row = p / YCH, pc = p % YCH;
        voff_y[j] = row * n_out * 2 + n0 * 2 + (((g.mode == 1 ? pc : pc ^ ((row & 3) << 2))) << 4);
    }
#pragma unroll
    for (int j = 0; j < XI; ++j) {
        const int p = (wave * XI + j) * 64 + lane, row = p / XCH, pc = p % XCH;
        voff_x[j] = row * k_in * 2 + k0 * 2 + (((g.mode == 1 ? pc : pc ^ ((row & 3) << 2))) << 4);
    }
    if (g.mode == 6) {  // dY only: tile tn reads rows 4 tn .. 4 tn + 3 of the stage, full width (16 KB contiguous when n_out = 2048)
#pragma unroll
        for (int j = 0; j < YI; ++j) voff_y[j] = tn * (kTok / 8) * n_out * 2 + ((wave * YI + j) * 64 + lane) * 16;
    }
    if (g.mode == 4 || g.mode == 5) {  // contiguous streams inside the dY slice: 16 KB + 16 KB per stage
        const int region = g.mode == 4 ? tile * (m_len / g.tiles) * n_out * 2 : 0;  // private part of the slice / shared
#pragma unroll
        for (int j = 0; j < YI; ++j) voff_y[j] = region + ((wave * YI + j) * 64 + lane) * 16;
#pragma unroll
        for (int j = 0; j < XI; ++j) voff_x[j] = region + YB + ((wave * XI + j) * 64 + lane) * 16;
        ystep = xstep = STAGE;
    }
    int cur = st0;  // stage index the next issue reads
    auto issue = [&](int b) {
        unsigned char* base = smem + b * STAGE;
#pragma unroll
        for (int j = 0; j < YI; ++j)
            if (g.mode != 3)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_y, (lds_void*)(base + (wave * YI + j) * 1024), 16, voff_y[j] + cur * ystep, 0, 0, 0);
#pragma unroll
        for (int j = 0; j < XI; ++j)
            if (g.mode != 2 && g.mode != 6)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(g.mode >= 4 ? rs_y : rs_x, (lds_void*)(base + YB + (wave * XI + j) * 1024), 16,
                                                         voff_x[j] + cur * xstep, 0, 0, 0);
        cur = cur + 1 == nst ? 0 : cur + 1;
    };
    for (int s = 0; s < AH; ++s) issue(s);
    int nb = AH;
    for (int t = 0; t < nst; ++t) {
        if (AH == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (g.mode == 2 || g.mode == 3 || g.mode == 6) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((AH - 1) * YI) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((AH - 1) * (YI + XI)) : "memory");
        __builtin_amdgcn_s_barrier();
        issue(nb);  // (the tail re-reads the first stages: a few per cent extra traffic, keeps the loop uniform)
        nb = nb == AH ? 0 : nb + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) sink[blockIdx.x] = *(unsigned*)(smem + 64);
}

#define CHECK(x)                                                 \
    do {                                                         \
        hipError_t e = (x);                                      \
        if (e != hipSuccess) {                                   \
            printf("%s failed: %s\n", #x, hipGetErrorString(e)); \
            return 1;                                            \
        }                                                        \
    } while (0)

template <typename K>
int run(const char* name, K kern, const unsigned short* dy, const unsigned short* x, int rows, int n_out, int k_in, Geo g, unsigned* sink,
        unsigned char* flush, size_t flush_bytes) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const int grid = 8 * g.per_xcd;
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
        CHECK(hipMemsetAsync(flush, rep, flush_bytes, 0));  // evict L2 / MALL between repetitions
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(512), 0, 0, dy, x, rows, n_out, k_in, g, sink);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (rep) best = ms < best ? ms : best;
    }
    const double l1 = (double)g.slices * g.tiles * ((double)g.rows_per_slice * (TN + TK) * 2);
    const double hbm = (double)rows * (n_out + k_in) * 2;
    printf("%-26s stagger %3d  %7.1f us   L2->LDS %6.2f TB/s (%5.1f B/clk/CU)   compulsory HBM %5.2f TB/s\n", name, g.stagger, best * 1e3,
           l1 / (best * 1e-3) / 1e12, l1 / 256 / (best * 1e-3) / 2.1e9, hbm / (best * 1e-3) / 1e12);
    return 0;
}

int main(int argc, char** argv) {
    const int rows = argc > 1 ? atoi(argv[1]) : 98304, n_out = argc > 2 ? atoi(argv[2]) : 2048, k_in = argc > 3 ? atoi(argv[3]) : 512;
    const int pad = argc > 5 ? atoi(argv[5]) : 0;
    Geo g;
    g.tiles_k = k_in / TK;
    g.tiles = (n_out / TN) * g.tiles_k;
    g.slices = 256 / g.tiles;
    g.rows_per_slice = ((rows + g.slices - 1) / g.slices + kTok - 1) / kTok * kTok;
    g.per_xcd = (g.slices * g.tiles + 7) / 8;
    unsigned short *dy, *x;
    unsigned* sink;
    unsigned char* flush;
    const size_t flush_bytes = (size_t)1 << 30;
    CHECK(hipMalloc(&dy, (size_t)rows * (n_out + pad) * 2));
    CHECK(hipMalloc(&x, (size_t)rows * k_in * 2));
    CHECK(hipMemset(dy, 1, (size_t)rows * (n_out + pad) * 2));
    CHECK(hipMemset(x, 1, (size_t)rows * k_in * 2));
    CHECK(hipMalloc(&sink, 4096 * 4));
    CHECK(hipMalloc(&flush, flush_bytes));
    printf("rows %d n_out %d k_in %d: %d tiles x %d slices, %d rows per slice\n", rows, n_out, k_in, g.tiles, g.slices, g.rows_per_slice);
    const int only = argc > 4 ? atoi(argv[4]) : -1;
    if (pad) printf("dY row stride padded by %d elements\n", pad);
    for (int mode = 0; mode <= 6; ++mode) {
        if (only >= 0 && mode != only) continue;
        g.mode = mode;
        for (int stg : {0, 201, 202, 401, 402, 801, 1601}) {
        g.stagger = argc > 6 ? atoi(argv[6]) : stg;
        if (argc > 6 && stg) continue;
        printf("mode %d\n", mode);
        run("2 stages in flight", fill<2>, dy, x, rows, n_out + pad, k_in, g, sink, flush, flush_bytes);
        run("3 stages in flight", fill<3>, dy, x, rows, n_out + pad, k_in, g, sink, flush, flush_bytes);
        }
    }
    return 0;
}
