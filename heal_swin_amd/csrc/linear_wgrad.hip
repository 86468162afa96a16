// Weight (and bias) gradient of the path's Linear layers:   dW[n, k] = sum_m dY[m, n] * X[m, k],   db[n] = sum_m dY[m, n]
// for bf16 activations (qkv, proj, fc1, fc2, PatchMerging.reduction, PatchExpand.expand, concat_back_dim), fp32 results
// (the master weights are fp32, so the gradient never passes through bf16).
//
// Shape regime: the output is tiny (128x128 .. 2048x512) and the reduction runs over every token of the batch
// (98 304 .. 1 572 864 rows): a "TN" GEMM whose only parallelism is the reduction axis.  Structure:
//   * workgroup = one 128 x 128 output tile x one SLICE of token rows; 4 waves in 2 x 2, each a 64 x 64 sub-tile
//     (2 x 2 v_mfma_f32_32x32x16_bf16 accumulators); partial tiles go to a workspace and a second kernel sums the slices
//     (deterministic, no atomics);
//   * both MFMA operands need 8 consecutive TOKENS of one column per lane, but tokens are the row index of dY and X in
//     memory: the 32-token tiles are staged row-major in LDS (row stride 320 B: the 4 rows of a transposing read fall
//     on disjoint banks) and fragments come from ds_read_b64_tr_b16 (hardware 4x16 transpose);
//   * register-staged double buffering: the next tile's global loads are issued before the current tile's MFMAs and
//     written to the other LDS buffer after them (one barrier per 32 tokens);
//   * all tiles of one token slice get consecutive ids on ONE XCD (id % 8), so dY / X slices are re-read from that XCD's
//     L2, not from HBM: HBM traffic stays at one pass over dY and X;
//   * the bias gradient falls out of the dY staging loads (column sums in registers), only in the k-tile-0 workgroups.
// Roofline: flop/byte = N*K/(N+K): HBM-bound at C <= 256 (stage 0/1), MFMA-bound from C = 512.
#include <cstdlib>
#include <type_traits>
#include <utility>

#include "hs_gelu.h"

namespace hs {
namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

constexpr int kTileK = 128;           // output tile extent along k_in
constexpr int kTok = 32;              // tokens per staged tile
constexpr int kMaxSlices = 2048;
constexpr int kReduceChunks = 32;     // first-stage groups of the slice reduction
// LDS row stride: data bytes + 64 B skew, so the 4 token rows of a transposing read fall on disjoint banks
__host__ __device__ constexpr int row_stride(int cols) { return cols * 2 + 64; }

template <int LD>
__device__ __forceinline__ bf16x8 tr_frag(const unsigned char* tile, int tok0, int col0, int lane) {
    const int L = lane & 15, nblk = (lane >> 4) & 1;
    const unsigned char* a = tile + (tok0 + (L >> 2)) * LD + (col0 + nblk * 16 + (L & 3) * 4) * 2;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)a);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a + 4 * LD));
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
}

struct Geometry {
    int tile_n, tile_k, tiles_n, tiles_k, tiles, slices, chunks, per_xcd, dma, reads_first;
    int64_t rows_per_slice;
};

inline Geometry geometry_for(int64_t rows, int n_out, int k_in, int tile_k, int wgs_per_cu, int tile_n = 0, int elt = 2) {
    Geometry g;
    g.tile_n = tile_n ? tile_n : ((n_out % 256 == 0 || n_out >= 512) ? 256 : 128);
    g.tile_k = tile_k;
    g.tiles_n = (n_out + g.tile_n - 1) / g.tile_n;
    g.tiles_k = (k_in + tile_k - 1) / tile_k;
    g.tiles = g.tiles_n * g.tiles_k;
    // one resident round of equal workgroups over the CUs this library may plan for (256 minus hs_set_reserved_cus()); at
    // least 512 tokens per slice
    int64_t want = ((int64_t)usable_cus() * wgs_per_cu) / g.tiles;
    int64_t max_by_rows = (rows + 511) / 512;
    if (want > max_by_rows) want = max_by_rows;
    if (want > kMaxSlices) want = kMaxSlices;
    if (want < 1) want = 1;
    g.slices = (int)want;
    int64_t rps = (rows + g.slices - 1) / g.slices;
    g.rows_per_slice = ((rps + kTok - 1) / kTok) * kTok;
    g.chunks = g.slices > 2 * kReduceChunks ? kReduceChunks : 1;
    g.per_xcd = (g.slices * g.tiles + 7) / 8;
    // the LDS-DMA kernels address a slice through 32-bit buffer offsets
    g.dma = g.rows_per_slice * (int64_t)(n_out > k_in ? n_out : k_in) * elt < ((int64_t)1 << 31);
    // stage schedule of the LDS-DMA kernels: 3 = next stage's DMA issued behind the fragment reads + the second wave group half a
    // stage out of phase (8-wave tile).  (0 = DMA before the reads, 1 = behind them, 2 = behind the first MFMA group: the schedules
    // round 2 measured against it, profiles/archive_r01_r04/r02_wgrad_schedule_ab.txt)
    g.reads_first = 3;
    return g;
}

// Tile shapes: 256 x 256 (8 waves, one workgroup per CU) when both extents allow it -- a third less L2 -> LDS fill traffic
// per flop than 256 x 128, and the fill path is what bounds the compute-heavy shapes; else 256 x 128 or 128 x 128 (4 waves,
// two workgroups per CU).  HS_WGRAD_VARIANT=0 selects the register-staged kernel (A/B measurements).
inline Geometry make_geometry(int64_t rows, int n_out, int k_in) {
    static const int variant = getenv("HS_WGRAD_VARIANT") ? atoi(getenv("HS_WGRAD_VARIANT")) : 1;
    if (variant == 1 && k_in % 256 == 0 && (n_out % 256 == 0 || n_out >= 512)) {
        const Geometry g = geometry_for(rows, n_out, k_in, 256, 1);
        if (g.dma) return g;
    }
    Geometry g = geometry_for(rows, n_out, k_in, kTileK, 2);
    if (variant != 1) g.dma = 0;
    return g;
}

// fp32 activations: 128 x 128 tiles, 16-token stages (16 KB), three workgroups per CU
inline Geometry make_geometry_f32(int64_t rows, int n_out, int k_in) { return geometry_for(rows, n_out, k_in, kTileK, 3, 128, 4); }

// block id -> (slice, tile).  The dispatcher places block b on XCD b % 8; XCD x takes the contiguous range
// [x * per_xcd, (x + 1) * per_xcd) of slice-major work ids, so the tiles of one token slice run on one XCD (two at a range
// boundary) and re-read that slice's dY / X rows from its L2 rather than from HBM.  Returns false for the padding blocks.
__device__ __forceinline__ bool block_to_work(const Geometry& g, int b, int& slice, int& tile) {
    const int v = (b & 7) * g.per_xcd + (b >> 3);
    slice = v / g.tiles;
    tile = v % g.tiles;
    return (b >> 3) < g.per_xcd && v < g.slices * g.tiles;
}

// NB = 32-row blocks per wave along n: the workgroup tile is (64*NB) x 128, waves in 2 x 2, each (32*NB) x 64
template <int NB>
__global__ void __launch_bounds__(256, 2) wgrad_kernel(const uint16_t* __restrict__ dy, const uint16_t* __restrict__ x,
                                                       float* __restrict__ part_w, float* __restrict__ part_b, int64_t rows,
                                                       int n_out, int k_in, Geometry g) {
    constexpr int TN = 64 * NB;                      // tile extent along n_out
    constexpr int LDY = row_stride(TN), LDX = row_stride(kTileK);
    constexpr int YB = kTok * LDY, XB = kTok * LDX;  // bytes of one staged dY / X tile
    constexpr int YCH = TN / 8;                      // 16-byte chunks per dY tile row
    constexpr int YPASS = (kTok * YCH) / 256;        // staging passes over the dY tile (2 or 4)
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * (YB + XB)];
    auto ytile = [&](int b) { return smem + b * (YB + XB); };
    auto xtile = [&](int b) { return smem + b * (YB + XB) + YB; };
    float* bred = (float*)smem;  // bias partials [256 / YCH row groups][TN], reuses the tiles after the main loop

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5;
    const int wn = wave >> 1, wk = wave & 1;  // wave's (32*NB) x 64 sub-tile inside the tile

    int slice, tile;
    if (!block_to_work(g, blockIdx.x, slice, tile)) return;
    const int tn = tile / g.tiles_k, tk = tile % g.tiles_k;
    const int n0 = tn * TN, k0 = tk * kTileK;
    const int64_t m_begin = (int64_t)slice * g.rows_per_slice;
    int64_t m_end = m_begin + g.rows_per_slice;
    if (m_end > rows) m_end = rows;

    // staging: X tile: thread -> (row tid/16 + 16*pass, chunk tid%16); dY tile: (row tid/YCH + (256/YCH)*pass, chunk tid%YCH)
    const int xr = tid >> 4, xc = tid & 15;
    const int yr = tid / YCH, yc = tid % YCH;
    const bool ycol_ok = n0 + yc * 8 < n_out, xcol_ok = k0 + xc * 8 < k_in;  // n_out, k_in are multiples of 8
    const uint16_t* yptr = dy + n0 + yc * 8;
    const uint16_t* xptr = x + k0 + xc * 8;
    const bool do_bias = part_b != nullptr && tk == 0;
    float bsum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    f32x16 acc[NB][2];
#pragma unroll
    for (int a = 0; a < NB; ++a)
#pragma unroll
        for (int b2 = 0; b2 < 2; ++b2)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b2][r] = 0.f;

    uint4 ry[YPASS], rx[2];
    auto load_regs = [&](int64_t m0) {
#pragma unroll
        for (int ps = 0; ps < YPASS; ++ps) {
            const int64_t m = m0 + yr + (256 / YCH) * ps;
            ry[ps] = (m < m_end && ycol_ok) ? *(const uint4*)(yptr + m * n_out) : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) {
            const int64_t m = m0 + xr + 16 * ps;
            rx[ps] = (m < m_end && xcol_ok) ? *(const uint4*)(xptr + m * k_in) : make_uint4(0, 0, 0, 0);
        }
    };
    auto store_lds = [&](int buf) {
#pragma unroll
        for (int ps = 0; ps < YPASS; ++ps) {
            *(uint4*)(ytile(buf) + (yr + (256 / YCH) * ps) * LDY + yc * 16) = ry[ps];
            if (do_bias) {
                const uint32_t w[4] = {ry[ps].x, ry[ps].y, ry[ps].z, ry[ps].w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    bsum[2 * i] += __uint_as_float(w[i] << 16);
                    bsum[2 * i + 1] += __uint_as_float(w[i] & 0xffff0000u);
                }
            }
        }
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) *(uint4*)(xtile(buf) + (xr + 16 * ps) * LDX + xc * 16) = rx[ps];
    };

    if (m_begin < m_end) {
        load_regs(m_begin);
        store_lds(0);
    }
    __syncthreads();
    int buf = 0;
    for (int64_t m0 = m_begin; m0 < m_end; m0 += kTok) {
        const bool more = m0 + kTok < m_end;
        if (more) load_regs(m0 + kTok);  // in flight during the MFMAs below
        // all fragment reads of the stage first: the second half's LDS latency hides under the first half's MFMAs
        bf16x8 af[2][NB], bf[2][2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int tok0 = ks * 16 + 8 * half;
#pragma unroll
            for (int j = 0; j < 2; ++j) bf[ks][j] = tr_frag<LDX>(xtile(buf), tok0, wk * 64 + j * 32, lane);
#pragma unroll
            for (int i = 0; i < NB; ++i) af[ks][i] = tr_frag<LDY>(ytile(buf), tok0, wn * 32 * NB + i * 32, lane);
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < NB; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks][i], bf[ks][j], acc[i][j], 0, 0, 0);
        if (more) store_lds(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }

    // partial tile -> workspace [slice][n_out][k_in]; accumulator: column = k (lane & 31), rows = n
    float* dst = part_w + (int64_t)slice * ((int64_t)n_out * k_in + n_out);
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int kk = k0 + wk * 64 + j * 32 + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int nn = n0 + wn * 32 * NB + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (nn < n_out && kk < k_in) dst[(int64_t)nn * k_in + kk] = acc[i][j][r];
            }
        }
    if (do_bias) {  // fold the row groups that share a column chunk (tile buffers are free: last loop barrier passed)
#pragma unroll
        for (int i = 0; i < 8; ++i) bred[yr * TN + yc * 8 + i] = bsum[i];
        __syncthreads();
        if (tid < TN) {
            float t = 0.f;
#pragma unroll
            for (int rg = 0; rg < 256 / YCH; ++rg) t += bred[rg * TN + tid];
            if (n0 + tid < n_out) part_b[(int64_t)slice * ((int64_t)n_out * k_in + n_out) + n0 + tid] = t;
        }
    }
}

// LDS-DMA variant of the kernel above: the token tiles go global -> LDS directly (buffer_load_dwordx4 ... lds), three
// stages deep, so two stages (48 KB per workgroup) are in flight while one is multiplied: the register-staged kernel
// can only keep ONE stage in flight and is bound by the L2/HBM round trip of that single prefetch, not by MFMA or LDS.
//   * LDS image: unpadded row-major tiles; the DMA writes wave-uniform base + lane * 16 B, so the bank skew comes from
//     an XOR on the 16-byte chunk index (chunk ^ ((row & 3) << 2)), applied to the per-lane SOURCE address on the way in
//     and to the transposing reads on the way out;
//   * buffer descriptors spanning exactly the slice's rows: token rows past the end read as zeros (ragged last stage);
//     column overrun of a partial tile reads the next row / zeros and only ever reaches accumulators that are not stored;
//   * per stage: s_waitcnt vmcnt(own loads of the NEXT stage) -> s_barrier -> issue stage t+2 -> fragments + MFMA.
//     The barrier both publishes stage t and retires every wave's reads of the buffer that stage t+2 overwrites.
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

// one MFMA operand (8 tokens x 1 column per lane) by two transposing reads issued from inline asm; a = LDS byte address of
// the lane's first 4-row group, ks = 16-token half of the stage
template <int RB>
__device__ __forceinline__ s16x8 tr_frag_asm(uint32_t a, int ks) {
    s16x4 lo, hi;
    if (ks == 0) {
        asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo) : "v"(a));
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(a), "n"(4 * RB));
    } else {
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo) : "v"(a), "n"(16 * RB));
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(a), "n"(20 * RB));
    }
    return s16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

// GX: the X operand is gelu(X) -- the weight gradient of the Linear BEHIND a GELU taken from the saved pre-activation (fc2 of the fused
// Mlp block, csrc/mlp_fused.hip, which then does not write gelu(h) at all).  The activation is applied to the MFMA fragments
// between their LDS read and the product: 16 packed evaluations per stage and wave, which an HBM-bound launch (C <= 128: the only
// shapes that use it) has the VALU slots for.
template <int NB, int TK, bool GX = false>
__global__ void __launch_bounds__(TK * 2, 2) wgrad_dma_kernel(const uint16_t* __restrict__ dy, const uint16_t* __restrict__ x,
                                                           float* __restrict__ part_w, float* __restrict__ part_b,
                                                           int64_t rows, int n_out, int k_in, Geometry g, int ldy, int ldx, int yc0, int xc0) {
    // ldy / ldx: row strides of dY / X in elements; yc0 / xc0: first column of the operand inside its row (the operands may be
    // column blocks of wider matrices -- the hi / lo parts of a bf16x3 split, ops.split3).  The descriptors cover whole rows of the
    // WIDE matrices, so a tile that overhangs the operand's columns reads its neighbours (never stored), not unmapped memory
#if defined(__HIP_DEVICE_COMPILE__)  // buffer-resource / LDS-DMA builtins exist in the device pass only
    constexpr int TN = 64 * NB;
    constexpr int WK = TK / 64, NW = 2 * WK, NT = 64 * NW;  // waves along k, waves, threads: waves in 2 x WK, each (32*NB) x 64
    constexpr int YRB = TN * 2, XRB = TK * 2;    // bytes of one staged dY / X tile row
    constexpr int YB = kTok * YRB, XB = kTok * XRB;  // bytes of one staged dY / X tile
    // the DMA runs NSTAGE - 1 stages ahead: what bounds the main loop is the bytes in flight per CU, so the
    // 256-wide tile (one workgroup per CU) takes a fourth 32 KB buffer: 96 instead of 64 KB in flight
    constexpr int STAGE = YB + XB, NSTAGE = TK == 256 ? 4 : 3, AH = NSTAGE - 1;
    constexpr int YI = YB / 1024 / NW, XI = XB / 1024 / NW;  // 1-KB DMA instructions per wave and stage
    constexpr int YCH = YRB / 16, XCH = XRB / 16;           // 16-byte chunks per dY / X tile row
    __shared__ __attribute__((aligned(16))) unsigned char smem[NSTAGE * STAGE];
    float* bred = (float*)smem;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5;
    const int wn = wave / WK, wk = wave % WK;
    int slice, tile;
    if (!block_to_work(g, blockIdx.x, slice, tile)) return;
    const int tn = tile / g.tiles_k, tk = tile % g.tiles_k;
    const int n0 = tn * TN, k0 = tk * TK;
    const int64_t m_begin = (int64_t)slice * g.rows_per_slice;
    int64_t m_end = m_begin + g.rows_per_slice;
    if (m_end > rows) m_end = rows;
    const int m_len = m_end > m_begin ? (int)(m_end - m_begin) : 0;
    const int nst = (m_len + kTok - 1) / kTok;
    const bool do_bias = part_b != nullptr && tk == 0;

    // descriptors over [m_begin, m_end) x the full row; raw (stride 0) buffers return 0 past num_records
    const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc((void*)(dy + m_begin * ldy), 0, m_len * ldy * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)(x + m_begin * ldx), 0, m_len * ldx * 2, 0x00020000);
    // DMA instruction q of a tile fills LDS bytes [q KB, q KB + 1 KB): position p = 64 q + lane -> (row, physical chunk)
    int voff_y[YI], voff_x[XI];
#pragma unroll
    for (int j = 0; j < YI; ++j) {
        const int p = (wave * YI + j) * 64 + lane, row = p / YCH, pc = p % YCH;
        voff_y[j] = row * ldy * 2 + (yc0 + n0) * 2 + ((pc ^ ((row & 3) << 2)) << 4);
    }
#pragma unroll
    for (int j = 0; j < XI; ++j) {
        const int p = (wave * XI + j) * 64 + lane, row = p / XCH, pc = p % XCH;
        voff_x[j] = row * ldx * 2 + (xc0 + k0) * 2 + ((pc ^ ((row & 3) << 2)) << 4);
    }
    const int ystep = kTok * ldy * 2, xstep = kTok * ldx * 2;
    auto issue = [&](int b) {
        unsigned char* base = smem + b * STAGE;
#pragma unroll
        for (int j = 0; j < YI; ++j) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_y, (lds_void*)(base + (wave * YI + j) * 1024), 16, voff_y[j], 0, 0, 0);
            voff_y[j] += ystep;
        }
#pragma unroll
        for (int j = 0; j < XI; ++j) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lds_void*)(base + YB + (wave * XI + j) * 1024), 16, voff_x[j], 0, 0, 0);
            voff_x[j] += xstep;
        }
    };

    // transposing fragment reads: lane -> (row L/4 of a 4-row group, 4 columns) ; swizzled chunk = chunk ^ (row_in << 2)
    const int L = lane & 15, nblk = (lane >> 4) & 1, row_in = L >> 2;
    const int lane_lo = (nblk * 2 + ((L & 3) >> 1)) * 16 + (L & 1) * 8;
    int yoff[NB], xoff[2];
#pragma unroll
    for (int i = 0; i < NB; ++i) yoff[i] = (8 * half + row_in) * YRB + ((((wn * NB + i) ^ row_in) << 2) << 4) + lane_lo;
#pragma unroll
    for (int j = 0; j < 2; ++j) xoff[j] = YB + (8 * half + row_in) * XRB + ((((wk * 2 + j) ^ row_in) << 2) << 4) + lane_lo;
    // The transposing reads go through inline asm: behind the ds_read_tr16 builtin hipcc (ROCm 7.2) drains every LDS-DMA in
    // flight (s_waitcnt vmcnt(0)) before the first read of a stage, which serialises the pipeline.  The compiler does not
    // count asm reads, so their completion is awaited by the explicit lgkmcnt waits below (LDS operations return in order).
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
    // bias partial sums: thread -> (row group tid / YCH, logical chunk tid % YCH)
    const int yr = tid / YCH, yc = tid % YCH;
    const int boff = yr * YRB + ((yc ^ ((yr & 3) << 2)) << 4);  // rows yr + (NT / YCH) * ps share (row & 3)
    float bsum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    f32x16 acc[NB][2];
#pragma unroll
    for (int a = 0; a < NB; ++a)
#pragma unroll
        for (int b2 = 0; b2 < 2; ++b2)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b2][r] = 0.f;

    // one stage; the buffer indices are compile-time constants (loop unrolled by NSTAGE) so that the compiler can tell the
    // DMA's destination buffer from the one being read: with run-time indices it orders them with an s_waitcnt vmcnt(0)
    s16x8 af[2][NB], bf[2][2];  // the stage's MFMA operands (the second wave group keeps them across the barrier, see below)
    auto read_frags = [&](auto buf_c) {
        constexpr int buf = decltype(buf_c)::value;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int j = 0; j < 2; ++j) bf[ks][j] = tr_frag_asm<XRB>(lds0 + buf * STAGE + xoff[j], ks);
#pragma unroll
            for (int i = 0; i < NB; ++i) af[ks][i] = tr_frag_asm<YRB>(lds0 + buf * STAGE + yoff[i], ks);
        }
    };
    auto wait_half = [&](int ks) {  // ks = 0: first half landed once at most the second half's 2 * (NB + 2) reads are outstanding
        if constexpr (NB == 4) {
            if (ks == 0) asm volatile("s_waitcnt lgkmcnt(12)" : "+v"(bf[0][0]), "+v"(bf[0][1]), "+v"(af[0][0]), "+v"(af[0][1]), "+v"(af[0][2]), "+v"(af[0][3]));
            else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bf[1][0]), "+v"(bf[1][1]), "+v"(af[1][0]), "+v"(af[1][1]), "+v"(af[1][2]), "+v"(af[1][3]));
        } else {
            if (ks == 0) asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(bf[0][0]), "+v"(bf[0][1]), "+v"(af[0][0]), "+v"(af[0][1]));
            else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bf[1][0]), "+v"(bf[1][1]), "+v"(af[1][0]), "+v"(af[1][1]));
        }
    };
    auto mma_half = [&](int ks) {
        if constexpr (GX) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                u32x4 w = __builtin_bit_cast(u32x4, bf[ks][j]);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const f32x2 v = gelu2(f32x2{__uint_as_float(w[e] << 16), __uint_as_float(w[e] & 0xffff0000u)});
                    w[e] = pack_bf16x2(v.x, v.y);
                }
                bf[ks][j] = __builtin_bit_cast(s16x8, w);
            }
        }
#pragma unroll
        for (int i = 0; i < NB; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[ks][i]), __builtin_bit_cast(bf16x8, bf[ks][j]), acc[i][j], 0, 0, 0);
    };
    auto bias_sums = [&](auto buf_c) {  // column sums of the staged dY tile (also asm reads: a plain LDS load would drain the DMA queue)
        constexpr int buf = decltype(buf_c)::value;
        constexpr int PS = kTok / (NT / YCH);
        u32x4 v[PS];
        const uint32_t ba = lds0 + buf * STAGE + boff;
        asm volatile("ds_read_b128 %0, %1" : "=v"(v[0]) : "v"(ba));
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v[1]) : "v"(ba), "n"((NT / YCH) * YRB));
        if constexpr (PS == 4) {
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v[2]) : "v"(ba), "n"(2 * (NT / YCH) * YRB));
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v[3]) : "v"(ba), "n"(3 * (NT / YCH) * YRB));
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]));
        } else {
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[0]), "+v"(v[1]));
        }
#pragma unroll
        for (int ps = 0; ps < PS; ++ps) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                bsum[2 * i] += __uint_as_float(v[ps][i] << 16);
                bsum[2 * i + 1] += __uint_as_float(v[ps][i] & 0xffff0000u);
            }
        }
    };
    // ALTERNATING WAVE GROUPS (8-wave tile): the workgroup barrier of every stage phase-locks the two waves of a SIMD -- both
    // read their fragments, both wait for the LDS, both then want the matrix pipe, both idle at the next barrier -- so the pipe
    // sits idle for a whole barrier + DMA issue + LDS round trip per stage (measured: MFMA busy 49 %, ~2070 cycles per stage
    // for 1024 cycles of MFMA work per SIMD).  Waves 4..7 (the second wave of each SIMD) therefore run HALF A STAGE OUT OF PHASE:
    // behind barrier t they first multiply the fragments of stage t - 1, which they fetched before the barrier, and only then
    // fetch stage t's; waves 0..3 fetch first and multiply second.  One group always has MFMAs to issue while the other waits.
    const bool late_group = NW == 8 && g.reads_first == 3 && wave >= NW / 2;
    auto stage = [&](int t, auto buf_c, auto nbuf_c) {
        constexpr int nbuf = decltype(nbuf_c)::value;
        // stage t has landed once only the younger stages still issued (at most AH - 1, fewer at the tail) are outstanding
        const int younger = nst - 1 - t;
        if (AH - 1 >= 2 && younger >= 2)
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (YI + XI)) : "memory");
        else if (younger >= 1)
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(YI + XI) : "memory");
        else
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (late_group) {
            if (t > 0) {
                mma_half(0);
                mma_half(1);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (t + AH < nst) issue(nbuf);
            read_frags(buf_c);
            if (do_bias) bias_sums(buf_c);
            wait_half(0);
            wait_half(1);  // complete before this wave arrives at the next barrier (which releases the buffer for a refill)
            return;
        }
        if (g.reads_first == 0 && t + AH < nst) issue(nbuf);
        read_frags(buf_c);
        // the next stage's DMA issued BEHIND this stage's fragment reads: its ~100 issue cycles per piece then cover the LDS
        // latency of the reads instead of standing in front of them (1.5-2 % over issuing it first)
        if (g.reads_first != 0 && g.reads_first != 2 && t + AH < nst) issue(nbuf);
        wait_half(0);
        mma_half(0);
        __builtin_amdgcn_sched_barrier(0);  // keep the first half's MFMAs ahead of the second wait
        if (g.reads_first == 2 && t + AH < nst) issue(nbuf);  // (A/B) in the shadow of the first half's MFMAs
        wait_half(1);
        mma_half(1);
        if (do_bias) bias_sums(buf_c);
    };
    if (nst > 0) issue(0);
    if (nst > 1) issue(1);
    if (AH > 2 && nst > 2) issue(2);
    using std::integral_constant;
    for (int t = 0; t < nst; t += NSTAGE) {  // stage t computes buffer t % NSTAGE and refills the one stage t - 1 consumed
        if constexpr (NSTAGE == 3) {
            stage(t, integral_constant<int, 0>{}, integral_constant<int, 2>{});
            if (t + 1 < nst) stage(t + 1, integral_constant<int, 1>{}, integral_constant<int, 0>{});
            if (t + 2 < nst) stage(t + 2, integral_constant<int, 2>{}, integral_constant<int, 1>{});
        } else {
            stage(t, integral_constant<int, 0>{}, integral_constant<int, 3>{});
            if (t + 1 < nst) stage(t + 1, integral_constant<int, 1>{}, integral_constant<int, 0>{});
            if (t + 2 < nst) stage(t + 2, integral_constant<int, 2>{}, integral_constant<int, 1>{});
            if (t + 3 < nst) stage(t + 3, integral_constant<int, 3>{}, integral_constant<int, 2>{});
        }
    }

    if (late_group && nst > 0) {  // the last stage's fragments are still to be multiplied
        mma_half(0);
        mma_half(1);
    }

    float* dst = part_w + (int64_t)slice * ((int64_t)n_out * k_in + n_out);
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int kk = k0 + wk * 64 + j * 32 + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int nn = n0 + wn * 32 * NB + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (nn < n_out && kk < k_in) dst[(int64_t)nn * k_in + kk] = acc[i][j][r];
            }
        }
    if (do_bias) {
        __syncthreads();  // every wave is done with the stage buffers
#pragma unroll
        for (int i = 0; i < 8; ++i) bred[yr * TN + yc * 8 + i] = bsum[i];
        __syncthreads();
        if (tid < TN) {
            float t = 0.f;
#pragma unroll
            for (int rg = 0; rg < NT / YCH; ++rg) t += bred[rg * TN + tid];
            if (n0 + tid < n_out) part_b[(int64_t)slice * ((int64_t)n_out * k_in + n_out) + n0 + tid] = t;
        }
    }
#endif
}

// ------------------------------------------------------------------------------------------------ fp32 activations
// The reference's precision = 32 runs: the same split-token scheme on v_mfma_f32_32x32x2_f32.  The instruction contracts
// TWO tokens (lane half h supplies token 2j + h) and each lane needs one element of one column per operand -- so the
// row-major token tiles are read as they lie (lanes = consecutive columns: conflict-free ds_read_b32; no transposing reads,
// no swizzle) and the LDS-DMA image is the plain tile.  128 x 128 tiles, 4 waves of 64 x 64, 16-token stages of 16 KB,
// three stages, three workgroups per CU; MFMA-bound (the fp32 MFMA peak is 1/16 of bf16's).  hipBLASLt ran these
// reductions at 10-40 TFLOP/s on the token-heavy stages.
constexpr int kTokF = 16;

template <int OFF>
__device__ __forceinline__ float lds_b32(uint32_t a) {
    float v;
    asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(a), "n"(OFF));
    return v;
}
template <int OFF>
__device__ __forceinline__ u32x4 lds_b128(uint32_t a) {
    u32x4 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(a), "n"(OFF));
    return v;
}
template <typename F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    static_for_impl(f, std::make_integer_sequence<int, N>{});
}

__global__ void __launch_bounds__(256, 3) wgrad_dma_f32_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                               float* __restrict__ part_w, float* __restrict__ part_b, int64_t rows,
                                                               int n_out, int k_in, Geometry g) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int TN = 128, TK = 128;
    constexpr int RB = TN * 4;                       // bytes of one staged tile row (both tiles are 128 floats wide)
    constexpr int YB = kTokF * RB, XB = kTokF * RB;  // 8 KB each
    // the DMA runs NSTAGE - 1 stages ahead: what bounds the main loop is the bytes in flight per CU, so the
    // 256-wide tile (one workgroup per CU) takes a fourth 32 KB buffer: 96 instead of 64 KB in flight
    constexpr int STAGE = YB + XB, NSTAGE = TK == 256 ? 4 : 3, AH = NSTAGE - 1;
    constexpr int YI = YB / 1024 / 4, XI = XB / 1024 / 4;  // 1-KB DMA instructions per wave and stage (2 + 2)
    __shared__ __attribute__((aligned(16))) unsigned char smem[NSTAGE * STAGE];
    float* bred = (float*)smem;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const int wn = wave >> 1, wk = wave & 1;
    int slice, tile;
    if (!block_to_work(g, blockIdx.x, slice, tile)) return;
    const int tn = tile / g.tiles_k, tk = tile % g.tiles_k;
    const int n0 = tn * TN, k0 = tk * TK;
    const int64_t m_begin = (int64_t)slice * g.rows_per_slice;
    int64_t m_end = m_begin + g.rows_per_slice;
    if (m_end > rows) m_end = rows;
    const int m_len = m_end > m_begin ? (int)(m_end - m_begin) : 0;
    const int nst = (m_len + kTokF - 1) / kTokF;
    const bool do_bias = part_b != nullptr && tk == 0;

    const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc((void*)(dy + m_begin * n_out), 0, m_len * n_out * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)(x + m_begin * k_in), 0, m_len * k_in * 4, 0x00020000);
    int voff_y[YI], voff_x[XI];
#pragma unroll
    for (int j = 0; j < YI; ++j) {
        const int p = (wave * YI + j) * 64 + lane, row = p >> 5, ch = p & 31;  // 32 16-byte chunks per 512-byte row
        voff_y[j] = row * n_out * 4 + n0 * 4 + (ch << 4);
    }
#pragma unroll
    for (int j = 0; j < XI; ++j) {
        const int p = (wave * XI + j) * 64 + lane, row = p >> 5, ch = p & 31;
        voff_x[j] = row * k_in * 4 + k0 * 4 + (ch << 4);
    }
    const int ystep = kTokF * n_out * 4, xstep = kTokF * k_in * 4;
    auto issue = [&](int b) {
        unsigned char* base = smem + b * STAGE;
#pragma unroll
        for (int j = 0; j < YI; ++j) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_y, (lds_void*)(base + (wave * YI + j) * 1024), 16, voff_y[j], 0, 0, 0);
            voff_y[j] += ystep;
        }
#pragma unroll
        for (int j = 0; j < XI; ++j) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lds_void*)(base + YB + (wave * XI + j) * 1024), 16, voff_x[j], 0, 0, 0);
            voff_x[j] += xstep;
        }
    };
    // LDS reads through inline asm (a compiler-visible LDS load beside the DMA queue is preceded by s_waitcnt vmcnt(0))
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
    const uint32_t ya = lds0 + half * RB + (wn * 64 + l31) * 4;       // dY: token 2j + half, column wn*64 + a*32 + l31
    const uint32_t xa = lds0 + YB + half * RB + (wk * 64 + l31) * 4;  // X : token 2j + half, column wk*64 + b*32 + l31
    const uint32_t ba = lds0 + (tid >> 5) * RB + (tid & 31) * 16;     // bias sums: thread -> (row of an 8-row group, 4 columns)
    float bsum[4] = {0.f, 0.f, 0.f, 0.f};

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b2 = 0; b2 < 2; ++b2)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b2][r] = 0.f;

    auto stage = [&](int t, auto buf_c, auto nbuf_c) {
        constexpr int buf = decltype(buf_c)::value, nbuf = decltype(nbuf_c)::value;
        // stage t has landed once only the younger stages still issued (at most AH - 1, fewer at the tail) are outstanding
        const int younger = nst - 1 - t;
        if (AH - 1 >= 2 && younger >= 2)
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (YI + XI)) : "memory");
        else if (younger >= 1)
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(YI + XI) : "memory");
        else
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (t + AH < nst) issue(nbuf);
        const uint32_t yb = ya + buf * STAGE, xb = xa + buf * STAGE;
        float fa[8][2], fb[8][2];  // [token pair][32-column block]
        static_for<8>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            fa[j][0] = lds_b32<2 * j * RB>(yb);
            fa[j][1] = lds_b32<2 * j * RB + 128>(yb);
            fb[j][0] = lds_b32<2 * j * RB>(xb);
            fb[j][1] = lds_b32<2 * j * RB + 128>(xb);
        });
        asm volatile("s_waitcnt lgkmcnt(15)"  // LDS operations return in order: pairs 0..3 (16 reads) have landed
                     : "+v"(fa[0][0]), "+v"(fa[0][1]), "+v"(fb[0][0]), "+v"(fb[0][1]), "+v"(fa[1][0]), "+v"(fa[1][1]), "+v"(fb[1][0]),
                       "+v"(fb[1][1]), "+v"(fa[2][0]), "+v"(fa[2][1]), "+v"(fb[2][0]), "+v"(fb[2][1]), "+v"(fa[3][0]), "+v"(fa[3][1]),
                       "+v"(fb[3][0]), "+v"(fb[3][1]));
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b2 = 0; b2 < 2; ++b2)
                    acc[a][b2] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[j][a], fb[j][b2], acc[a][b2], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(fa[4][0]), "+v"(fa[4][1]), "+v"(fb[4][0]), "+v"(fb[4][1]), "+v"(fa[5][0]), "+v"(fa[5][1]), "+v"(fb[5][0]),
                       "+v"(fb[5][1]), "+v"(fa[6][0]), "+v"(fa[6][1]), "+v"(fb[6][0]), "+v"(fb[6][1]), "+v"(fa[7][0]), "+v"(fa[7][1]),
                       "+v"(fb[7][0]), "+v"(fb[7][1]));
#pragma unroll
        for (int j = 4; j < 8; ++j)
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b2 = 0; b2 < 2; ++b2)
                    acc[a][b2] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[j][a], fb[j][b2], acc[a][b2], 0, 0, 0);
        if (do_bias) {  // column sums of the staged dY tile: rows (tid >> 5) and (tid >> 5) + 8
            u32x4 v0 = lds_b128<0>(ba + buf * STAGE), v1 = lds_b128<8 * RB>(ba + buf * STAGE);
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v0), "+v"(v1));
#pragma unroll
            for (int i = 0; i < 4; ++i) bsum[i] += __uint_as_float(v0[i]) + __uint_as_float(v1[i]);
        }
    };
    if (nst > 0) issue(0);
    if (nst > 1) issue(1);
    if (AH > 2 && nst > 2) issue(2);
    using std::integral_constant;
    for (int t = 0; t < nst; t += NSTAGE) {  // stage t computes buffer t % NSTAGE and refills the one stage t - 1 consumed
        if constexpr (NSTAGE == 3) {
            stage(t, integral_constant<int, 0>{}, integral_constant<int, 2>{});
            if (t + 1 < nst) stage(t + 1, integral_constant<int, 1>{}, integral_constant<int, 0>{});
            if (t + 2 < nst) stage(t + 2, integral_constant<int, 2>{}, integral_constant<int, 1>{});
        } else {
            stage(t, integral_constant<int, 0>{}, integral_constant<int, 3>{});
            if (t + 1 < nst) stage(t + 1, integral_constant<int, 1>{}, integral_constant<int, 0>{});
            if (t + 2 < nst) stage(t + 2, integral_constant<int, 2>{}, integral_constant<int, 1>{});
            if (t + 3 < nst) stage(t + 3, integral_constant<int, 3>{}, integral_constant<int, 2>{});
        }
    }

    float* dst = part_w + (int64_t)slice * ((int64_t)n_out * k_in + n_out);
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b2 = 0; b2 < 2; ++b2) {
            const int kk = k0 + wk * 64 + b2 * 32 + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int nn = n0 + wn * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (nn < n_out && kk < k_in) dst[(int64_t)nn * k_in + kk] = acc[a][b2][r];
            }
        }
    if (do_bias) {
        __syncthreads();  // every wave is done with the stage buffers
        const int rg = tid >> 5, c4 = (tid & 31) * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) bred[rg * TN + c4 + i] = bsum[i];
        __syncthreads();
        if (tid < TN) {
            float t = 0.f;
#pragma unroll
            for (int r2 = 0; r2 < 8; ++r2) t += bred[r2 * TN + tid];
            if (n0 + tid < n_out) part_b[(int64_t)slice * ((int64_t)n_out * k_in + n_out) + n0 + tid] = t;
        }
    }
#endif
}

}  // namespace
}  // namespace hs

extern "C" {

int64_t hs_linear_wgrad_workspace(int64_t rows, int n_out, int k_in) {
    if (rows <= 0 || n_out <= 0 || k_in <= 0) return 0;
    // the larger of the bf16 and fp32 geometries (the call does not know the dtype)
    const hs::Geometry g = hs::make_geometry(rows, n_out, k_in), f = hs::make_geometry_f32(rows, n_out, k_in);
    const int64_t a = g.slices + (g.chunks > 1 ? g.chunks : 0), b = f.slices + (f.chunks > 1 ? f.chunks : 0);
    return (a > b ? a : b) * ((int64_t)n_out * k_in + n_out);
}

namespace {
int linear_wgrad_impl(const void* dy, const void* x, float* dw, float* dbias, float* workspace, int64_t rows, int n_out, int k_in,
                      int accumulate, int dtype, void* stream, int ldy, int ldx, int yc0, int xc0, bool gelu_x = false);
}

int hs_linear_wgrad(const void* dy, const void* x, float* dw, float* dbias, float* workspace, int64_t rows, int n_out,
                    int k_in, int accumulate, int dtype, void* stream) {
    return linear_wgrad_impl(dy, x, dw, dbias, workspace, rows, n_out, k_in, accumulate, dtype, stream, n_out, k_in, 0, 0);
}

int hs_linear_wgrad_gelu_supported(int64_t rows, int n_out, int k_in, int dtype) {
    if (dtype != HS_BF16 || rows <= 0 || n_out <= 0 || k_in <= 0 || n_out % 4 || k_in % 8) return 0;
    const hs::Geometry g = hs::make_geometry(rows, n_out, k_in);
    return g.dma && g.tile_k != 256 && g.tile_n != 256;
}

int hs_linear_wgrad_gelu(const void* dy, const void* h, float* dw, float* dbias, float* workspace, int64_t rows, int n_out, int k_in,
                         int accumulate, int dtype, void* stream) {
    return linear_wgrad_impl(dy, h, dw, dbias, workspace, rows, n_out, k_in, accumulate, dtype, stream, n_out, k_in, 0, 0, true);
}

int hs_linear_wgrad_ld(const void* dy, int64_t ldy, int64_t ycol0, const void* x, int64_t ldx, int64_t xcol0, float* dw, float* dbias,
                       float* workspace, int64_t rows, int n_out, int k_in, int accumulate, void* stream) {
    HS_CHECK_ARG(ycol0 >= 0 && xcol0 >= 0 && ycol0 + n_out <= ldy && xcol0 + k_in <= ldx && ldy % 8 == 0 && ldx % 8 == 0 &&
                 ycol0 % 8 == 0 && xcol0 % 8 == 0 && ldy < (1 << 20) && ldx < (1 << 20),
                 "hs_linear_wgrad_ld: column blocks must lie inside the rows; strides and offsets are multiples of 8 elements");
    return linear_wgrad_impl(dy, x, dw, dbias, workspace, rows, n_out, k_in, accumulate, HS_BF16, stream, (int)ldy, (int)ldx, (int)ycol0,
                             (int)xcol0);
}

namespace {
int linear_wgrad_impl(const void* dy, const void* x, float* dw, float* dbias, float* workspace, int64_t rows, int n_out, int k_in,
                      int accumulate, int dtype, void* stream, int ldy, int ldx, int yc0, int xc0, bool gelu_x) {
    using namespace hs;
    HS_CHECK_ARG(dy && x && dw && workspace, "null pointer");
    HS_CHECK_ARG(rows > 0 && n_out > 0 && k_in > 0, "bad shape");
    HS_CHECK_ARG(dtype == HS_BF16 || dtype == HS_F32, "dtype must be HS_F32 or HS_BF16");
    // bf16: k_in: 16-byte X rows.  n_out: multiples of 8, or of 4 on the LDS-DMA path (its dword-aligned buffer loads read a
    // narrow dY row -- the 12-class segmentation head -- together with its successors; the surplus columns land in
    // accumulators that are never stored).  fp32: multiples of 4 (16-byte rows); LDS-DMA path only.
    if (dtype == HS_BF16 && (n_out % 4 || k_in % 8))
        return fail(HS_ERR_UNSUPPORTED, "bf16: n_out must be a multiple of 4 and k_in a multiple of 8");
    if (dtype == HS_F32 && (n_out % 4 || k_in % 4)) return fail(HS_ERR_UNSUPPORTED, "fp32: n_out and k_in must be multiples of 4");
    Geometry g = dtype == HS_F32 ? make_geometry_f32(rows, n_out, k_in) : make_geometry(rows, n_out, k_in);
    if (gelu_x && !(dtype == HS_BF16 && g.dma && g.tile_k != 256 && g.tile_n != 256))
        return fail(HS_ERR_UNSUPPORTED, "hs_linear_wgrad_gelu: bf16 and the 128 x 128 LDS-DMA tile only (n_out <= 128-class shapes)");
    const bool strided = ldy != n_out || ldx != k_in || yc0 || xc0;
    if (strided) {  // column blocks of wider matrices: LDS-DMA kernels only, 32-bit offsets inside a token slice
        if (!g.dma || g.rows_per_slice * (int64_t)(ldy > ldx ? ldy : ldx) * 2 >= ((int64_t)1 << 31))
            return fail(HS_ERR_UNSUPPORTED, "hs_linear_wgrad_ld: a token slice exceeds the 2 GiB buffer-offset range");
    }
    if (dtype == HS_BF16 && n_out % 8 && !g.dma) return fail(HS_ERR_UNSUPPORTED, "n_out must be a multiple of 8 for slices beyond 2 GiB");
    if (dtype == HS_F32 && !g.dma) return fail(HS_ERR_UNSUPPORTED, "fp32: a token slice exceeds the 2 GiB buffer-offset range");
    const int64_t n = (int64_t)n_out * k_in, rec = n + n_out;
    float* part_w = workspace;
    float* part_b = dbias ? workspace + n : nullptr;  // bias partials live behind each slice's weight partial
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((unsigned)(8 * g.per_xcd));
    const uint16_t* dyp = (const uint16_t*)dy;
    const uint16_t* xp = (const uint16_t*)x;
    if (dtype == HS_F32)
        hipLaunchKernelGGL(wgrad_dma_f32_kernel, grid, dim3(256), 0, s, (const float*)dy, (const float*)x, part_w, part_b, rows, n_out,
                           k_in, g);
    else if (g.dma && g.tile_k == 256)
        hipLaunchKernelGGL((wgrad_dma_kernel<4, 256>), grid, dim3(512), 0, s, dyp, xp, part_w, part_b, rows, n_out, k_in, g, ldy, ldx, yc0, xc0);
    else if (g.dma && g.tile_n == 256)
        hipLaunchKernelGGL((wgrad_dma_kernel<4, 128>), grid, dim3(256), 0, s, dyp, xp, part_w, part_b, rows, n_out, k_in, g, ldy, ldx, yc0, xc0);
    else if (g.dma && gelu_x)
        hipLaunchKernelGGL((wgrad_dma_kernel<2, 128, true>), grid, dim3(256), 0, s, dyp, xp, part_w, part_b, rows, n_out, k_in, g, ldy, ldx, yc0, xc0);
    else if (g.dma)
        hipLaunchKernelGGL((wgrad_dma_kernel<2, 128>), grid, dim3(256), 0, s, dyp, xp, part_w, part_b, rows, n_out, k_in, g, ldy, ldx, yc0, xc0);
    else if (g.tile_n == 256)
        hipLaunchKernelGGL(wgrad_kernel<4>, grid, dim3(256), 0, s, dyp, xp, part_w, part_b, rows, n_out, k_in, g);
    else
        hipLaunchKernelGGL(wgrad_kernel<2>, grid, dim3(256), 0, s, dyp, xp, part_w, part_b, rows, n_out, k_in, g);
    HS_LAUNCH_CHECK("linear_wgrad");
    const int64_t count = dbias ? rec : n;  // without a bias the tail of each record is never written nor read
    if (accumulate & HS_ACC_DEFER)  // the slice sum joins the stream's queue of deferred reductions (csrc/reduce_many.hip)
        return reduce_defer(part_w, rec, g.slices, n, count, dw, dbias, accumulate & 1, s);
    return reduce_now(part_w, rec, g.slices, n, count, dw, dbias, accumulate & 1, s);
}
}  // namespace

}  // extern "C"
