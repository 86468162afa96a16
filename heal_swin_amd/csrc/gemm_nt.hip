// bf16 "NT" GEMM with fused epilogues for the forward and input-gradient products of the path's Linear layers:
//
//     C[m, n] = sum_k A[m, k] * B[n, k]   (+ sum_k A2[m, k] * B2[n, k])   -> epilogue -> bf16
//
// A = activations (token rows, k contiguous), B = a weight as nn.Linear stores it ([out, in], k contiguous); the input
// gradient uses the transposed bf16 weight copy, so it is the same product.  The optional second (A2, B2) segment is the
// decoder's skip connection cat([x, skip]) W^T without the concatenation (swin_hp_transformer.py:772-775).
// Epilogues (fp32 on the accumulators, before the single rounding to bf16):
//     EPI_BIAS   c = acc + bias                                                    (qkv, proj, fc2, reduction, expand, head)
//     EPI_GELU   h = acc + bias -> c ;  aux = dropout(gelu(h))                      (fc1 -> act -> drop, ref :39-41)
//     EPI_DGELU  c = acc * mask * gelu'(aux)                                        (input gradient of fc2 through the GELU)
//     EPI_RESID  c = acc + bias + aux                                               (residual gradient folded into a dgrad)
// so the standalone GELU passes over the 4C-wide hidden tensor (13.5 % of the round-1 step) do not exist.
//
// Structure (gfx950):
//   * workgroup tile BM x BN x 64, waves in WM x WN, each wave (BM/WM) x (BN/WN) as 32x32 accumulators of
//     v_mfma_f32_32x32x16_bf16.  The product is formed TRANSPOSED (D = Bfrag * Afrag^T: accumulator column = m, rows = n),
//     so a lane owns 4 consecutive n of one output row: 8-byte row-major stores, bias as a float4;
//   * operands go global -> LDS by buffer_load ... lds (16 B per lane, 1 KB per wave-instruction), two stages, the loads of
//     step s+1 in flight under the MFMAs of step s; the stream of steps runs ACROSS output tiles (persistent workgroups):
//     the first loads of the next tile are in flight under the epilogue of the current one;
//   * LDS image: tile rows of 128 B paired into 256-B super-rows, 16-byte chunk index XORed with the super-row number
//     (applied to the per-lane SOURCE address of the DMA and to the fragment reads): the 16-lane groups of ds_read_b128
//     touch 16 distinct 16-byte slots (conflict-free), and every 128-B global line is still fetched by 8 adjacent lanes;
//   * tile ids are dealt to XCDs in contiguous ranges (block b runs on XCD b % 8), consecutive ids share the A row panel,
//     so concurrently running workgroups of an XCD re-read A from that XCD's L2;
//   * edges: rows beyond M / N fall outside the buffer descriptors (reads return 0, stores are dropped); a K tail and the
//     n >= N columns are predicated per lane by pointing the access outside the descriptor.
// 128 x 128 tiles run two workgroups per CU (64 KB LDS each): one's epilogue (VALU: erf GELU) overlaps the other's MFMAs.
#include <cstdlib>
#include <type_traits>

#include "hs_gelu.h"

namespace hs {
HS_DEFINE_SEED_EPOCH_SETTER(set_seed_epoch_gemm_nt)
namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef unsigned int u32x2 __attribute__((__vector_size__(8)));    // the types the b64 / b128 buffer builtins take
typedef unsigned int u32x4v __attribute__((__vector_size__(16)));
typedef __attribute__((address_space(3))) void lds_void;

// HS_GEMM_EPI_RING (A/B: 0 = register loads, one HBM round trip per row block behind a full vmcnt(0)): the epilogue's INPUT rows
// (h for GELU', the residual) of the 256 x 256 tile go global -> LDS by DMA through two patches per wave inside the consumed stage
// buffer, requested two row blocks ahead and waited for by COUNT
#ifndef HS_GEMM_EPI_RING
#define HS_GEMM_EPI_RING 1
#endif
// HS_GEMM_EXP (measurement builds only, tools/gemm_overlap_premise.sh): bit 1 = the second half of the waves stores one output
// tile's worth of bytes (into `aux`, which the bias epilogue does not use) from INSIDE the k-steps, a 1-KB instruction per wave and
// 16-deep sub-step: the skeleton of an epilogue whose stores ride under the next tile's MFMAs (profiles/archive_r01_r04/r03_gemm_overlap_premise.txt:
// it costs 65-75 % of what the same bytes cost in a serial epilogue)
// HS_GEMM_STORE_AUX: cache policy of the epilogue's row-segment stores (gfx950: bit 0 = sc0, bit 1 = nt, bit 4 = sc1).  Default 2 = NON-TEMPORAL:
// an output tile is written once and read by another kernel; stored with the default policy its lines are allocated in the XCD's L2
// (4 MB for 32 CUs x 128-256 KB of output per tile) and push out the A panel / weight lines the neighbouring workgroups re-read.  Same box,
// s2 fc1 (us): bias 239.7 / 238.7 -> 225.1 / 228.4, + GELU 302.7 / 306.0 -> 289.2 / 289.5, GELU' 318.0 / 317.1 -> 315.8 / 311.7; whole step
// 147.2 -> 144.6 ms (HEAL-SWIN-B), 45.9 -> 44.7 ms (paper T @ 256); sc1 (write-through) alone: nothing (profiles/r06_gemm_store_policy.txt)
#ifndef HS_GEMM_STORE_AUX
#define HS_GEMM_STORE_AUX 2
#endif
// HS_GEMM_IN_AUX: cache policy of the epilogue's INPUT rows (h for GELU', the residual), read once through the DMA ring.  Default 2 = non-temporal,
// for the reason above: GELU' s2 292.0 / 293.5 -> 282.8 / 286.4 us, s1 426 / 423 -> 414 / 415 (profiles/r06_gemm_store_policy.txt)
#ifndef HS_GEMM_IN_AUX
#define HS_GEMM_IN_AUX 2
#endif
#ifndef HS_GEMM_EXP
#define HS_GEMM_EXP 0
#endif
#ifndef HS_GEMM_STAGGER_CYCLES
#define HS_GEMM_STAGGER_CYCLES 40000  // one 256 x 256 x 512 tile with a GELU epilogue (tools/gemm_trace.py)
#endif
enum { EPI_BIAS = 0, EPI_GELU = 1, EPI_DGELU = 2, EPI_RESID = 3 };
__device__ constexpr uint32_t kOob = 0x7FFFFF00u;  // a byte offset outside every descriptor below
constexpr int64_t kMaxRecords = 0x7FFFFE00;  // descriptors are clamped to this many bytes (tiles address < 2 GiB from their origin)

struct GemmParams {
    const uint16_t *a, *b;
    int64_t lda, ldb;
    int k;
    const uint16_t *a2, *b2;
    int64_t lda2, ldb2;
    int k2;
    const float* bias;
    uint16_t* c;
    uint16_t* aux;
    int64_t m;
    int n;
    int tiles_n, tiles, per_xcd, blocks_per_xcd;
    float drop_p;
    uint64_t seed;
#ifdef HS_GEMM_TRACE
    uint64_t* trace;  // measurement build: (shader clock << 8 | event code) per wave of workgroups 0 and 9, kTraceCap entries each
#endif
};
#ifdef HS_GEMM_TRACE
constexpr int kTraceCap = 240;
#endif

// Vector-memory instructions younger than the input request of row block i at the moment the epilogue waits for it.  Issue
// order: D_0, D_1, then per block b its 4 stores S_b and the 4 DMA instructions of D_{b+2} (loads and stores retire in order).
constexpr int epi_younger_ops(int tm, int i) {
    const int stores = i < 2 ? i : 1;
    const int last_d = i + 1 < tm - 1 ? i + 1 : tm - 1;
    return 4 * (stores + (last_d - i));
}

template <int OFF>
__device__ __forceinline__ u32x4 lds_read_b128(uint32_t addr) {
    u32x4 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}

// NSTAGE LDS stage buffers (the DMA runs NSTAGE - 1 k-steps ahead of the MFMAs); ALIAS: the epilogue's per-wave patches lie
// inside the stage buffer that was just consumed (one extra barrier per tile) instead of in LDS of their own
// FAST (k and k2 multiples of 64: no K tail): the operand DMA of a k-step is issued by the FIRST HALF of the waves alone (twice the
// pieces each, their k offset in the scalar operand), the other half runs MFMAs and fragment reads only -- 8-16 % on the 256 x 256
// tile at every shape (profiles/archive_r01_r04/r03_gemm_role_split.txt); the addressing form alone changes nothing
template <int BM, int BN, int WM, int WN, int NSTAGE, bool ALIAS, int EPI, bool DROP, bool FAST>
__global__ void __launch_bounds__(WM * WN * 64, 2) gemm_nt_kernel(GemmParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int NW = WM * WN;
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;  // 32x32 accumulators per wave along m / n
    constexpr int AB = BM * 128, BB = BN * 128, STAGE = AB + BB;
    constexpr int IW = FAST ? NW / 2 : NW;                   // waves that issue the operand DMA
    constexpr int AI = AB / 1024 / IW, BI = BB / 1024 / IW;  // 1-KB DMA instructions per issuing wave and stage
    static_assert(AI >= 1 && BI >= 1 && TM >= 1 && TN >= 1, "tile too small for the wave grid");
    static_assert(!ALIAS || NW * 4096 <= STAGE, "patches do not fit a stage buffer");
    constexpr bool HAS_IN = EPI == EPI_DGELU || EPI == EPI_RESID;
    constexpr bool RING = HS_GEMM_EPI_RING && HAS_IN && ALIAS && TM == 4 && TN == 2 && STAGE >= NW * 8192;
#ifdef HS_GEMM_TRACE
    // (the three-stage 256 x 128 tile has no room for the event buffers beside the bias slots: not traced)
    constexpr int TRACE_LDS = NSTAGE * STAGE + (ALIAS ? 0 : NW * 4096) + 2048 + NW * kTraceCap * 8 <= 163840 ? NW * kTraceCap * 8 : 0;
#else
    constexpr int TRACE_LDS = 0;
#endif
    // LBIAS: the tile's BN bias values travel global -> LDS by one DMA piece of wave 0 when the ISSUE cursor enters the tile (two 1-KB
    // slots by tile parity behind the stages) and the epilogue reads them with ds_read_b128.  A global load at the top of the
    // epilogue is the youngest entry of the wave's in-order vmcnt queue: on the DMA-issuing waves it waited for the whole first
    // k-step of the NEXT tile (64 KB from L2 / HBM) before row block 0 could start -- 3700 cycles per tile on the waves every
    // barrier then waits for (profiles/archive_r01_r04/r03_gemm_pass_overlap.txt: row block 0 8582 cycles on wave 0, 4878 on wave 7).
    // (Not on the 128 x 128 tile: its two workgroups per CU use all 160 KB already.)
    constexpr bool LBIAS = ALIAS && EPI != EPI_DGELU;
    constexpr int BIAS_OFF = NSTAGE * STAGE + (ALIAS ? 0 : NW * 4096), BIAS_LDS = LBIAS ? 2048 : 0;
    __shared__ __attribute__((aligned(16))) unsigned char smem[BIAS_OFF + BIAS_LDS + TRACE_LDS];  // stages (+ epilogue patches) (+ bias slots)

    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;

    // ---- this workgroup's tiles: XCD x owns ids [x * per_xcd, (x + 1) * per_xcd), dealt round-robin to its workgroups
    const int xcd = blockIdx.x & 7, lb = blockIdx.x >> 3;
    const int id_end = min((xcd + 1) * p.per_xcd, p.tiles);
    const int id0 = xcd * p.per_xcd + lb;
    if (id0 >= id_end) return;
    const int nk1 = (p.k + 63) >> 6, nk2 = (p.k2 + 63) >> 6, nk = nk1 + nk2;
#if HS_GEMM_EXP & 24
    {  // measurement build: start phases of the workgroups of an XCD spread over one tile period (bit 3: two phases, bit 4: four),
       // so that the epilogues (store bursts) of neighbouring CUs do not coincide
        const int phases = (HS_GEMM_EXP & 16) ? 4 : 2;
        const long long wait = (long long)(lb % phases) * (HS_GEMM_STAGGER_CYCLES / phases);
        const long long t0 = clock64();
        while (clock64() - t0 < wait) __builtin_amdgcn_s_sleep(8);
    }
#endif

    // ---- DMA lane mapping: LDS position q (16-B units inside a tile) = (super-row R = q / 16, physical chunk q % 16);
    // logical chunk = physical ^ (R & 15); tile row = 2 R + (logical >> 3), 16-byte column chunk = logical & 7
    // (FAST keeps FOUR offsets per operand whatever the piece count: pieces j and j + 4 of a wave differ by 32 tile rows and
    // nothing else -- the swizzle term depends on j % 4 only -- and that part rides in the scalar offset together with k)
    constexpr bool SC = FAST;  // scalar-offset form of the DMA addresses
    static_assert(!FAST || (AI % 4 == 0 && BI % 4 == 0), "FAST: whole groups of four pieces per issuing wave");
    constexpr int AJ = SC ? 4 : AI, BJ = SC ? 4 : BI;
    int a_row[AJ], a_col[AJ], b_row[BJ], b_col[BJ];
#pragma unroll
    for (int j = 0; j < AJ; ++j) {
        const int q = (wave * AI + j) * 64 + lane, R = q >> 4, lc = (q & 15) ^ (R & 15);
        a_row[j] = 2 * R + (lc >> 3);
        a_col[j] = (lc & 7) << 4;
    }
#pragma unroll
    for (int j = 0; j < BJ; ++j) {
        const int q = (wave * BI + j) * 64 + lane, R = q >> 4, lc = (q & 15) ^ (R & 15);
        b_row[j] = 2 * R + (lc >> 3);
        b_col[j] = (lc & 7) << 4;
    }

    // ---- issue side: the (tile, K segment) the DMA currently reads from.  Descriptors, row offsets and the row length are
    // rebuilt only when the issue cursor enters a new tile or segment, not per k-step.
    __amdgpu_buffer_rsrc_t ra, rb;
    int a_off[AJ], b_off[BJ];  // byte offset of this lane's (row, chunk) inside the tile, k-step 0
    int kseg = 0;              // bytes of a row of the current segment
    int a_ld32 = 0, b_ld32 = 0;  // (FAST) bytes of 32 operand rows
    int par_i = 0;               // (LBIAS) slot the next tile's bias goes to
    __amdgpu_buffer_rsrc_t rbias = __builtin_amdgcn_make_buffer_rsrc((void*)p.bias, 0, p.bias ? p.n * 4 : 0, 0x00020000);
    auto retarget = [&](int id, bool s2, bool enter) {
        const int tm = id / p.tiles_n, tn = id - tm * p.tiles_n;
        const int64_t m0 = (int64_t)tm * BM;
        const int n0 = tn * BN;
        if constexpr (LBIAS) {
            if (enter) {
                // lanes beyond the tile's BN floats (and columns beyond n) read outside the descriptor: zeros
                if (wave == 0)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rbias, (lds_void*)(smem + BIAS_OFF + par_i * 1024), 16,
                                                             lane * 4 < BN ? (uint32_t)(lane * 16) : kOob, n0 * 4, 0, 0);
                par_i ^= 1;
            }
        }
        const uint16_t* ap = s2 ? p.a2 : p.a;
        const uint16_t* bp = s2 ? p.b2 : p.b;
        const int64_t lda = s2 ? p.lda2 : p.lda, ldb = s2 ? p.ldb2 : p.ldb;
        kseg = (s2 ? p.k2 : p.k) * 2;
        int64_t abytes = (p.m - m0 - 1) * lda * 2 + kseg, bbytes = (int64_t)(p.n - n0 - 1) * ldb * 2 + kseg;
        abytes = abytes > kMaxRecords ? kMaxRecords : abytes;
        bbytes = bbytes > kMaxRecords ? kMaxRecords : bbytes;
        ra = __builtin_amdgcn_make_buffer_rsrc((void*)(ap + m0 * lda), 0, (int)abytes, 0x00020000);
        rb = __builtin_amdgcn_make_buffer_rsrc((void*)(bp + (int64_t)n0 * ldb), 0, (int)bbytes, 0x00020000);
#pragma unroll
        for (int j = 0; j < AJ; ++j) a_off[j] = a_row[j] * ((int)lda * 2) + a_col[j];
#pragma unroll
        for (int j = 0; j < BJ; ++j) b_off[j] = b_row[j] * ((int)ldb * 2) + b_col[j];
        a_ld32 = (int)lda * 64;
        b_ld32 = (int)ldb * 64;
    };
    // DMA pieces [first, first + count) of the AI + BI pieces of one k-step (byte offset kb inside the row) into stage `buf`
    auto issue_pieces = [&](int kb, int buf, auto first_c, auto count_c) {
        constexpr int first = decltype(first_c)::value, count = decltype(count_c)::value;
        if (IW < NW && wave >= IW) return;
        unsigned char* base = smem + buf * STAGE;
#pragma unroll
        for (int q = first; q < first + count; ++q) {
            if constexpr (SC) {
            // (k % 64 == 0: no K-tail predicate; the k offset and the 32-row group ride in the scalar operand, which takes
            // part in the descriptor's range check like the vector part)
            if (q < AI) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_void*)(base + (wave * AI + q) * 1024), 16, (uint32_t)a_off[q % AJ],
                                                         kb + (q >> 2) * a_ld32, 0, 0);
            } else {
                const int j = q - AI;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_void*)(base + AB + (wave * BI + j) * 1024), 16, (uint32_t)b_off[j % BJ],
                                                         kb + (j >> 2) * b_ld32, 0, 0);
            }
            } else {

            if (q < AI) {
                const uint32_t voff = kb + a_col[q] < kseg ? (uint32_t)(a_off[q] + kb) : kOob;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_void*)(base + (wave * AI + q) * 1024), 16, voff, 0, 0, 0);
            } else {
                const int j = q - AI;
                const uint32_t voff = kb + b_col[j] < kseg ? (uint32_t)(b_off[j] + kb) : kOob;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_void*)(base + AB + (wave * BI + j) * 1024), 16, voff, 0, 0, 0);
            }
            }
        }
    };

    // ---- fragment addresses: operand row r (tile row), lane half h supplies k-chunk 2 ksub + h of the 16-deep MFMA step:
    // byte = R * 256 + ((((r & 1) * 8 + 2 ksub + h) ^ (R & 15)) << 4) = frag_base ^ (ksub << 5)
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
#ifdef HS_GEMM_TRACE
    const bool tr_on = TRACE_LDS > 0 && p.trace && (blockIdx.x == 0 || blockIdx.x == 9);
    int tr_n = 0;
    const uint32_t tr_base = lds0 + BIAS_OFF + BIAS_LDS + wave * kTraceCap * 8;
    auto TR = [&](int code) {
        if (tr_on && tr_n < kTraceCap) {
            const uint64_t t = (clock64() << 8) | (uint64_t)code;
            if (lane == 0) asm volatile("ds_write_b64 %0, %1" ::"v"(tr_base + tr_n * 8), "v"(t) : "memory");
            ++tr_n;
        }
    };
#else
    auto TR = [&](int) {};
#endif
    uint32_t a_frag[TM], b_frag[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int r = wm * (BM / WM) + i * 32 + l31, R = r >> 1;
        a_frag[i] = lds0 + R * 256 + (((((r & 1) << 3) + half) ^ (R & 15)) << 4);
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int r = wn * (BN / WN) + j * 32 + l31, R = r >> 1;
        b_frag[j] = lds0 + AB + R * 256 + (((((r & 1) << 3) + half) ^ (R & 15)) << 4);
    }

    f32x16 acc[TN][TM];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][i][r] = 0.f;
#if HS_GEMM_EXP & 2
    // the store stream of the experiment: this workgroup's share of `aux`, 1 KB per instruction
    int64_t exp_total = p.m * (int64_t)p.n * 2;
    exp_total = exp_total > kMaxRecords ? kMaxRecords : exp_total;
    const __amdgpu_buffer_rsrc_t rexp = __builtin_amdgcn_make_buffer_rsrc((void*)p.aux, 0, (int)(p.aux ? exp_total : 0), 0x00020000);
    const uint32_t exp_share = (uint32_t)((exp_total / gridDim.x) & ~(int64_t)1023);
    uint32_t exp_pos = 0;  // bytes of the share written so far (wraps)
    const uint32_t exp_base = blockIdx.x * exp_share;
#endif

    // ---- one k-step out of stage buffer `buf`: four 16-deep MFMA sub-steps, their fragment reads streamed ahead of them, and
    // the DMA pieces of a later k-step (into stage buffer `buf_next`, free since the barrier) issued behind each MFMA group
    auto compute = [&](int buf, bool prefetch, int kb_next, int buf_next) {
        const uint32_t bo = buf * STAGE;
        // fragment reads are streamed one or two 16-deep sub-steps ahead of their MFMAs through a ring of NSET register sets
        // (two for the 128 x 64 wave tile, whose 128 accumulator registers leave no room for a third)
        constexpr int NSET = (TM == 4 && TN == 2) ? 2 : 3;
        u32x4 fa[NSET][TM], fb[NSET][TN];
        auto read_frags = [&](auto ks_c) {
            constexpr int ks = decltype(ks_c)::value, set = ks % NSET;
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[set][j] = lds_read_b128<0>((b_frag[j] ^ (ks << 5)) + bo);
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[set][i] = lds_read_b128<0>((a_frag[i] ^ (ks << 5)) + bo);
        };
        constexpr int PIECES = AI + BI;
        constexpr int PER = TM + TN;
        auto wait_frags = [&](auto ks_c, auto left_c) {
            constexpr int ks = decltype(ks_c)::value, left = decltype(left_c)::value, set = ks % NSET;
            (void)fa;  // (named outside the if-constexpr branches: clang otherwise refuses the implicit capture)
            (void)fb;
            // reads of sub-step ks have landed once at most `left` later reads are outstanding (LDS returns in order)
            if constexpr (TM == 2 && TN == 2)
                asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(fa[set][0]), "+v"(fa[set][1]), "+v"(fb[set][0]), "+v"(fb[set][1]) : "n"(left));
            else if constexpr (TM == 4 && TN == 2)
                asm volatile("s_waitcnt lgkmcnt(%6)"
                             : "+v"(fa[set][0]), "+v"(fa[set][1]), "+v"(fa[set][2]), "+v"(fa[set][3]), "+v"(fb[set][0]), "+v"(fb[set][1])
                             : "n"(left));
            else
                static_assert(TM == 2 || TM == 4, "add a wait form for this wave tile");
        };
        // MFMAs of sub-step ks, then DMA pieces [first, first + cnt) of a later step in their shadow
        auto mma = [&](auto ks_c, auto first_c, auto cnt_c) {
            constexpr int ks = decltype(ks_c)::value, set = ks % NSET;
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                        acc[j][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fb[set][j]),
                                                                            __builtin_bit_cast(bf16x8, fa[set][i]), acc[j][i], 0, 0, 0);
                }
            if (decltype(cnt_c)::value > 0 && prefetch) issue_pieces(kb_next, buf_next, first_c, cnt_c);
#if HS_GEMM_EXP & 2
            if (wave >= NW / 2) {
                const uint32_t off = exp_pos + (uint32_t)(wave - NW / 2) * 1024u;
                const uint32_t so = off + 1024u <= exp_share ? exp_base + off : kOob;
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4v, fa[set][0]), rexp, lane * 16, so, 0);
            }
            exp_pos += (NW / 2) * 1024u;
            if (exp_pos >= exp_share) exp_pos = 0;
#endif
            __builtin_amdgcn_sched_barrier(0);
        };
        using I0 = std::integral_constant<int, 0>;
        using I1 = std::integral_constant<int, 1>;
        using I2 = std::integral_constant<int, 2>;
        using I3 = std::integral_constant<int, 3>;
        using IPER = std::integral_constant<int, PER>;
        read_frags(I0{});
        read_frags(I1{});
        constexpr int PQ = (PIECES + 3) / 4;  // a quarter of the next free buffer's pieces behind each MFMA group
        constexpr int c3 = PIECES - 3 * PQ > 0 ? PIECES - 3 * PQ : 0;
        wait_frags(I0{}, IPER{});
        mma(I0{}, I0{}, std::integral_constant<int, PQ>{});
        read_frags(I2{});
        wait_frags(I1{}, IPER{});
        mma(I1{}, std::integral_constant<int, PQ>{}, std::integral_constant<int, PQ>{});
        read_frags(I3{});  // re-uses the set of a group whose MFMAs have all been issued
        wait_frags(I2{}, IPER{});
        mma(I2{}, std::integral_constant<int, 2 * PQ>{}, std::integral_constant<int, (3 * PQ <= PIECES ? PQ : PIECES - 2 * PQ)>{});
        wait_frags(I3{}, I0{});
        mma(I3{}, std::integral_constant<int, 3 * PQ>{}, std::integral_constant<int, c3>{});
    };

    // ---- epilogue of tile `id`.  A lane owns output row m = l31 of each 32-row block and 4 consecutive n per register group.
    // Row-per-lane 8-byte global stores touch 32 cache lines per instruction and are issue-bound (16 of them per lane cost
    // thousands of cycles), so each wave passes its [32 rows][64 columns] block through a private 4 KB LDS patch
    // (16-byte chunk ^ (row & 7): the 8-byte writes are 2-way, the 16-byte reads conflict-free) and stores / loads WHOLE
    // 128-byte row segments: 4 dwordx4 instructions per block, 8 rows each.  All patch accesses are inline asm: a
    // compiler-visible LDS access beside the DMA queue would be preceded by s_waitcnt vmcnt(0) and stall the epilogue
    // behind the next tile's first loads.  Needs n % 8 == 0; otherwise (the 12-class head) the direct 8-byte form is used.
    const uint32_t own_rel = wave * 4096 + l31 * 128 + 8 * half;     // + ((chunk ^ (l31 & 7)) << 4), chunk = 4 j + g
    // OUTPUT blocks are written with the two 8-byte halves of every 16-byte chunk exchanged in rows 8-15 and 24-31: rows r and r + 8
    // of a 16-lane ds_write_b64 group share chunk slot and half otherwise (2-way conflict on every patch write: 1540 instead of
    // 770 LDS cycles per row block and CU); the row view undoes it for free -- its row is lane / 8 + 8 t, so bit 3 is t & 1 and the
    // odd t swap register halves
    const uint32_t own_out_rel = wave * 4096 + l31 * 128 + 8 * (half ^ ((l31 >> 3) & 1));
    const uint32_t row_rel = wave * 4096 + (lane >> 3) * 128 + (((lane & 7) ^ (lane >> 3)) << 4);  // + t * 1024: row lane/8 + 8 t
    auto epilogue = [&](int id, int buf_done, int par_c) {
        const uint32_t patch_off = ALIAS ? buf_done * STAGE : NSTAGE * STAGE;  // byte offset of the patch area inside smem
        TR(10);
        if (ALIAS) __builtin_amdgcn_s_barrier();  // every wave is done reading the stage buffer the patches live in
#ifdef HS_GEMM_WAVE_STAGGER
        // measurement build: the waves enter the epilogue HS_GEMM_WAVE_STAGGER x 64 cycles apart, so that their VALU / LDS-write / store phases
        // (which they otherwise all run at the same moment on CU-wide resources) interleave
        for (int w = 0; w < wave; ++w) __builtin_amdgcn_s_sleep(HS_GEMM_WAVE_STAGGER);
#endif
        TR(11);
        const int tm = id / p.tiles_n, tn = id - tm * p.tiles_n;
        const int64_t m0 = (int64_t)tm * BM;
        const int n0 = tn * BN;
        const int n2 = p.n * 2;
        int64_t cbytes = (p.m - m0) * (int64_t)n2;
        cbytes = cbytes > kMaxRecords ? kMaxRecords : cbytes;
        const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc((void*)(p.c + m0 * p.n), 0, (int)(p.c ? cbytes : 0), 0x00020000);
        const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)(p.aux + m0 * p.n), 0, (int)(p.aux ? cbytes : 0), 0x00020000);
        const ElemRng rng(p.drop_p, p.seed);
        const bool wide = RING || ((TN & 1) == 0 && (p.n & 7) == 0);  // whole-row-segment path
        // a wave's BN / WN columns are handled 64 at a time (one patch pass per half jh: wave tiles of 64 or 128 columns)
        constexpr int JH = TN / 2 > 0 ? TN / 2 : 1, TJ = TN < 2 ? TN : 2;
#pragma unroll
        for (int jh = 0; jh < JH; ++jh) {
        const int ncol0 = n0 + wn * (BN / WN) + jh * 64;  // first of the 64 columns of this pass
        float4 bias4[TJ][4];
        // (LBIAS: a column block's four float4 are read from the LDS slot where they are used -- 4 broadcast reads per block instead of
        // 32 registers held across the pass, which the RESID epilogue of the 128 x 64 wave tile does not have)
        const uint32_t baddr = lds0 + BIAS_OFF + par_c * 1024 + (wn * (BN / WN) + jh * 64 + 4 * half) * 4;
        auto lds_bias = [&](int j, float4 (&b)[4]) {
            u32x4 bq[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) asm volatile("ds_read_b128 %0, %1" : "=v"(bq[g]) : "v"(baddr + (j * 32 + 8 * g) * 4));
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bq[0]), "+v"(bq[1]), "+v"(bq[2]), "+v"(bq[3]));
#pragma unroll
            for (int g = 0; g < 4; ++g) b[g] = __builtin_bit_cast(float4, bq[g]);
        };
        if constexpr (LBIAS) {
        } else {
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = ncol0 + j * 32 + 4 * half + 8 * g;
                bias4[j][g] = (EPI != EPI_DGELU && p.bias && n < p.n) ? *(const float4*)(p.bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        // row-lane view of a block: lane -> (row lane/8 + 8 t, 16-byte chunk lane % 8).  The per-lane part of the byte offset is ONE
        // register; the row block / row group part is wave-uniform and rides in the instruction's scalar offset (it takes part
        // in the descriptor's range check like the vector part).  Per-(i, t) vector offsets were hoisted out of the tile loop
        // by the compiler and spilled: every reload inside the store sequence is a `s_waitcnt vmcnt(0)`, i.e. a full drain of
        // the stores issued so far -- two of them cost the bias epilogue 5000 cycles per tile (tools/gemm_trace.py)
        const int rl_col = ncol0 + (lane & 7) * 8;
        const uint32_t rl_voff = rl_col < p.n ? (uint32_t)((lane >> 3) * n2 + rl_col * 2) : kOob;
        auto row_soff = [&](int i, int t) -> int { return (wm * (BM / WM) + i * 32 + 8 * t) * n2; };
        // input ring: row block i passes through patch slot i % 2 (slot s of wave w at patch_off + s * NW * 4096 + w * 4096).  A DMA
        // lane writes LDS position lane * 16 of its 1-KB piece = (row lane / 8, physical chunk lane % 8), so it FETCHES logical
        // chunk (lane % 8) ^ (lane / 8) of that row: the image the own-lane reads expect, every 128-byte row segment still one
        // request of 8 adjacent lanes.  (Launched only with n % 8 == 0: hs_gemm_nt.)
        const int in_col = ncol0 + (((lane & 7) ^ (lane >> 3)) << 3);
        const uint32_t in_voff = in_col < p.n ? (uint32_t)((lane >> 3) * n2 + in_col * 2) : kOob;
        auto request_in = [&](int i) {
            unsigned char* base = smem + patch_off + (i & 1) * (NW * 4096) + wave * 4096;
#pragma unroll
            for (int t = 0; t < 4; ++t)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_void*)(base + t * 1024), 16, in_voff, row_soff(i, t), 0, HS_GEMM_IN_AUX);
        };
        if constexpr (RING) {
            request_in(0);
            request_in(1);
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int ml = wm * (BM / WM) + i * 32 + l31;
            // ---- input block (h or the residual) in the own-lane view
            const uint32_t pbase = lds0 + patch_off + (RING ? (i & 1) * (NW * 4096) : 0);
            const uint32_t own_addr = pbase + own_rel, row_addr = pbase + row_rel;
            const uint32_t own_out = lds0 + patch_off + (RING ? (i & 1) * (NW * 4096) : 0) + own_out_rel;
            u32x2 xin[TJ][4];
            if (HAS_IN) {
                if (wide) {
                    if constexpr (RING) {
                        // this block's rows have landed once only the younger operations are outstanding
                        constexpr int Y0 = epi_younger_ops(TM, 0), Y1 = epi_younger_ops(TM, 1), Y2 = epi_younger_ops(TM, 2),
                                      Y3 = epi_younger_ops(TM, 3);
                        if (i == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(Y0) : "memory");
                        else if (i == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(Y1) : "memory");
                        else if (i == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(Y2) : "memory");
                        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(Y3) : "memory");
                    } else {
                        u32x4 rws[4];
#pragma unroll
                        for (int t = 0; t < 4; ++t)
                            rws[t] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, rl_voff, row_soff(i, t), 0));
                        asm volatile("ds_write_b128 %0, %1" ::"v"(row_addr), "v"(rws[0]) : "memory");
                        asm volatile("ds_write_b128 %0, %1 offset:1024" ::"v"(row_addr), "v"(rws[1]) : "memory");
                        asm volatile("ds_write_b128 %0, %1 offset:2048" ::"v"(row_addr), "v"(rws[2]) : "memory");
                        asm volatile("ds_write_b128 %0, %1 offset:3072" ::"v"(row_addr), "v"(rws[3]) : "memory");
                    }
#pragma unroll
                    for (int j = 0; j < TJ; ++j)
#pragma unroll
                        for (int g = 0; g < 4; ++g)
                            asm volatile("ds_read_b64 %0, %1" : "=v"(xin[j][g]) : "v"(own_addr + (((4 * j + g) ^ (l31 & 7)) << 4)));
                    asm volatile("s_waitcnt lgkmcnt(0)"
                                 : "+v"(xin[0][0]), "+v"(xin[0][1]), "+v"(xin[0][2]), "+v"(xin[0][3]), "+v"(xin[TJ - 1][0]),
                                   "+v"(xin[TJ - 1][1]), "+v"(xin[TJ - 1][2]), "+v"(xin[TJ - 1][3]));
                } else {
#pragma unroll
                    for (int j = 0; j < TJ; ++j)
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const int n = ncol0 + j * 32 + 4 * half + 8 * g;
                            xin[j][g] = __builtin_amdgcn_raw_buffer_load_b64(rx, n < p.n ? (uint32_t)(ml * n2 + n * 2) : kOob, 0, 0);
                        }
                }
            }
            // ---- arithmetic on the accumulators, two elements at a time (packed fp32, hs_gelu.h).  Dropout (DROP) is a
            // kernel instantiation of its own: next to the branch-free p = 0 arithmetic its counter hashes cost the
            // 128 x 64 wave tile ~70 spilled registers, some of them inside the main loop
            u32x2 o1[TJ][4], o2[TJ][4];  // o1 -> c ; o2 -> aux (EPI_GELU only)
            {
#pragma unroll
                for (int j = 0; j < TJ; ++j) {
                    if constexpr (LBIAS) lds_bias(j, bias4[j]);
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int n = ncol0 + j * 32 + 4 * half + 8 * g;
                        f32x16& a16 = acc[TJ * jh + j][i];
                        f32x2 v[2] = {f32x2{a16[4 * g], a16[4 * g + 1]} + f32x2{bias4[j][g].x, bias4[j][g].y},
                                      f32x2{a16[4 * g + 2], a16[4 * g + 3]} + f32x2{bias4[j][g].z, bias4[j][g].w}};
                        // dropout: the lane's four consecutive elements are the first (half = 0) or second half of a chunk of the
                        // generator (hs_device.h): one chunk key, one multiply per element pair, formed where the pair is used
                        uint32_t ck = 0;
                        if (DROP) ck = rng.chunk_key((uint64_t)((m0 + ml) * p.n + n) >> 3);
                        auto drop2 = [&](int t) {
                            const uint32_t hh = ElemRng::pair_bits(ck, t == 0 ? (half ? ElemRng::kM2 : ElemRng::kM0) : (half ? ElemRng::kM3 : ElemRng::kM1));
                            return f32x2{rng.keep_lo(hh), rng.keep_hi(hh)};
                        };
                        if (EPI == EPI_GELU) {
                            // h is packed here and kept (fp32) in the accumulator registers; the activation is computed from them
                            // by make_o2 below, at the place in the block's sequence that the wave's role asks for
                            o1[j][g] = u32x2{pack_bf16x2(v[0].x, v[0].y), pack_bf16x2(v[1].x, v[1].y)};
                            a16[4 * g] = v[0].x; a16[4 * g + 1] = v[0].y; a16[4 * g + 2] = v[1].x; a16[4 * g + 3] = v[1].y;
                            continue;
                        } else {
                            if (EPI == EPI_DGELU || EPI == EPI_RESID) {
#pragma unroll
                                for (int t = 0; t < 2; ++t) {
                                    const f32x2 x = {__uint_as_float(xin[j][g][t] << 16), __uint_as_float(xin[j][g][t] & 0xffff0000u)};
                                    if (EPI == EPI_DGELU) {
                                        v[t] *= gelu_grad2(x);
                                        if (DROP) v[t] *= drop2(t);
                                    } else {
                                        v[t] += x;
                                    }
                                }
                            }
                            o1[j][g] = u32x2{pack_bf16x2(v[0].x, v[0].y), pack_bf16x2(v[1].x, v[1].y)};
                        }
#pragma unroll
                        for (int r = 0; r < 4; ++r) a16[4 * g + r] = 0.f;
                    }
                }
            }
            auto make_o2 = [&]() {  // (EPI_GELU) aux = dropout(gelu(h)); clears the accumulators.  Four element pairs in lock step.
#pragma unroll
                for (int j = 0; j < TJ; ++j)
#pragma unroll
                    for (int g0 = 0; g0 < 4; g0 += 2) {
                        f32x16& a16 = acc[TJ * jh + j][i];
                        f32x2 v[4] = {f32x2{a16[4 * g0], a16[4 * g0 + 1]}, f32x2{a16[4 * g0 + 2], a16[4 * g0 + 3]},
                                      f32x2{a16[4 * g0 + 4], a16[4 * g0 + 5]}, f32x2{a16[4 * g0 + 6], a16[4 * g0 + 7]}};
#if !(HS_GEMM_EXP & 4)  // (measurement build: bit 2 = no activation arithmetic, aux = h -- the floor of a two-output epilogue)
                        // (four pairs in lock step, gelu2_n<4>, measured: no faster -- the block is bound by its patch round trips
                        // and store issue, not by the chains' latency; profiles/r06_gemm_epilogue_experiments.txt)
#pragma unroll
                        for (int t = 0; t < 4; ++t) v[t] = gelu2(v[t]);
#endif
#pragma unroll
                        for (int gg = 0; gg < 2; ++gg) {
                            const int g = g0 + gg;
                            if (DROP) {
                                const int n = ncol0 + j * 32 + 4 * half + 8 * g;
                                const uint32_t ck = rng.chunk_key((uint64_t)((m0 + ml) * p.n + n) >> 3);
#pragma unroll
                                for (int t = 0; t < 2; ++t) {
                                    const uint32_t hh = ElemRng::pair_bits(ck, t == 0 ? (half ? ElemRng::kM2 : ElemRng::kM0) : (half ? ElemRng::kM3 : ElemRng::kM1));
                                    v[2 * gg + t] *= f32x2{rng.keep_lo(hh), rng.keep_hi(hh)};
                                }
                            }
                            o2[j][g] = u32x2{pack_bf16x2(v[2 * gg].x, v[2 * gg].y), pack_bf16x2(v[2 * gg + 1].x, v[2 * gg + 1].y)};
#pragma unroll
                            for (int r = 0; r < 4; ++r) a16[4 * g + r] = 0.f;
                        }
                    }
            };
            TR(20);  // arithmetic of the block done
            // ---- outputs
            auto emit = [&](const __amdgpu_buffer_rsrc_t& rs, const u32x2 (&o)[TJ][4]) {
                if (wide) {
#pragma unroll
                    for (int j = 0; j < TJ; ++j)
#pragma unroll
                        for (int g = 0; g < 4; ++g)
                            asm volatile("ds_write_b64 %0, %1" ::"v"(own_out + (((4 * j + g) ^ (l31 & 7)) << 4)), "v"(o[j][g]) : "memory");
                    u32x4 rws[4];
                    // (rows 8-15 / 24-31 of the block = odd t: the chunk halves lie exchanged, own_out_rel -- ds_read2_b64 with the
                    // two 8-byte offsets in reverse order returns them in place.  NOT a register swap after the read: hipcc put the
                    // four v_mov straight behind the buffer_store_dwordx4 of the previous row group, over its data registers, and
                    // lanes 12-15 of every 16 stored the NEW first dword (no hazard is listed for a store with an SGPR offset,
                    // none is inserted; seen on gfx950 / ROCm 7.2, tests/test_gpu_gemm.py caught it))
                    asm volatile("ds_read_b128 %0, %1" : "=v"(rws[0]) : "v"(row_addr));
                    asm volatile("ds_read2_b64 %0, %1 offset0:129 offset1:128" : "=v"(rws[1]) : "v"(row_addr));
                    asm volatile("ds_read_b128 %0, %1 offset:2048" : "=v"(rws[2]) : "v"(row_addr));
                    asm volatile("ds_read2_b64 %0, %1 offset0:129 offset1:128" : "=v"(rws[3]) : "v"(row_addr + 2048));
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(rws[0]), "+v"(rws[1]), "+v"(rws[2]), "+v"(rws[3]));
                    TR(22);  // patch round trip done
#pragma unroll
                    for (int t = 0; t < 4; ++t)
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4v, rws[t]), rs, rl_voff, row_soff(i, t), HS_GEMM_STORE_AUX);
                    // the data registers stay live across one wait state behind the last store (see above)
                    asm volatile("s_nop 0" ::"v"(rws[0]), "v"(rws[1]), "v"(rws[2]), "v"(rws[3]));
                } else {
#pragma unroll
                    for (int j = 0; j < TJ; ++j)
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const int n = ncol0 + j * 32 + 4 * half + 8 * g;
                            __builtin_amdgcn_raw_buffer_store_b64(o[j][g], rs, n < p.n ? (uint32_t)(ml * n2 + n * 2) : kOob, 0, HS_GEMM_STORE_AUX);
                        }
                }
            };
            if (EPI == EPI_GELU) {
                // The two waves of a SIMD (w and w + NW / 2) take the outputs in OPPOSITE order: the GELU arithmetic is bound by the
                // SIMD's VALU rate (~770 cycles per block and wave alone, twice that when both partners are in it), the h output is
                // LDS patch + store issue (~1100) -- out of phase each hides under the other (tools/gemm_trace.py,
                // profiles/r06_gemm_trace_epilogue_phases.txt)
                if (NW == 8 && wave >= NW / 2) {
                    make_o2();
                    emit(rx, o2);
                    TR(21);
                    __builtin_amdgcn_sched_barrier(0);
                    if (p.c) emit(rc, o1);
                } else {
                    if (p.c) emit(rc, o1);
                    TR(21);  // first output's stores issued
                    __builtin_amdgcn_sched_barrier(0);
                    make_o2();
                    emit(rx, o2);
                }
            } else {
                emit(rc, o1);
            }
            if constexpr (RING) {
                if (i + 2 < TM) request_in(i + 2);  // the patch is free again: its rows are in registers
            }
            TR(12 + i);
        }
        }
    };

    // ---- the stream of k-steps over this workgroup's tiles; the issue cursor runs NSTAGE - 1 steps ahead of the compute cursor
    const int stride = p.blocks_per_xcd;
    int id_i = id0, ks_i = 0;  // next step to issue
    int id_c = id0, ks_c = 0;  // step being computed
    int par_c = 0;             // (LBIAS) slot that holds the bias of the tile being computed
    auto advance_issue = [&]() {
        ++ks_i;
        if (ks_i == nk) {
            ks_i = 0;
            id_i += stride;
            if (id_i < id_end) retarget(id_i, nk1 == 0, true);
        } else if (ks_i == nk1) {
            retarget(id_i, true, false);
        }
    };
    auto kb_of = [&](int ks) { return (ks >= nk1 ? ks - nk1 : ks) * 128; };
    retarget(id_i, false, true);
    constexpr int AHEAD = NSTAGE - 1, PIECES_ALL = AI + BI;
    int issued = 0;  // steps issued and not yet computed
#pragma unroll
    for (int q = 0; q < AHEAD; ++q) {
        if (id_i < id_end) {
            issue_pieces(kb_of(ks_i), q, std::integral_constant<int, 0>{}, std::integral_constant<int, PIECES_ALL>{});
            advance_issue();
            ++issued;
        }
    }
    int buf = 0, buf_free = AHEAD;  // buffer being computed; buffer the next issue goes to
#if HS_GEMM_EXP & 32
    // measurement build: static priority for the second-dispatched half of the waves (the SIMD partners of the DMA-issuing half): at equal
    // priority the older wave wins every arbitration and the younger one sets the pace of the k-step (tools/gemm_trace.py: 2900 vs 2600 cycles)
    if (NW == 8 && wave >= NW / 2) __builtin_amdgcn_s_setprio(1);
#endif
#if HS_GEMM_EXP & 64
    if (NW == 8 && wave < NW / 2) __builtin_amdgcn_s_setprio(1);  // (the opposite assignment, for the A/B)
#endif
    bool drained = false;  // an epilogue's stores are in the queue behind the operand pieces
    constexpr int kEpiOps = TM * (TN / 2 > 0 ? TN / 2 : 1) * 4;  // vector-memory operations of the SMALLEST epilogue (four row-segment stores per row block)
    static_assert((AHEAD - 1) * PIECES_ALL + kEpiOps <= 63, "vmcnt immediate");
    TR(0);
    while (true) {
        TR(3);
        // this wave's loads of the current step have landed: everything older than the steps still allowed in flight
        if (IW < NW && wave >= IW) {
            // (this wave issues no operand DMA -- nothing of its own to wait for; its epilogue stores are fire-and-forget, and
            // an epilogue waits for its own input loads itself)
        } else if (AHEAD > 1 && issued == AHEAD && !drained)
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((AHEAD - 1) * PIECES_ALL) : "memory");
        else if (drained && issued == AHEAD)
            // behind an epilogue the queue holds, youngest last: [this step's pieces] [the later steps' pieces] [the epilogue's stores and
            // input requests].  Waiting for the STORES' acknowledgements (vmcnt(0)) cost the issuing waves -- the ones every barrier waits
            // for -- ~700 cycles per tile (tools/gemm_trace.py: "epilogue end -> next step"); the pieces are covered once no more than the
            // later steps' pieces and kEpiOps epilogue operations remain (kEpiOps = the fewest an epilogue issues: one output's stores)
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((AHEAD - 1) * PIECES_ALL + kEpiOps) : "memory");
        else if (drained && issued == 1)
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kEpiOps) : "memory");
        else
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        drained = false;
        TR(1);
        __builtin_amdgcn_s_barrier();  // ... everyone's have; and everyone is done reading the buffer that is re-filled next
        TR(2);
        const bool more = id_i < id_end;
        const bool last = ks_c + 1 == nk;
        compute(buf, more, kb_of(ks_i), buf_free);
        --issued;
        if (more) {
            advance_issue();
            ++issued;
        }
        const int buf_done = buf;
        buf = buf + 1 == NSTAGE ? 0 : buf + 1;
        buf_free = buf_free + 1 == NSTAGE ? 0 : buf_free + 1;
        if (last) {
            epilogue(id_c, buf_done, par_c);
            par_c ^= 1;
            drained = true;
            ks_c = 0;
            id_c += stride;
            if (id_c >= id_end) break;
        } else {
            ++ks_c;
        }
    }
#ifdef HS_GEMM_TRACE
    if (tr_on) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        uint64_t* out = p.trace + ((blockIdx.x == 0 ? 0 : 1) * NW + wave) * kTraceCap;
        for (int i = lane; i < kTraceCap; i += 64) {
            u32x2 v = {0u, 0u};
            if (i < tr_n) {
                asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(tr_base + i * 8) : "memory");
            }
            out[i] = ((uint64_t)v[1] << 32) | v[0];
        }
    }
#endif
#endif
}

template <int BM, int BN, int WM, int WN, int NSTAGE, bool ALIAS, bool FAST>
int launch_tile(GemmParams& p, int epi, int wgs_per_cu, hipStream_t s) {
    const int tiles_m = (int)((p.m + BM - 1) / BM);
    p.tiles_n = (p.n + BN - 1) / BN;
    const int64_t tiles = (int64_t)tiles_m * p.tiles_n;
    if (tiles > 0x7fffffff) return fail(HS_ERR_UNSUPPORTED, "too many output tiles");
    p.tiles = (int)tiles;
    p.per_xcd = (p.tiles + 7) / 8;
    const int resident = usable_cus_per_xcd() * wgs_per_cu;  // workgroups per XCD in one resident round (32 CUs per XCD minus the reserved ones)
    p.blocks_per_xcd = p.per_xcd < resident ? p.per_xcd : resident;
    const dim3 grid((unsigned)(8 * p.blocks_per_xcd)), block(WM * WN * 64);
    const bool drop = p.drop_p > 0.f;
#define HS_GEMM_LAUNCH(E, D) hipLaunchKernelGGL((gemm_nt_kernel<BM, BN, WM, WN, NSTAGE, ALIAS, E, D, FAST>), grid, block, 0, s, p)
    switch (epi) {
        case EPI_BIAS: HS_GEMM_LAUNCH(EPI_BIAS, false); break;
        case EPI_GELU:
            if (drop) HS_GEMM_LAUNCH(EPI_GELU, true);
            else HS_GEMM_LAUNCH(EPI_GELU, false);
            break;
        case EPI_DGELU:
            if (drop) HS_GEMM_LAUNCH(EPI_DGELU, true);
            else HS_GEMM_LAUNCH(EPI_DGELU, false);
            break;
        default: HS_GEMM_LAUNCH(EPI_RESID, false); break;
    }
#undef HS_GEMM_LAUNCH
    HS_LAUNCH_CHECK("gemm_nt");
    return HS_OK;
}

}  // namespace
}  // namespace hs

namespace {
int g_tile_variant = 0;  // hs_gemm_nt_set_tile (measurement hook): 1 = 128x128, 2 = 256x128, 3 = 256x256; 0 = heuristic
}

#ifdef HS_GEMM_TRACE
namespace {
uint64_t* g_trace = nullptr;
}
extern "C" int hs_gemm_nt_set_trace(void* buf) {
    g_trace = (uint64_t*)buf;
    return 0;
}
#endif

extern "C" {

int hs_gemm_nt_set_tile(int variant) {
    g_tile_variant = variant;
    return HS_OK;
}

int hs_gemm_nt(const void* a, int64_t lda, const void* b, int64_t ldb, int k, const void* a2, int64_t lda2, const void* b2,
               int64_t ldb2, int k2, const float* bias, void* c, void* aux, int64_t m, int n, int epilogue, float drop_p,
               uint64_t seed, int dtype, void* stream) {
    using namespace hs;
    HS_CHECK_ARG(dtype == HS_BF16, "hs_gemm_nt: bf16 activations only (fp32 runs use the library GEMM)");
    HS_CHECK_ARG(a && b && m > 0 && n > 0 && k > 0, "hs_gemm_nt: null operand or empty shape");
    HS_CHECK_ARG(epilogue >= EPI_BIAS && epilogue <= EPI_RESID, "hs_gemm_nt: unknown epilogue %d", epilogue);
    HS_CHECK_ARG(k2 == 0 || (a2 && b2 && k2 > 0), "hs_gemm_nt: second segment needs a2, b2, k2 > 0");
    HS_CHECK_ARG(c || (epilogue == EPI_GELU && aux), "hs_gemm_nt: no output");
    HS_CHECK_ARG(epilogue == EPI_BIAS || aux, "hs_gemm_nt: this epilogue needs aux");
    HS_CHECK_ARG(drop_p >= 0.f && drop_p <= 1.f, "hs_gemm_nt: drop_p must be in [0, 1]");
    // 16-byte operand chunks and 8-byte output groups
    if (k % 8 || k2 % 8 || lda % 8 || ldb % 8 || (k2 && (lda2 % 8 || ldb2 % 8)) || n % 4)
        return fail(HS_ERR_UNSUPPORTED, "hs_gemm_nt: k, k2 and the row strides must be multiples of 8, n a multiple of 4");
    if (lda * 2 * 256 > kMaxRecords || ldb * 2 * 256 > kMaxRecords || (int64_t)n * 2 * 256 > kMaxRecords)
        return fail(HS_ERR_UNSUPPORTED, "hs_gemm_nt: row stride too large");
    GemmParams p{};
    p.a = (const uint16_t*)a; p.b = (const uint16_t*)b; p.lda = lda; p.ldb = ldb; p.k = k;
    p.a2 = (const uint16_t*)a2; p.b2 = (const uint16_t*)b2; p.lda2 = lda2; p.ldb2 = ldb2; p.k2 = k2;
    p.bias = bias; p.c = (uint16_t*)c; p.aux = (uint16_t*)aux; p.m = m; p.n = n;
    p.drop_p = drop_p; p.seed = seed;
#ifdef HS_GEMM_TRACE
    p.trace = g_trace;
#endif
    // Tile variants: 1 = 128x128 x 2 stages, own epilogue patches, two 4-wave workgroups per CU; 2 = 256x128 x 3 stages and
    // 3 = 256x256 x 2 stages: one 8-wave workgroup per CU, patches inside the consumed stage buffer.  Measured choice
    // (tools/bench_gemm_nt.py, profiles/archive_r01_r04/r02_gemm_nt_vs_library.*): 256x128 x 3 is the all-round shape; 256x256 halves the
    // L2 -> LDS fill per flop and wins wide outputs with k >= 512 and every GELU / GELU' epilogue with n >= 512 (fewer,
    // larger tiles: the VALU-bound epilogue is paid per output element, the barrier / drain around it per tile);
    // 128x128 when the launch would not fill the chip otherwise.  (hs_gemm_nt_set_tile forces a variant for A/B runs.)
    int variant = g_tile_variant;
    if (!variant) {
        const int64_t tiles2 = ((m + 255) / 256) * ((n + 127) / 128), tiles3 = ((m + 255) / 256) * ((n + 255) / 256);
        if (tiles2 < 256)
            variant = 1;
        else if ((epilogue == EPI_GELU || epilogue == EPI_DGELU) ? n >= 512 : (n >= 1024 && k + k2 >= 512))
            variant = 3;
        else if (n >= 256 && tiles3 >= 768 && k % 64 == 0 && k2 % 64 == 0)
            // with the role-separated DMA issue (FAST) the 256 x 256 tile also wins the narrower outputs (profiles/
            // r03_gemm_role_split.txt: s0 qkv 365 vs 392 us, s1 qkv 207 vs 258, s2 proj 60 vs 66) as long as its tiles
            // fill three resident rounds (stage-3 proj / fc2, 384 tiles = 1.5 rounds, stay on the 256 x 128 tile)
            variant = 3;
        else
            variant = 2;
    }
    // (the 256 x 256 kernels take an epilogue input through the DMA ring, which moves whole 16-byte chunks)
    if (variant == 3 && (epilogue == EPI_DGELU || epilogue == EPI_RESID) && n % 8) variant = 2;
    hipStream_t st = (hipStream_t)stream;
    // role-separated DMA issue (FAST) wherever there is no K tail (8-16 % on the 256 x 256 tile, profiles/archive_r01_r04/r03_gemm_role_split.txt)
    const bool fast = k % 64 == 0 && k2 % 64 == 0;
    switch (variant) {
        case 2: return fast ? launch_tile<256, 128, 4, 2, 3, true, true>(p, epilogue, 1, st) : launch_tile<256, 128, 4, 2, 3, true, false>(p, epilogue, 1, st);
        case 3: return fast ? launch_tile<256, 256, 2, 4, 2, true, true>(p, epilogue, 1, st) : launch_tile<256, 256, 2, 4, 2, true, false>(p, epilogue, 1, st);
        default: return launch_tile<128, 128, 2, 2, 2, false, false>(p, epilogue, 2, st);
    }
}

}  // extern "C"
