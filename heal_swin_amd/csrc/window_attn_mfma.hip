// MFMA path of the fused  shift -> window_partition -> attention -> window_reverse -> shift_back  op for the
// production shape: window = 64 tokens (8x8 nested block), head_dim = 32, bf16 activations (gfx950 / CDNA4).
//
// Work decomposition
//   forward:  workgroup = HG wavefronts = HG consecutive heads ("head group") of one window at a time; wave g owns head g.
//   backward: workgroup = HG heads x 2 wavefronts (query half in the score phase, key tile in the dK / dV phase).
//   HG*64 B of every token row's q (and k, v, dO) slice are contiguous, so the workgroup's cooperative loads move whole
//   128-B lines (HG = 2/4) with 16 B per lane.  Workgroups are persistent: a fixed head group, grid-striding over the B*nW
//   windows, which keeps the head's relative-position bias (64x64 fp32, pre-multiplied by log2 e) in REGISTERS for the whole
//   launch, in exactly the accumulator layout of the score tile.
//
// Gather/scatter: window w, row i is token  idx[w*64+i]  (or (w*64+i+roll) mod N) of the UNshifted qkv tensor;
// the output row goes back to the same token, so shift and shift_back cost nothing beyond an index load.
//
// Forward, per (window, head), all on one wave:
//   S^T = K Q^T          8 x v_mfma_f32_32x32x16_bf16.  Computing the TRANSPOSED scores puts a whole query row
//                        in one lane pair (lane l and l^32 hold the 64 keys of query l&31), so the row max / sum
//                        are in-register reductions plus one cross-half exchange -- no LDS, no 64-lane butterflies.
//   softmax              t = S^T * (scale*log2 e) + bias*log2 e (+ mask) -> exp2(t - max) / sum, fp32.
//   O^T = V^T P^T        8 x MFMA.  The accumulator registers of S^T are, after bf16 packing, directly the B operand
//                        (lane = query row, 8 key slots); the result has lane = query, registers = features, and every
//                        lane stores 16-byte pieces of its token's row straight to HBM (pack_rows_t).
// LDS images per head: Q, K and V row-major [64][32] bf16 with a 16-B-chunk XOR swizzle (conflict-free ds_read_b128 for the
// MFMA A/B fragments of Q and K; the fragments that need 8 consecutive TOKENS of one feature -- V in the forward, K, Q, dO
// in the backward -- come from the same tiles through ds_read_b64_tr_b16, the hardware 4x16 transpose).
// Cosine attention: k rows are L2-normalised while being staged; the query norm and the head's logit scale are one
// per-lane factor applied to the fp32 scores.
#include "window_attn.h"

// HS_ATTN_STORE_AUX (measurement builds): cache policy of the row stores of O / dQ / dK / dV (gfx950: bit 1 = nt); see csrc/gemm_nt.hip
#ifndef HS_ATTN_STORE_AUX
#define HS_ATTN_STORE_AUX 0
#endif
// HS_ATTN_LOAD_AUX: cache policy of the q / k / v / dO row loads (every row is read once per head group)
#ifndef HS_ATTN_LOAD_AUX
#define HS_ATTN_LOAD_AUX 0
#endif
#ifndef HS_ATTN_LOAD_AUX_FWD
#define HS_ATTN_LOAD_AUX_FWD HS_ATTN_LOAD_AUX
#endif
#ifndef HS_ATTN_LOAD_AUX_BWD
#define HS_ATTN_LOAD_AUX_BWD HS_ATTN_LOAD_AUX
#endif

#include <cstdlib>
#include <type_traits>

namespace hs {
HS_DEFINE_SEED_EPOCH_SETTER(set_seed_epoch_attn_mfma)
namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int kWs = 64, kHd = 32;
constexpr int kTileBytes = kWs * kHd * 2;  // 4096: [64][32] bf16, rows of 64 B = 4 chunks of 16 B
constexpr float kLog2e = 1.4426950408889634f;
constexpr float kLn2 = 0.6931471805599453f;
constexpr float kNormEps = 1e-12f;
// 1 / max(|x|, eps) from the squared norm: v_rsq_f32 (1 ulp) + a clamp instead of the correctly rounded sqrt and division hipcc
// expands to ~20 instructions each -- four of them per row block were half of what cosine attention added to the kernels' VALU
// count (profiles/r05_attn_pmc_T256_vs_D256.txt: 66 vs 37 VALU per MFMA in the forward); results are bf16 rows
__device__ __forceinline__ float inv_norm(float sumsq) { return fminf(__builtin_amdgcn_rsqf(sumsq), 1.f / kNormEps); }
constexpr float kMaskLog2 = -100.f * kLog2e;

__device__ __forceinline__ int swz(int row, int chunk) { return row * 64 + ((chunk ^ ((row >> 2) & 3)) << 4); }
__device__ __forceinline__ float bf_lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf_hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }
__device__ __forceinline__ uint32_t pack_bf16(float a, float b) { return pack_bf16x2(a, b); }

typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

constexpr int kPsLd = 136;                    // bytes per query row of the P / dS' scratch (64 keys * 2 B + 8 pad)
constexpr int kPsBytes = kWs * kPsLd;         // 8704

// 4 token rows x feature column (l&31) from a swizzled row-major [64][32] bf16 tile (hardware-transposed 8-byte read)
__device__ __forceinline__ s16x4 tr_read_tile(const unsigned char* tile, int row0, int lane) {
    const int L = lane & 15, nblk = (lane >> 4) & 1;
    const int row = row0 + (L >> 2);
    const int byte_in_row = nblk * 32 + (L & 3) * 8;
    const int chunk = byte_in_row >> 4;
    const unsigned char* a = tile + row * 64 + (((chunk ^ ((row >> 2) & 3)) << 4) | (byte_in_row & 8));
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)a);
}
// the same from the [64 q][64 key] scratch (row stride kPsLd): rows = q (contraction index), columns = key block colblk*32..
__device__ __forceinline__ s16x4 tr_read_scratch(const unsigned char* scr, int row0, int colblk, int lane) {
    const int L = lane & 15, nblk = (lane >> 4) & 1;
    const unsigned char* a = scr + (row0 + (L >> 2)) * kPsLd + (colblk * 32 + nblk * 16 + (L & 3) * 4) * 2;
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)a);
}
__device__ __forceinline__ bf16x8 join(s16x4 a, s16x4 b) {
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    s16x8 v = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    return __builtin_bit_cast(bf16x8, v);
}

// ================================================================================================ backward
// Per (window, head) 40 MFMAs (the minimum: S, dP, dV, dK, dQ), on the head's two waves:
//   S^T  = K^ Q^T , dP^T = V dO^T                       (lane = query, registers = keys; wave t: queries t*32..)
//   P^T  = exp2(S^T*f + bias - lse) ;  dS^T = P^T o (dP^T - D) ,  D[q] = sum_k P[q][k] dP[q][k]  (= dO[q].O[q], but formed
//                                     from the fp32 P and dP of this very pass: a query's 64 keys sit in one lane pair, so D is
//                                     32 FMAs + one lane^32 exchange, sum_k dS[q][k] = 0 holds to fp32 rounding instead of to the
//                                     bf16 rounding of the stored O, and the O rows are not read at all: 7 instead of 8 row
//                                     streams per token and head)
//   dS' = dS * f_q   (f_q = head scale [* 1/|q| for cosine])  ->  ONE bf16 matrix feeds both dK and dQ
//   dq = X - q^(q^.X) , dk = (dK^ - k^(k^.dK^))/|k|     for cosine attention (plain: dq = X, dk = dK^), X = dS' K^
// The head's bias gradient is accumulated in registers over all windows of the launch (same layout as the bias) and
// written once per workgroup to a partial buffer that a second tiny kernel reduces -- no atomics, deterministic.
//
// Arrangement (round 4; the round-3 kernel was issue / barrier bound -- profiles/archive_r01_r04/r03_attn_bwd_ablation.txt: 34.5 VALU
// instructions per MFMA, 5 barriers per window, 72 % of its time left with all global traffic removed; this one: 23.0, 3
// barriers, stage 0 of HEAL-SWIN-B 680 -> 555-570 us, profiles/archive_r01_r04/r04_attn_v2_ab.txt, r04_attn_pmc_*.json):
//  * TRANSPOSED output products.  dQ^T = K^^T dS'^T, dK^^T = Q^T dS', dV^T = dO^T P are the same MFMAs with the A and B operands
//    exchanged; the accumulator then has lane = token, registers = 4 consecutive features x 4 groups.  bf16 packing gives 8-byte
//    pieces, one v_permlane32_swap per dword pairs them into 16 contiguous bytes, and each lane stores its token's row piece
//    straight to HBM: no 2-byte LDS writes, no staging tiles, no read-back pass, no barrier around it (PMC: HBM traffic stays
//    1.02 x algorithmic -- the 32-byte pieces of a line merge in L2).
//  * dK / dV split by KEY tile instead of by query half.  Wave t of a head writes the P / dS' rows of its 32 queries to two LDS
//    scratches (row-major [q][key]); after one barrier wave t forms dK^T and dV^T of key tile t over ALL 64 queries (4 + 4
//    MFMAs, B operands by ds_read_b64_tr_b16 from the scratches, A operands the same way from the Q / dO tiles).  No fp32
//    partial-sum exchange.
//  * 32-bit addressing through per-image buffer descriptors, window counters advanced without 64-bit divisions, the
//    relative-position bias kept in registers (pre-multiplied by log2 e) for the whole launch, the label scan done once per
//    workgroup by a ballot.
// Loads of window i+1 are requested right after window i's staging barrier and claimed (s_waitcnt) in front of window i's last
// stores, so no wait ever covers a store's round trip.
struct LdsLayoutBwd {
    // per head: Q | K^ | V | dO tiles (4096 each) | dS' scratch | P scratch ([64 q][64 key] bf16, padded rows); then per-row scalars
    int scr_s, scr_p, head, qinv, kinv, lab, flag, total;
    __host__ __device__ explicit LdsLayoutBwd(int hg) {
        scr_s = 4 * kTileBytes;
        scr_p = scr_s + kPsBytes;
        head = scr_p + kPsBytes;  // 33792 bytes per head
        qinv = hg * head;
        kinv = qinv + hg * kWs * 4;
        lab = kinv + hg * kWs * 4;
        flag = lab + kWs;
        total = flag + 16;
    }
};

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;

// 16 accumulator values of a transposed output tile (lane = token, register r = feature (r&3) + 8*(r>>2) + 4*half) -> two
// 16-byte pieces of the token's 64-byte head slice: lanes < 32 hold bytes [0,16) and [32,48), lanes >= 32 bytes [16,32) and [48,64)
__device__ __forceinline__ void pack_rows_t(const float (&v)[16], u32x4& p0, u32x4& p1) {
    uint32_t w[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) w[i] = pack_bf16x2(v[2 * i], v[2 * i + 1]);
    // groups m = 0..3 are dwords (2m, 2m+1); pair (0,1) and pair (2,3): vdst = group m, src = group m+1
#pragma unroll
    for (int m = 0; m < 4; m += 2)
#pragma unroll
        for (int d = 0; d < 2; ++d) {
            const auto r = __builtin_amdgcn_permlane32_swap(w[2 * m + d], w[2 * m + 2 + d], false, false);
            w[2 * m + d] = r[0];
            w[2 * m + 2 + d] = r[1];
        }
    p0 = u32x4{w[0], w[1], w[2], w[3]};
    p1 = u32x4{w[4], w[5], w[6], w[7]};
}

template <int HG, bool DROP, bool COS>
__global__ void __launch_bounds__(128 * HG, 2) attn_bwd_mfma_kernel(AttnParams p, float* __restrict__ dbias_part,
                                                                 float* __restrict__ dscale_part, int slots, int groups) {
    if constexpr (DROP) apply_seed_epoch(p);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const LdsLayoutBwd L(HG);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bxcd = blockIdx.x & 7, blocal = blockIdx.x >> 3;
    const int by = blocal % groups, bx = bxcd + 8 * (blocal / groups);
    if (bx >= slots) return;
    const int g = wv >> 1, t = wv & 1;  // head inside the group; query half (score phase) = key tile (dK / dV phase)
    const int half = lane >> 5, l31 = lane & 31;
    const int h = by * HG + g;
    const int C = p.C, nH = p.nH;
    const int N = (int)p.N;
    const int nW = N / kWs;
    const int total_windows = p.B * nW;  // B * N < 2^31 is checked by the dispatcher
    constexpr bool cosine = COS;
    // register budget (256 at two waves per SIMD): the plain instantiation keeps the bias in registers for the whole launch and
    // requests the next rows right behind the staging barrier; the cosine / dropout ones re-read the bias per window from L2
    // and request the rows once the score accumulators are dead
    // (cosine + dropout, the instantiation with the most live state: no pipelining, the rows are requested at the top of their
    // own window -- as in the round-3 kernel)
    constexpr bool BIASREG = !COS && !DROP, EARLY = !COS && !DROP, NOPIPE = COS && DROP;
    const float hscale = p.head_scale[h];
    const bool has_idx = p.idx != nullptr;
    const int roll = (int)p.roll;
    const uint32_t c3b = 3u * (uint32_t)C * 2u, cb = (uint32_t)C * 2u;  // bytes per token row of qkv / dout
    const uint32_t img_qkv = (uint32_t)N * c3b, img_do = (uint32_t)N * cb;  // bytes per image (< 2^31, dispatcher)

    unsigned char* my = smem + g * L.head;
    unsigned char* q_tile = my;
    unsigned char* k_tile = my + kTileBytes;
    unsigned char* v_tile = my + 2 * kTileBytes;
    unsigned char* do_tile = my + 3 * kTileBytes;
    unsigned char* scr_s = my + L.scr_s;
    unsigned char* scr_p = my + L.scr_p;
    float* qinv_s = (float*)(smem + L.qinv);
    float* kinv_s = (float*)(smem + L.kinv);
    unsigned char* lab_s = smem + L.lab;
    uint32_t* flag_s = (uint32_t*)(smem + L.flag);
    const int qq = t * 32 + l31;  // this lane's query row (score phase) and key row (dK / dV phase)

    // staging geometry: 128*HG threads move 32 rows x HG*64 B per pass, 2 passes per tile.  (Re-derived from an opaque copy
    // of the thread id wherever it is used: hoisted, the per-thread addresses would be pinned in registers for the whole kernel.)
#define HS_STAGE_GEOMETRY                                                                          \
    int tid_o = tid;                                                                               \
    asm volatile("" : "+v"(tid_o));                                                                \
    const int srow = tid_o / (4 * HG), sc = tid_o % (4 * HG), sg = sc >> 2, scc = sc & 3;          \
    const uint32_t colb = (uint32_t)(by * HG * kHd + sc * 8) * 2u;                                 \
    unsigned char* st = smem + sg * L.head;                                                        \
    (void)scc;                                                                                     \
    (void)st;                                                                                      \
    (void)colb;

    auto image_rsrc = [&](const void* base, uint32_t bytes_per_image, int b_l) {
        const uint64_t a = (uint64_t)base + (uint64_t)b_l * bytes_per_image;
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a), hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
        return __builtin_amdgcn_make_buffer_rsrc((void*)(((uint64_t)hi << 32) | lo), 0, (int)bytes_per_image, 0x00020000);
    };
    // natural-order token (inside its image) of shifted row j, without a table
    auto rolled = [&](int j) {
        const int s = j + roll;
        return s >= N ? s - N : s;
    };

    // bias (x log2 e) and bias gradient of this wave's 32 queries, in the S^T accumulator layout, for the whole launch
    float bias2[2][16], dbacc[2][16];
    float dscale_acc = 0.f;

    // ---- software pipeline state: rows of the NEXT window (registers), token rows of the one after (table mode)
    u32x4 ldq[2], ldk[2], ldv[2], lddo[2];
    int tok_ld[2] = {0, 0}, tok_st = 0;  // token rows (inside the image) of the rows in flight: this thread's 2 staging rows, its store row
    int tok_ld2[2] = {0, 0}, tok_st2 = 0;  // table mode: the same for the window after (requested one window ahead of the rows)
    float lse_next = 0.f;
    unsigned lab_next = 0;

    int b_cur = bx / nW, w_cur = bx - b_cur * nW;  // (one 32-bit division per launch)
    auto advance = [&](int& b_l, int& w_l) {
        w_l += slots;
        while (w_l >= nW) {
            w_l -= nW;
            ++b_l;
        }
    };
    auto request_tokens = [&](int w_l) {  // table mode: token rows of window w_l -> tok_ld2 / tok_st2 (loads)
        HS_STAGE_GEOMETRY
        const int j_l = w_l * kWs;
        tok_ld2[0] = p.idx[j_l + srow];
        tok_ld2[1] = p.idx[j_l + 32 + srow];
        tok_st2 = p.idx[j_l + qq];
    };
    auto issue_loads = [&](int b_l, int w_l) {
        HS_STAGE_GEOMETRY
        const int j_l = w_l * kWs;
        if (has_idx) {
            tok_ld[0] = tok_ld2[0];
            tok_ld[1] = tok_ld2[1];
            tok_st = tok_st2;
        } else {
            tok_ld[0] = rolled(j_l + srow);
            tok_ld[1] = rolled(j_l + 32 + srow);
            tok_st = rolled(j_l + qq);
        }
        const __amdgpu_buffer_rsrc_t rq = image_rsrc(p.qkv, img_qkv, b_l), rd = image_rsrc(p.dout, img_do, b_l);
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
            const uint32_t vo = (uint32_t)tok_ld[rb] * c3b + colb;
            ldq[rb] = __builtin_amdgcn_raw_buffer_load_b128(rq, vo, 0, HS_ATTN_LOAD_AUX_BWD);
            ldk[rb] = __builtin_amdgcn_raw_buffer_load_b128(rq, vo, cb, HS_ATTN_LOAD_AUX_BWD);
            ldv[rb] = __builtin_amdgcn_raw_buffer_load_b128(rq, vo, 2 * cb, HS_ATTN_LOAD_AUX_BWD);
            lddo[rb] = __builtin_amdgcn_raw_buffer_load_b128(rd, (uint32_t)tok_ld[rb] * cb + colb, 0, HS_ATTN_LOAD_AUX_BWD);
        }
        lse_next = p.lse[((int64_t)b_l * nH + h) * N + j_l + qq];
        if (p.labels && wv == 0) lab_next = p.labels[j_l + lane];
    };
    // every value the prefetch produced is claimed at ONE point (the compiler puts its s_waitcnt vmcnt there); see the loop
    auto claim = [&]() {
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
            asm volatile("" : "+v"(ldq[rb]), "+v"(ldk[rb]), "+v"(ldv[rb]), "+v"(lddo[rb]));
        }
        asm volatile("" : "+v"(lse_next), "+v"(lab_next), "+v"(tok_ld2[0]), "+v"(tok_ld2[1]), "+v"(tok_st2));
    };
    auto lds_barrier = [&]() {  // orders LDS traffic only: global loads and stores stay in flight
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };

    if (bx < total_windows) {
        if (has_idx) {
            request_tokens(w_cur);
            asm volatile("" : "+v"(tok_ld2[0]), "+v"(tok_ld2[1]), "+v"(tok_st2));
        }
        if constexpr (!NOPIPE) {
            issue_loads(b_cur, w_cur);
            int b_n = b_cur, w_n = w_cur;
            advance(b_n, w_n);
            if (has_idx && b_n < p.B) request_tokens(w_n);
        }
    }
    const bool has_bias = p.bias != nullptr;
    const float* bsrc = has_bias ? p.bias + ((int64_t)h * kWs + qq) * kWs + 4 * half : (const float*)p.qkv + 4 * half;
    const float bscale = has_bias ? kLog2e : 0.f;  // (unconditional loads: without a bias they read qkv bytes and are zeroed here)
    if constexpr (BIASREG) {
        float4 b4[2][4];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) b4[kt][rg] = *(const float4*)(bsrc + kt * 32 + 8 * rg);
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                bias2[kt][4 * rg] = b4[kt][rg].x * bscale;
                bias2[kt][4 * rg + 1] = b4[kt][rg].y * bscale;
                bias2[kt][4 * rg + 2] = b4[kt][rg].z * bscale;
                bias2[kt][4 * rg + 3] = b4[kt][rg].w * bscale;
            }
    }
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) dbacc[kt][r] = 0.f;
    claim();  // (once in front of the loop: inside it the rows are never "pending" at the loop header)

    for (int wi = bx; wi < total_windows; wi += slots) {
        const int b = b_cur, w = w_cur;
        const int j0 = w * kWs;
        if constexpr (NOPIPE) {
            issue_loads(b, w);
            int b_n = b, w_n = w;
            advance(b_n, w_n);
            if (has_idx && b_n < p.B) request_tokens(w_n);  // (table mode: the NEXT window's token rows still travel one window ahead)
            claim();
        }
        const int tok_store = tok_st;  // this lane's token row of the current window
        const float lse_cur = lse_next;
        HS_STAGE_GEOMETRY
        int lane_o = lane;
        asm volatile("" : "+v"(lane_o));  // (per-lane LDS addresses are re-derived per window instead of pinned in ~40 registers)
        const int half = lane_o >> 5, l31 = lane_o & 31;
        const int qq = t * 32 + l31;
        const int lane = lane_o;

        // ------------------------------------------------------------ stage q, k^, v, dO; norms; label scan
        if (p.labels && wv == 0) {
            lab_s[lane] = (unsigned char)lab_next;
            const unsigned first = __builtin_amdgcn_readfirstlane(lab_next);
            const bool any = __ballot(lab_next != first) != 0ull;
            if (lane == 0) flag_s[0] = any ? 1u : 0u;
        }
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
            const int row = rb * 32 + srow;
            const u32x4 vq = ldq[rb];
            u32x4 vk = ldk[rb];
            if (cosine) {
                float sq = 0.f, sk = 0.f;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    sq += bf_lo(vq[i]) * bf_lo(vq[i]) + bf_hi(vq[i]) * bf_hi(vq[i]);
                    sk += bf_lo(vk[i]) * bf_lo(vk[i]) + bf_hi(vk[i]) * bf_hi(vk[i]);
                }
                sq += __shfl_xor(sq, 1, 64);
                sq += __shfl_xor(sq, 2, 64);
                sk += __shfl_xor(sk, 1, 64);
                sk += __shfl_xor(sk, 2, 64);
                const float kinv = inv_norm(sk);
#pragma unroll
                for (int i = 0; i < 4; ++i) vk[i] = pack_bf16(bf_lo(vk[i]) * kinv, bf_hi(vk[i]) * kinv);
                if (scc == 0) {
                    qinv_s[sg * kWs + row] = inv_norm(sq);
                    kinv_s[sg * kWs + row] = kinv;
                }
            }
            const int off = swz(row, scc);
            *(u32x4*)(st + off) = vq;
            *(u32x4*)(st + kTileBytes + off) = vk;
            *(u32x4*)(st + 2 * kTileBytes + off) = ldv[rb];
            *(u32x4*)(st + 3 * kTileBytes + off) = lddo[rb];
        }
        lds_barrier();  // A: tiles, norms, labels visible

        // ------------------------------------------------------------ request the next window's rows (a whole window ahead)
        int b_n = b_cur, w_n = w_cur;
        advance(b_n, w_n);
        auto prefetch = [&]() {
            if (wi + slots < total_windows) {
                issue_loads(b_n, w_n);
                if (has_idx) {
                    int b_nn = b_n, w_nn = w_n;
                    advance(b_nn, w_nn);
                    if (b_nn < p.B) request_tokens(w_nn);
                }
            }
        };
        if constexpr (EARLY) prefetch();
        b_cur = b_n;
        w_cur = w_n;
        const bool mixed = p.labels ? (flag_s[0] != 0u) : false;  // wave-uniform (one LDS word)

        // ------------------------------------------------------------ S^T = K^ Q^T and dP^T = V dO^T for this wave's 32 queries
        float4 biasv[2][4];
        if constexpr (!BIASREG) {  // in flight (L2) during the MFMAs below
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) biasv[kt][rg] = *(const float4*)(bsrc + kt * 32 + 8 * rg);
        }
        f32x16 accS[2], accP[2];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                accS[kt][r] = 0.f;
                accP[kt][r] = 0.f;
            }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int chunk = ks * 2 + half;
            const bf16x8 qf = *(const bf16x8*)(q_tile + swz(qq, chunk));
            const bf16x8 df = *(const bf16x8*)(do_tile + swz(qq, chunk));
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
                const int off = swz(kt * 32 + l31, chunk);
                const bf16x8 kf = *(const bf16x8*)(k_tile + off);
                const bf16x8 vf = *(const bf16x8*)(v_tile + off);
                accS[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf, accS[kt], 0, 0, 0);
                accP[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, df, accP[kt], 0, 0, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);

        // ------------------------------------------------------------ P, dS' (fp32); bias / scale gradients
        const float qinv = cosine ? qinv_s[g * kWs + qq] : 1.f;
        const float fqn = hscale * qinv;  // d s / d (q . k^)
        const float fq2 = fqn * kLog2e;
        const float nlse2 = -lse_cur * kLog2e;
        const DropRng rng(p, ((int64_t)b * nH + h) * N + j0 + qq);
        if constexpr (!BIASREG) {
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) {
                    bias2[kt][4 * rg] = biasv[kt][rg].x * bscale;
                    bias2[kt][4 * rg + 1] = biasv[kt][rg].y * bscale;
                    bias2[kt][4 * rg + 2] = biasv[kt][rg].z * bscale;
                    bias2[kt][4 * rg + 3] = biasv[kt][rg].w * bscale;
                }
        }
        float dsum = 0.f, s_pds = 0.f, s_ps = 0.f;
        uint32_t keepbits = 0u;
        uint32_t pS[2][8], pP[2][8];
        {
        // (Measured, not kept: the same arithmetic on register PAIRS with v_pk_fma / v_pk_mul / v_pk_add -- 17 instead of 23 VALU
        // instructions per MFMA -- is 4-5 % SLOWER at every stage (stage 0: 584-591 us against 555-570, stage 2: 171 against
        // 163-166; profiles/archive_r01_r04/r04_attn_v2_ab.txt): a packed fp32 op occupies the SIMD for two passes, so it frees issue slots but no
        // ALU time, and it lengthens every dependent chain.)
        // pass 1: P and the (dropout-masked) dP in place; D = sum_k P dP over this lane's 32 keys
        auto pass1 = [&](auto masked_tag) {
            constexpr bool MASKED = decltype(masked_tag)::value;
            int mylab = 0;
            if constexpr (MASKED) mylab = lab_s[qq];
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float sraw = accS[kt][r];
                    float tt = fmaf(sraw, fq2, bias2[kt][r]);
                    if constexpr (MASKED)
                        if (lab_s[kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half] != mylab) tt += kMaskLog2;
                    const float pr = __builtin_amdgcn_exp2f(tt + nlse2);
                    float dpv = accP[kt][r];
                    if constexpr (DROP) {  // the keep bit is kept for pass 2 (the mask is evaluated once per probability)
                        const float mk = rng.mult_half(kt * 32 + (r & 3) + 8 * (r >> 2), half);
                        keepbits |= (mk != 0.f ? 1u : 0u) << (kt * 16 + r);
                        dpv *= mk;
                    }
                    dsum = fmaf(pr, dpv, dsum);
                    if constexpr (COS) {
                        const float ps = pr * sraw;
                        s_ps += ps;
                        s_pds = fmaf(ps, dpv, s_pds);
                    }
                    accS[kt][r] = pr;
                    accP[kt][r] = dpv;
                    if constexpr (COS)
                        if ((r & 7) == 7) asm volatile("" : "+v"(dsum), "+v"(s_ps), "+v"(s_pds));
                }
        };
        if (mixed)
            pass1(std::true_type{});
        else
            pass1(std::false_type{});
        dsum += __shfl_xor(dsum, 32, 64);  // the other 32 keys of this query
        if constexpr (COS) dscale_acc = fmaf(qinv, s_pds - dsum * s_ps, dscale_acc);
        // pass 2: dS = P o (dP - D); bias gradient; bf16 operands  pS = dS' = dS * f_q,  pP = dropped P
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                float ds2[2], pp2[2];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int r = 2 * i + e;
                    const float pr = accS[kt][r];
                    const float dsv = pr * (accP[kt][r] - dsum);
                    dbacc[kt][r] += dsv;
                    ds2[e] = dsv * fqn;
                    pp2[e] = DROP ? (((keepbits >> (kt * 16 + r)) & 1u) ? pr * rng.keep_scale : 0.f) : pr;
                }
                pS[kt][i] = pack_bf16x2(ds2[0], ds2[1]);
                pP[kt][i] = pack_bf16x2(pp2[0], pp2[1]);
            }
        }
        // rows of this lane's query in the two scratches: keys kt*32 + 8*rg + 4*half .. +3 are registers 4rg..4rg+3
        {
            unsigned char* rs = scr_s + qq * kPsLd + 8 * half;
            unsigned char* rp = scr_p + qq * kPsLd + 8 * half;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) {
                    *(uint2*)(rs + (kt * 32 + 8 * rg) * 2) = make_uint2(pS[kt][2 * rg], pS[kt][2 * rg + 1]);
                    *(uint2*)(rp + (kt * 32 + 8 * rg) * 2) = make_uint2(pP[kt][2 * rg], pP[kt][2 * rg + 1]);
                }
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (!EARLY && !NOPIPE) prefetch();

        const __amdgpu_buffer_rsrc_t rdq = image_rsrc(p.dqkv, img_qkv, b);
        const uint32_t vst = (uint32_t)tok_store * c3b + (uint32_t)(h * kHd) * 2u + 16u * half;
        // ------------------------------------------------------------ dQ^T = K^^T dS'^T for this wave's 32 queries (lane = query)
        {
            f32x16 xt;
#pragma unroll
            for (int r = 0; r < 16; ++r) xt[r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int kt = ks >> 1, c = ks & 1;
                const int kbase = kt * 32 + c * 16 + 4 * half;
                const bf16x8 ak = join(tr_read_tile(k_tile, kbase, lane), tr_read_tile(k_tile, kbase + 8, lane));
                const u32x4 bw = {pS[kt][4 * c], pS[kt][4 * c + 1], pS[kt][4 * c + 2], pS[kt][4 * c + 3]};
                xt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ak, __builtin_bit_cast(bf16x8, bw), xt, 0, 0, 0);
            }
            float x[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) x[r] = xt[r];
            if (cosine) {  // dq = X - q^ (q^ . X): features 8m + 4*half .. +3 of the raw q row are one 8-byte LDS read
                float qv[16];
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const uint2 wq = *(const uint2*)(q_tile + swz(qq, m) + 8 * half);
                    qv[4 * m] = bf_lo(wq.x);
                    qv[4 * m + 1] = bf_hi(wq.x);
                    qv[4 * m + 2] = bf_lo(wq.y);
                    qv[4 * m + 3] = bf_hi(wq.y);
                }
                float pq = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) pq = fmaf(x[r], qv[r], pq);
                pq += __shfl_xor(pq, 32, 64);
                pq *= qinv * qinv;
#pragma unroll
                for (int r = 0; r < 16; ++r) x[r] = fmaf(-qv[r], pq, x[r]);
            }
            u32x4 p0, p1;
            pack_rows_t(x, p0, p1);
            __builtin_amdgcn_raw_buffer_store_b128(p0, rdq, vst, 0, HS_ATTN_STORE_AUX);
            __builtin_amdgcn_raw_buffer_store_b128(p1, rdq, vst + 32u, 0, HS_ATTN_STORE_AUX);
        }
        __builtin_amdgcn_sched_barrier(0);
        lds_barrier();  // B: both scratches complete

        // ------------------------------------------------------------ dK^^T = Q^T dS' and dV^T = dO^T P for key tile t, all 64 queries
        u32x4 k0, k1, v0, v1;
        {
            f32x16 kt_acc, vt_acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                kt_acc[r] = 0.f;
                vt_acc[r] = 0.f;
            }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {  // 16 queries per step: half 0 -> +0..7, half 1 -> +8..15
                const int qrow = ks * 16 + 8 * half;
                const bf16x8 aq = join(tr_read_tile(q_tile, qrow, lane), tr_read_tile(q_tile, qrow + 4, lane));
                const bf16x8 bs = join(tr_read_scratch(scr_s, qrow, t, lane), tr_read_scratch(scr_s, qrow + 4, t, lane));
                kt_acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aq, bs, kt_acc, 0, 0, 0);
                const bf16x8 ao = join(tr_read_tile(do_tile, qrow, lane), tr_read_tile(do_tile, qrow + 4, lane));
                const bf16x8 bp = join(tr_read_scratch(scr_p, qrow, t, lane), tr_read_scratch(scr_p, qrow + 4, t, lane));
                vt_acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ao, bp, vt_acc, 0, 0, 0);
            }
            float x[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) x[r] = kt_acc[r];
            if (cosine) {  // dk = (dK^ - k^ (k^ . dK^)) / |k|   (the K tile holds the normalised rows)
                float kv[16];
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const uint2 wk = *(const uint2*)(k_tile + swz(qq, m) + 8 * half);
                    kv[4 * m] = bf_lo(wk.x);
                    kv[4 * m + 1] = bf_hi(wk.x);
                    kv[4 * m + 2] = bf_lo(wk.y);
                    kv[4 * m + 3] = bf_hi(wk.y);
                }
                float pk = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) pk = fmaf(x[r], kv[r], pk);
                pk += __shfl_xor(pk, 32, 64);
                const float kinv = kinv_s[g * kWs + qq];
#pragma unroll
                for (int r = 0; r < 16; ++r) x[r] = fmaf(-kv[r], pk, x[r]) * kinv;
            }
            pack_rows_t(x, k0, k1);
#pragma unroll
            for (int r = 0; r < 16; ++r) x[r] = vt_acc[r];
            pack_rows_t(x, v0, v1);
        }
        // The prefetched rows (requested a whole window ago, long landed) are claimed HERE, in front of the last stores: vmcnt
        // counts loads and stores together and the two complete out of order, so a wait for the rows at the top of the next
        // window would also wait for these stores' round trip.
        if constexpr (!NOPIPE) claim();
        __builtin_amdgcn_raw_buffer_store_b128(k0, rdq, vst, cb, HS_ATTN_STORE_AUX);
        __builtin_amdgcn_raw_buffer_store_b128(k1, rdq, vst + 32u, cb, HS_ATTN_STORE_AUX);
        __builtin_amdgcn_raw_buffer_store_b128(v0, rdq, vst, 2 * cb, HS_ATTN_STORE_AUX);
        __builtin_amdgcn_raw_buffer_store_b128(v1, rdq, vst + 32u, 2 * cb, HS_ATTN_STORE_AUX);
        lds_barrier();  // C: every wave is done with the tiles and the scratches
    }

    // ------------------------------------------------------------ per-workgroup partial parameter gradients
    if (dbias_part) {
        float* dst = dbias_part + ((int64_t)bx * nH + h) * kWs * kWs + (int64_t)qq * kWs;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg)
                *(float4*)(dst + kt * 32 + 8 * rg + 4 * half) =
                    make_float4(dbacc[kt][4 * rg], dbacc[kt][4 * rg + 1], dbacc[kt][4 * rg + 2], dbacc[kt][4 * rg + 3]);
    }
    if (dscale_part) {  // two waves per head: [slot][head][t]
        const float tot = wave_sum(dscale_acc);
        if (lane == 0) dscale_part[((int64_t)bx * nH + h) * 2 + t] = tot;
    }
}
#undef HS_STAGE_GEOMETRY

// ================================================================================================ forward
// The output product is transposed like the backward's (O^T = V^T P^T: lane = query, registers = features, each
// lane stores two 16-byte pieces of its token's head slice per query tile): no 2-byte LDS writes of O, no read-back pass, no
// barrier around it; V is staged ROW-major like Q and K (one 16-byte LDS write instead of eight 2-byte transposing ones) and its
// B fragments come from ds_read_b64_tr_b16.  32-bit addressing through per-image buffer descriptors; two barriers per window.
struct LdsLayoutFwd {
    int head, qinv, lab, flag, total;  // per head: Q | K^ | V tiles (4096 each)
    __host__ __device__ explicit LdsLayoutFwd(int hg) {
        head = 3 * kTileBytes;
        qinv = hg * head;
        lab = qinv + hg * kWs * 4;
        flag = lab + kWs;
        total = flag + 16;
    }
};

template <int HG, bool DROP>
__global__ void __launch_bounds__(64 * HG, 2) attn_fwd_mfma_kernel(AttnParams p, int slots, int groups) {
    if constexpr (DROP) apply_seed_epoch(p);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const LdsLayoutFwd L(HG);
    const int tid = threadIdx.x, lane = tid & 63;
    const int g = __builtin_amdgcn_readfirstlane(tid >> 6);  // head inside the group
    const int bxcd = blockIdx.x & 7, blocal = blockIdx.x >> 3;
    const int by = blocal % groups, bx = bxcd + 8 * (blocal / groups);
    if (bx >= slots) return;
    const int half = lane >> 5, l31 = lane & 31;
    const int h = by * HG + g;
    const int C = p.C, nH = p.nH;
    const int N = (int)p.N;
    const int nW = N / kWs;
    const int total_windows = p.B * nW;
    const bool cosine = (p.flags & HS_ATTN_COSINE) != 0;
    const float hscale = p.head_scale[h];
    const bool has_idx = p.idx != nullptr;
    const int roll = (int)p.roll;
    const uint32_t c3b = 3u * (uint32_t)C * 2u, cb = (uint32_t)C * 2u;
    const uint32_t img_qkv = (uint32_t)N * c3b, img_out = (uint32_t)N * cb;

    unsigned char* q_tile = smem + g * L.head;
    unsigned char* k_tile = q_tile + kTileBytes;
    unsigned char* v_tile = q_tile + 2 * kTileBytes;
    float* qinv_s = (float*)(smem + L.qinv);
    unsigned char* lab_s = smem + L.lab;
    uint32_t* flag_s = (uint32_t*)(smem + L.flag);

    // staging geometry: 12 steps = 3 parts (q, k, v) x 4 row blocks of 16 rows; a step moves 16 rows x HG*64 B.  (Re-derived
    // from an opaque copy of the thread id wherever it is used instead of being pinned in registers for the whole kernel.)
#define HS_STAGE_GEOMETRY                                                                          \
    int tid_o = tid;                                                                               \
    asm volatile("" : "+v"(tid_o));                                                                \
    const int srow = tid_o / (4 * HG), sc = tid_o % (4 * HG), sg = sc >> 2, scc = sc & 3;          \
    const uint32_t colb = (uint32_t)(by * HG * kHd + sc * 8) * 2u;                                 \
    unsigned char* st = smem + sg * L.head;                                                        \
    const int l31 = tid_o & 31;                                                                    \
    (void)scc;                                                                                     \
    (void)st;                                                                                      \
    (void)colb;                                                                                    \
    (void)l31;

    auto image_rsrc = [&](const void* base, uint32_t bytes_per_image, int b_l) {
        const uint64_t a = (uint64_t)base + (uint64_t)b_l * bytes_per_image;
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a), hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
        return __builtin_amdgcn_make_buffer_rsrc((void*)(((uint64_t)hi << 32) | lo), 0, (int)bytes_per_image, 0x00020000);
    };
    auto rolled = [&](int j) {
        const int s = j + roll;
        return s >= N ? s - N : s;
    };

    u32x4 ld[3][4];
    int tok_ld[4] = {0, 0, 0, 0}, tok_st[2] = {0, 0};
    int tok_ld2[4] = {0, 0, 0, 0}, tok_st2[2] = {0, 0};  // table mode: token rows of the window after the one in flight
    unsigned lab_next = 0;
    int b_cur = bx / nW, w_cur = bx - b_cur * nW;
    auto advance = [&](int& b_l, int& w_l) {
        w_l += slots;
        while (w_l >= nW) {
            w_l -= nW;
            ++b_l;
        }
    };
    auto request_tokens = [&](int w_l) {
        HS_STAGE_GEOMETRY
        const int j_l = w_l * kWs;
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) tok_ld2[rb] = p.idx[j_l + rb * 16 + srow];
        tok_st2[0] = p.idx[j_l + l31];
        tok_st2[1] = p.idx[j_l + 32 + l31];
    };
    auto issue_loads = [&](int b_l, int w_l) {
        HS_STAGE_GEOMETRY
        const int j_l = w_l * kWs;
        if (has_idx) {
#pragma unroll
            for (int rb = 0; rb < 4; ++rb) tok_ld[rb] = tok_ld2[rb];
            tok_st[0] = tok_st2[0];
            tok_st[1] = tok_st2[1];
        } else {
#pragma unroll
            for (int rb = 0; rb < 4; ++rb) tok_ld[rb] = rolled(j_l + rb * 16 + srow);
            tok_st[0] = rolled(j_l + l31);
            tok_st[1] = rolled(j_l + 32 + l31);
        }
        const __amdgpu_buffer_rsrc_t rq = image_rsrc(p.qkv, img_qkv, b_l);
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) {
            const uint32_t vo = (uint32_t)tok_ld[rb] * c3b + colb;
            ld[0][rb] = __builtin_amdgcn_raw_buffer_load_b128(rq, vo, 0, HS_ATTN_LOAD_AUX_FWD);
            ld[1][rb] = __builtin_amdgcn_raw_buffer_load_b128(rq, vo, cb, HS_ATTN_LOAD_AUX_FWD);
            ld[2][rb] = __builtin_amdgcn_raw_buffer_load_b128(rq, vo, 2 * cb, HS_ATTN_LOAD_AUX_FWD);
        }
        if (p.labels && g == 0) lab_next = p.labels[j_l + lane];
    };
    auto claim = [&]() {
#pragma unroll
        for (int part = 0; part < 3; ++part)
#pragma unroll
            for (int rb = 0; rb < 4; ++rb) asm volatile("" : "+v"(ld[part][rb]));
        asm volatile("" : "+v"(lab_next), "+v"(tok_ld2[0]), "+v"(tok_ld2[1]), "+v"(tok_ld2[2]), "+v"(tok_ld2[3]), "+v"(tok_st2[0]),
                     "+v"(tok_st2[1]));
    };
    auto lds_barrier = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };

    if (bx < total_windows) {
        if (has_idx) {
            request_tokens(w_cur);
            asm volatile("" : "+v"(tok_ld2[0]), "+v"(tok_ld2[1]), "+v"(tok_ld2[2]), "+v"(tok_ld2[3]), "+v"(tok_st2[0]), "+v"(tok_st2[1]));
        }
        issue_loads(b_cur, w_cur);
        int b_n = b_cur, w_n = w_cur;
        advance(b_n, w_n);
        if (has_idx && b_n < p.B) request_tokens(w_n);
    }

    // relative-position bias of this head (x log2 e), in the S^T accumulator layout: tile (kt, qt), register r holds query
    // qt*32 + l31, key kt*32 + (r&3) + 8*(r>>2) + 4*half.  Unconditional loads (without a bias they read qkv bytes, zeroed below).
    float biasr[2][2][16];
    {
        const bool has_bias = p.bias != nullptr;
        const float* bsrc = has_bias ? p.bias + ((int64_t)h * kWs + l31) * kWs + 4 * half : (const float*)p.qkv + 4 * half;
        const float bs = has_bias ? kLog2e : 0.f;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            float4 b4[2][4];
#pragma unroll
            for (int qt = 0; qt < 2; ++qt)
#pragma unroll
                for (int m = 0; m < 4; ++m) b4[qt][m] = *(const float4*)(bsrc + (has_bias ? qt * 32 * kWs : 0) + kt * 32 + 8 * m);
#pragma unroll
            for (int qt = 0; qt < 2; ++qt)
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    biasr[kt][qt][4 * m] = b4[qt][m].x * bs;
                    biasr[kt][qt][4 * m + 1] = b4[qt][m].y * bs;
                    biasr[kt][qt][4 * m + 2] = b4[qt][m].z * bs;
                    biasr[kt][qt][4 * m + 3] = b4[qt][m].w * bs;
                }
        }
    }
    claim();

    for (int wi = bx; wi < total_windows; wi += slots) {
        const int b = b_cur, w = w_cur;
        const int j0 = w * kWs;
        const int tq0 = tok_st[0], tq1 = tok_st[1];
        HS_STAGE_GEOMETRY
        const int half = tid_o >> 5 & 1;
        const int lane = tid_o & 63;

        // ------------------------------------------------------------ stage q, k^, v (row-major, swizzled); norms; label scan
        if (p.labels && g == 0) {
            lab_s[lane] = (unsigned char)lab_next;
            const unsigned first = __builtin_amdgcn_readfirstlane(lab_next);
            const bool any = __ballot(lab_next != first) != 0ull;
            if (lane == 0) flag_s[0] = any ? 1u : 0u;
        }
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) {
            const int row = rb * 16 + srow;
            const u32x4 vq = ld[0][rb];
            u32x4 vk = ld[1][rb];
            if (cosine) {
                float sq = 0.f, sk = 0.f;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    sq += bf_lo(vq[i]) * bf_lo(vq[i]) + bf_hi(vq[i]) * bf_hi(vq[i]);
                    sk += bf_lo(vk[i]) * bf_lo(vk[i]) + bf_hi(vk[i]) * bf_hi(vk[i]);
                }
                sq += __shfl_xor(sq, 1, 64);
                sq += __shfl_xor(sq, 2, 64);
                sk += __shfl_xor(sk, 1, 64);
                sk += __shfl_xor(sk, 2, 64);
                const float kinv = inv_norm(sk);
#pragma unroll
                for (int i = 0; i < 4; ++i) vk[i] = pack_bf16(bf_lo(vk[i]) * kinv, bf_hi(vk[i]) * kinv);
                if (scc == 0) qinv_s[sg * kWs + row] = inv_norm(sq);
            }
            const int off = swz(row, scc);
            *(u32x4*)(st + off) = vq;
            *(u32x4*)(st + kTileBytes + off) = vk;
            *(u32x4*)(st + 2 * kTileBytes + off) = ld[2][rb];
        }
        lds_barrier();  // A

        int b_n = b_cur, w_n = w_cur;
        advance(b_n, w_n);
        auto prefetch = [&]() {
            if (wi + slots < total_windows) {
                issue_loads(b_n, w_n);
                if (has_idx) {
                    int b_nn = b_n, w_nn = w_n;
                    advance(b_nn, w_nn);
                    if (b_nn < p.B) request_tokens(w_nn);
                }
            }
        };
        // (the dropout instantiation is at the register limit during the softmax: it requests the rows behind it)
        if constexpr (!DROP) prefetch();
        b_cur = b_n;
        w_cur = w_n;
        const bool mixed = p.labels ? (flag_s[0] != 0u) : false;

        // ------------------------------------------------------------ S^T = K^ Q^T
        f32x16 acc[2][2];
        const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 kf[2], qf[2];
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                const int row = tt * 32 + l31, chunk = ks * 2 + half;
                kf[tt] = *(const bf16x8*)(k_tile + swz(row, chunk));
                qf[tt] = *(const bf16x8*)(q_tile + swz(row, chunk));
            }
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int qt = 0; qt < 2; ++qt)
                    acc[kt][qt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[kt], qf[qt], ks == 0 ? zero16 : acc[kt][qt], 0, 0, 0);
        }

        // ------------------------------------------------------------ softmax over the keys of each query (log2 domain)
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            const int qq = qt * 32 + l31;
            const float fq = hscale * kLog2e * (cosine ? qinv_s[g * kWs + qq] : 1.f);
            float m = -INFINITY;
            if (!mixed) {
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float tv = fmaf(acc[kt][qt][r], fq, biasr[kt][qt][r]);
                        acc[kt][qt][r] = tv;
                        m = fmaxf(m, tv);
                    }
            } else {  // rare: windows cut by the shift boundary
                const int my = lab_s[qq];
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int key = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                        float tv = fmaf(acc[kt][qt][r], fq, biasr[kt][qt][r]);
                        if (lab_s[key] != my) tv += kMaskLog2;
                        acc[kt][qt][r] = tv;
                        m = fmaxf(m, tv);
                    }
            }
            m = fmaxf(m, __shfl_xor(m, 32, 64));
            float l = 0.f;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float e = __builtin_amdgcn_exp2f(acc[kt][qt][r] - m);
                    acc[kt][qt][r] = e;
                    l += e;
                }
            l += __shfl_xor(l, 32, 64);
            const float linv = __builtin_amdgcn_rcpf(l);  // (1 ulp; l in [1, 64])
            if constexpr (DROP) {
                const DropRng rng(p, ((int64_t)b * nH + h) * N + j0 + qq);
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        acc[kt][qt][r] *= linv * rng.mult_half(kt * 32 + (r & 3) + 8 * (r >> 2), half);
            } else {
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[kt][qt][r] *= linv;
            }
            if (p.lse && half == 0) p.lse[((int64_t)b * nH + h) * N + j0 + qq] = (m + __builtin_amdgcn_logf(l)) * kLn2;
        }

        if constexpr (DROP) prefetch();
        // ------------------------------------------------------------ O^T = V^T P^T (lane = query, registers = features)
        f32x16 o[2];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int kt = ks >> 1, c = ks & 1;
            const int kbase = kt * 32 + c * 16 + 4 * half;  // slots 0..3 -> keys kbase.., slots 4..7 -> kbase+8..
            const bf16x8 vf = join(tr_read_tile(v_tile, kbase, lane), tr_read_tile(v_tile, kbase + 8, lane));
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) {
                bf16x8 pf;
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) pf[jj] = (__bf16)acc[kt][qt][8 * c + jj];
                o[qt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, ks == 0 ? zero16 : o[qt], 0, 0, 0);
            }
        }
        u32x4 o00, o01, o10, o11;
        {
            float x[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) x[r] = o[0][r];
            pack_rows_t(x, o00, o01);
#pragma unroll
            for (int r = 0; r < 16; ++r) x[r] = o[1][r];
            pack_rows_t(x, o10, o11);
        }
        // the next window's rows are claimed in front of the stores (see the backward)
        claim();
        const __amdgpu_buffer_rsrc_t ro = image_rsrc(p.out, img_out, b);
        const uint32_t hb = (uint32_t)(h * kHd) * 2u + 16u * half;
        __builtin_amdgcn_raw_buffer_store_b128(o00, ro, (uint32_t)tq0 * cb + hb, 0, HS_ATTN_STORE_AUX);
        __builtin_amdgcn_raw_buffer_store_b128(o01, ro, (uint32_t)tq0 * cb + hb + 32u, 0, HS_ATTN_STORE_AUX);
        __builtin_amdgcn_raw_buffer_store_b128(o10, ro, (uint32_t)tq1 * cb + hb, 0, HS_ATTN_STORE_AUX);
        __builtin_amdgcn_raw_buffer_store_b128(o11, ro, (uint32_t)tq1 * cb + hb + 32u, 0, HS_ATTN_STORE_AUX);
        lds_barrier();  // B: every wave is done with the tiles
    }
}
#undef HS_STAGE_GEOMETRY

// dst[e] += sum over parts of src[part][e].  A block of 16 waves owns 256 consecutive elements (one float4 per lane); wave w sums
// the parts p = w, w + 16, ... with all of its loads in flight at once, the 16 partial rows are combined through LDS in a fixed
// order (deterministic).  (Round 3 gave one thread one element and all `parts` loads in sequence: 64 blocks and 17.8 us at
// stage 0 of HEAL-SWIN-B, where the table is smallest and the number of parts largest.)
__global__ void __launch_bounds__(1024) reduce_partials_kernel(const float* __restrict__ src, float* __restrict__ dst, int parts, int64_t n,
                                                                int overwrite) {
    __shared__ float4 part_s[16][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t e = ((int64_t)blockIdx.x * 64 + lane) * 4;  // n is a multiple of 256 (nH * 64 * 64)
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
    for (int s = w; s < parts; s += 16) {
        const float4 v = *(const float4*)(src + (int64_t)s * n + e);
        acc.x += v.x;
        acc.y += v.y;
        acc.z += v.z;
        acc.w += v.w;
    }
    part_s[w][lane] = acc;
    __syncthreads();
    if (w == 0) {
        float4 t = overwrite ? make_float4(0.f, 0.f, 0.f, 0.f) : *(const float4*)(dst + e);
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const float4 v = part_s[k][lane];
            t.x += v.x;
            t.y += v.y;
            t.z += v.z;
            t.w += v.w;
        }
        *(float4*)(dst + e) = t;
    }
}

// dhead_scale[h] += sum over slots and the head's two waves of part[slot][h][qt]
__global__ void reduce_scale_partials_kernel(const float* __restrict__ src, float* __restrict__ dst, int slots, int nH, int overwrite) {
    const int h = threadIdx.x;
    if (h >= nH) return;
    float acc = 0.f;
    for (int s2 = 0; s2 < slots; ++s2) acc += src[((int64_t)s2 * nH + h) * 2] + src[((int64_t)s2 * nH + h) * 2 + 1];
    dst[h] = overwrite ? acc : dst[h] + acc;
}

// Persistent grid = resident workgroups, counted PER XCD: the kernels map the head groups of a window slot onto one XCD
// (32 CUs x 8 wavefronts at <= 256 VGPRs), so an XCD holds floor(capacity / groups) slots; one workgroup beyond that would
// run as a second round and double the launch time.
int persistent_slots(const AttnParams& p, int groups, int waves_per_wg) {
    const int64_t windows = (int64_t)p.B * (p.N / kWs);
    const int capacity = usable_cus_per_xcd() * (8 / waves_per_wg);
    int64_t per_xcd = capacity / groups;
    if (per_xcd < 1) per_xcd = 1;
    int64_t slots = 8 * per_xcd;
    if (slots > windows) slots = windows;
    return (int)(slots < 1 ? 1 : slots);
}
int bwd_slots(const AttnParams& p, int hg) { return persistent_slots(p, p.nH / hg, 2 * hg); }  // two wavefronts per head
int fwd_slots(const AttnParams& p, int hg) { return persistent_slots(p, p.nH / hg, hg); }

// Heads per workgroup.  Forward (one wave per head): 4 where the head count allows (round 2: 2 / 1 heads per workgroup 3-7 % / 20 %
// slower).  Backward (two waves per head): round 2 measured pairs ahead of four heads on the 5-barrier kernel (8-wave barriers);
// with the 3-barrier kernel of round 4 FOUR heads per workgroup (8 waves, 137 KB of LDS, one workgroup per CU, 256-byte row
// segments) win at every stage of HEAL-SWIN-B (profiles/archive_r01_r04/r04_attn_bwd_hg4_ab.txt, same box: 570 -> 515, 290 -> 260, 163 -> 150,
// 96 -> 93 us).
int pick_head_group_bwd(const AttnParams& p) {
    const int nH = p.nH;
    // four heads per workgroup pay on the large launches only (one [B, N, C] tensor >= 64 MB: stages 0-2 of HEAL-SWIN-B at nside
    // 256); on the small ones (HEAL-SWIN-T: stage 2 at nside 256 88 vs 93 us, at nside 128 33 vs 44 us) two workgroups of four
    // waves per CU hide each other's prologue and barriers better
    if (nH % 4 == 0 && (int64_t)p.B * p.N * p.C * 2 >= (64ll << 20)) return 4;
    // 33 KB of LDS per head.  Three heads (stage 0 of the T model: 192-byte rows) go together -- whole rows per workgroup instead of
    // 64-byte slices whose line neighbours come from L2 (PMC: 1.24 x the algorithmic traffic with one head per workgroup,
    // profiles/r05_attn_pmc_T256_vs_D256.txt): T @ 128 stage 0 122 -> 92 us, T @ 256 462 -> 450 (profiles/archive_r01_r04/r04_attn_hg_T.txt; six
    // heads stay in pairs: 214 vs 238 us in threes)
    if (nH == 3) return 3;
    return nH % 2 == 0 ? 2 : 1;
}

template <int HG, bool DROP, bool COS>
int launch_bwd(const AttnParams& p, float* workspace, hipStream_t stream) {
    const LdsLayoutBwd L(HG);
    auto kern = attn_bwd_mfma_kernel<HG, DROP, COS>;
    static bool configured = false;
    if (!configured) {
        HS_HIP_CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, L.total));
        configured = true;
    }
    const int groups = p.nH / HG, slots = bwd_slots(p, HG);
    float* dbias_part = p.dbias ? workspace : nullptr;
    float* dscale_part = p.dhead_scale ? workspace + (int64_t)slots * p.nH * kWs * kWs : nullptr;
    const unsigned grid = 8u * (unsigned)((slots + 7) / 8) * (unsigned)groups;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(128 * HG), L.total, stream, p, dbias_part, dscale_part, slots, groups);
    HS_LAUNCH_CHECK("attn_bwd_mfma");
    if (dbias_part) {
        const int64_t n = (int64_t)p.nH * kWs * kWs;
        hipLaunchKernelGGL(reduce_partials_kernel, dim3((unsigned)(n / 256)), dim3(1024), 0, stream, dbias_part, p.dbias, slots, n, (p.flags & HS_ATTN_OVERWRITE_GRADS) ? 1 : 0);
        HS_LAUNCH_CHECK("reduce dbias partials");
    }
    if (dscale_part) {
        hipLaunchKernelGGL(reduce_scale_partials_kernel, dim3(1), dim3(256), 0, stream, dscale_part, p.dhead_scale, slots, p.nH, (p.flags & HS_ATTN_OVERWRITE_GRADS) ? 1 : 0);
        HS_LAUNCH_CHECK("reduce dscale partials");
    }
    return HS_OK;
}

int pick_head_group(const AttnParams& p) {
    const int nH = p.nH;
    // eight heads per workgroup (512-byte row segments, one workgroup of 8 waves per CU) on the large launches: with the
    // 2-barrier kernel of round 4 stages 1 / 2 of HEAL-SWIN-B run 177 -> 166 / 107 -> 99 us (profiles/archive_r01_r04/r04_attn_fwd_hg8_ab.txt;
    // round 3 had measured the opposite on the 3-barrier kernel); the small stage 3 (50 MB) loses 10 %
    if (nH % 8 == 0 && (int64_t)p.B * p.N * p.C * 2 >= (64ll << 20)) return 8;
    if (nH % 4 == 0) return 4;
    if (nH % 3 == 0) return 3;
    if (nH % 2 == 0) return 2;
    return 1;
}

template <int HG, bool DROP>
int launch_fwd(const AttnParams& p, hipStream_t stream) {
    const LdsLayoutFwd L(HG);
    auto kern = attn_fwd_mfma_kernel<HG, DROP>;
    static bool configured = false;
    if (!configured) {
        HS_HIP_CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, L.total));
        configured = true;
    }
    const int groups = p.nH / HG;
    const int slots = fwd_slots(p, HG);
    const unsigned grid = 8u * (unsigned)((slots + 7) / 8) * (unsigned)groups;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * HG), L.total, stream, p, slots, groups);
    HS_LAUNCH_CHECK("attn_fwd_mfma");
    return HS_OK;
}

}  // namespace

bool attn_mfma_supported(const AttnParams& p, int dtype) {
    // 16-byte vector access needs 8-element aligned columns: C % 8 == 0 holds since C = 32 * nH
    // (per-image buffer descriptors: one image of qkv must stay below 2 GiB)
    return dtype == HS_BF16 && p.Ws == kWs && p.hd == kHd && p.N * 3 * (int64_t)p.C * 2 < (1ll << 31);
}

int64_t attn_bwd_mfma_workspace_floats(const AttnParams& p) {
    const int hg = pick_head_group_bwd(p);
    return (int64_t)bwd_slots(p, hg) * p.nH * (kWs * kWs + 2);
}

int launch_attn_fwd_mfma(const AttnParams& p, hipStream_t stream) {
    const bool drop = p.drop_p > 0.f;
    switch (pick_head_group(p)) {
        case 8: return drop ? launch_fwd<8, true>(p, stream) : launch_fwd<8, false>(p, stream);
        case 4: return drop ? launch_fwd<4, true>(p, stream) : launch_fwd<4, false>(p, stream);
        case 3: return drop ? launch_fwd<3, true>(p, stream) : launch_fwd<3, false>(p, stream);
        case 2: return drop ? launch_fwd<2, true>(p, stream) : launch_fwd<2, false>(p, stream);
        default: return drop ? launch_fwd<1, true>(p, stream) : launch_fwd<1, false>(p, stream);
    }
}

template <int HG>
int launch_bwd_hg(const AttnParams& p, float* workspace, hipStream_t stream) {
    const bool drop = p.drop_p > 0.f, cos = (p.flags & HS_ATTN_COSINE) != 0;
    if (cos) return drop ? launch_bwd<HG, true, true>(p, workspace, stream) : launch_bwd<HG, false, true>(p, workspace, stream);
    return drop ? launch_bwd<HG, true, false>(p, workspace, stream) : launch_bwd<HG, false, false>(p, workspace, stream);
}

int launch_attn_bwd_mfma(const AttnParams& p, float* workspace, hipStream_t stream) {
    if (!workspace) return fail(HS_ERR_INVALID_ARG, "the MFMA backward needs a workspace (hs_window_attn_bwd_workspace)");
    switch (pick_head_group_bwd(p)) {
        case 4: return launch_bwd_hg<4>(p, workspace, stream);
        case 3: return launch_bwd_hg<3>(p, workspace, stream);
        case 2: return launch_bwd_hg<2>(p, workspace, stream);
        default: return launch_bwd_hg<1>(p, workspace, stream);
    }
}

}  // namespace hs
