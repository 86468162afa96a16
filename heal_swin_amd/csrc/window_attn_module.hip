// Fused WindowAttention MODULE forward for the stages whose weights fit the LDS (C = 96 / 128: stage 0 of HEAL-SWIN-T / -B):
//
//     out = [x +] proj( window_attention( qkv( [LayerNorm](x) ) ) )        one launch, x read once, out written once
//
// i.e. reference WindowAttention.forward (models_torch/swin_hp_transformer.py:124-174) together with the shift / window
// partition / reverse / shift back around it (:319-330) and, optionally, the block's norm1 in front (:315, v1 placement)
// and its residual add behind (:316).  qkv [B, N, 3C], the attention output and the probabilities never exist in HBM:
// per token 2 C * 2 B of traffic instead of 10 C * 2 B for  qkv GEMM -> hs_window_attn_fwd -> proj GEMM  (SURVEY 8d).
// Inference form: nothing is saved for a backward.  TRAINING form (TRAIN, round 4: hs_window_attn_module_fwd_train): the same
// launch also writes what the composed backward (hs_layernorm_bwd, hs_linear_wgrad, hs_window_attn_bwd, hs_gemm_nt) reads --
// LayerNorm(x) and its row statistics, qkv [B, N, 3C], the attention output [B, N, C] and the score rows' log-sum-exp --
// in natural token order, straight from the registers / LDS tiles they live in: per token 2 C (x in, out) + 5 C (saved) * 2 B
// instead of the 13 C * 2 B of LayerNorm -> qkv GEMM -> hs_window_attn_fwd -> proj GEMM (+ residual), none of it re-read.
//
// Workgroup = nH wavefronts (one per SIMD, up to 512 registers each), persistent, one 64-token window at a time:
//   * the bf16 weights of qkv ([3C, C], 96 KB at C = 128) stay in LDS for the whole launch, proj's ([C, C]) in REGISTERS
//     (wave w owns output channels 32 w .. 32 w + 31: 8 k-steps x 4 registers), the relative-position bias of head w too;
//   * x tiles arrive by buffer_load ... lds (row gather through the shift table / roll), the next window's in flight under
//     the current window's arithmetic; LayerNorm, if requested, is applied in place on the tile;
//   * wave h = head h, and every contraction is oriented so that its result is, after bf16 packing, DIRECTLY the MFMA operand
//     of the next one -- no LDS round trip between them (v_mfma_f32_32x32x16_bf16; lane = l, half = l / 32):
//         q^T[d][tok] = Wq x^T   (A = Wq rows d,  B = x rows tok)     lane = token, registers = 16 of the 32 features
//         k^T[d][tok] = Wk x^T                                        (same layout)
//         v  [tok][d] = x Wv^T   (A = x rows tok, B = Wv rows d)      lane = feature, registers = 16 of the 32 tokens
//         S^T[key][q] = k q^T    A = k^T registers (lane = key, 8 features per step), B = q^T registers (lane = query);
//                                the feature ORDER inside a step is whatever the accumulator layout gives -- it is the
//                                same for q and k, and a contraction does not care
//         softmax over the keys of each query: in registers + one lane^32 exchange (as hs_window_attn_fwd)
//         O^T[d][q]   = v^T P^T  A = v registers (lane = feature, 8 keys per step), B = P registers (lane = query)
//         Y^T[n][tok] = Wp O^T   A = Wp registers, B = O rows from an LDS tile the head waves fill (8-byte writes)
//     80 MFMAs per head and window: 48 (qkv) + 8 (scores) + 8 (P V) + 16 (proj);
//   * Y^T has lane = token and 4 consecutive channels per register group: staged through the (dead) x tile and stored as
//     whole 2C-byte token rows to out[token] (the scatter half of the shift).
// All LDS traffic of the main phases is inline asm: a compiler-visible LDS access beside the DMA queue would be preceded by
// s_waitcnt vmcnt(0) and serialise the next window's loads behind every phase.
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "window_attn.h"

namespace hs {
namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef unsigned int u32x2v __attribute__((__vector_size__(8)));
typedef unsigned int u32x4v __attribute__((__vector_size__(16)));
typedef __attribute__((address_space(3))) void lds_void;

constexpr int kWs = 64;
constexpr int kRowB = 256;  // bytes per LDS tile row (C <= 128 bf16; rows of C = 96 are padded)
constexpr float kLog2e = 1.4426950408889634f;
constexpr float kNormEps = 1e-12f;
// 1 / max(|x|, eps) from the squared norm: v_rsq_f32 (1 ulp) + a clamp instead of the correctly rounded sqrt and division hipcc
// expands to ~20 instructions each -- four of them per row block were half of what cosine attention added to the kernels' VALU
// count (profiles/r05_attn_pmc_T256_vs_D256.txt: 66 vs 37 VALU per MFMA in the forward); results are bf16 rows
__device__ __forceinline__ float inv_norm(float sumsq) { return fminf(__builtin_amdgcn_rsqf(sumsq), 1.f / kNormEps); }
constexpr float kMaskLog2 = -100.f * kLog2e;
constexpr float kLnEps = 1e-5f;
__device__ constexpr uint32_t kOob = 0x7FFFFF00u;

struct ModParams {
    const uint16_t* x;
    uint16_t* out;
    const uint16_t* qkv_w;   // [3C, C] bf16
    const float* qkv_b;      // [3C] or null
    const uint16_t* proj_w;  // [C, C] bf16
    const float* proj_b;     // [C] or null
    const float* ln_g;       // [C] or null: LayerNorm(x) in front
    const float* ln_b;
    const float* bias;        // [nH, 64, 64] or null
    const float* head_scale;  // [nH]
    const int32_t* idx;
    int64_t roll;
    const uint8_t* labels;
    int B;
    int64_t N;
    int slots;
    unsigned flags;
    int dma_mode;  // where the next window's x tile is requested: 0 (shipped) = in front of the LayerNorm, 1 = behind the
                   // second k-step of the qkv product, 2 = one piece behind each of its first k-steps
    // training form (null in the inference form), natural token order
    uint16_t* xn_out;   // [B, N, C]   LayerNorm(x): input of the qkv product (its weight gradient reads it)
    uint16_t* qkv_out;  // [B, N, 3C]  as the qkv Linear would have written it
    uint16_t* o_out;    // [B, N, C]   attention output: input of the proj product
    float* mean_out;    // [B, N]      LayerNorm statistics
    float* rstd_out;
    float* lse_out;     // [B, nH, N]  log-sum-exp of every score row, by SHIFTED position (as hs_window_attn_fwd)
    // optional second LayerNorm BEHIND the residual add (the block's norm2, ref :337): n2 = LayerNorm(out), with its statistics
    const float* ln2_g;
    const float* ln2_b;
    uint16_t* n2_out;   // [B, N, C]
    float* mean2_out;   // [B, N]
    float* rstd2_out;
#ifdef HS_MOD_TRACE
    unsigned long long* trace;  // measurement build: shader-clock stamps [wave][window < 8][phase < 16] of workgroup 0
#endif
};
constexpr unsigned kFlagResidual = 4u;  // out = x + module(x)   (HS_ATTN_RESIDUAL)

__device__ __forceinline__ uint32_t swz(int row, int chunk) { return row * kRowB + ((chunk ^ (row & 15)) << 4); }

__device__ __forceinline__ u32x4 ld128(uint32_t addr) {
    u32x4 v;
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr));
    return v;
}
__device__ __forceinline__ void st64(uint32_t addr, uint32_t a, uint32_t b) {
    const u32x2v v = {a, b};
    asm volatile("ds_write_b64 %0, %1" ::"v"(addr), "v"(v) : "memory");
}
__device__ __forceinline__ bf16x8 as_frag(const u32x4& v) { return __builtin_bit_cast(bf16x8, v); }
// 8 consecutive accumulator registers -> one bf16 MFMA operand
__device__ __forceinline__ bf16x8 pack8(const f32x16& a, int r0) {
    const u32x4 v = {pack_bf16x2(a[r0], a[r0 + 1]), pack_bf16x2(a[r0 + 2], a[r0 + 3]), pack_bf16x2(a[r0 + 4], a[r0 + 5]),
                     pack_bf16x2(a[r0 + 6], a[r0 + 7])};
    return __builtin_bit_cast(bf16x8, v);
}

// 16 accumulator values of a transposed output tile (lane = token, register r = feature (r&3) + 8*(r>>2) + 4*half) -> two
// 16-byte pieces of the token's 64-byte head slice: lanes < 32 hold bytes [0,16) and [32,48), lanes >= 32 bytes [16,32) and [48,64)
// (v_permlane32_swap pairs the 8-byte pieces of the two lane halves)
__device__ __forceinline__ void swap_rows_t(const uint32_t (&packed)[8], u32x4& p0, u32x4& p1) {  // packed[i] = registers 2i, 2i+1
    uint32_t w[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) w[i] = packed[i];
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
__device__ __forceinline__ void pack_rows_t(const f32x16& v, u32x4& p0, u32x4& p1) {
    uint32_t w[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) w[i] = pack_bf16x2(v[2 * i], v[2 * i + 1]);
    swap_rows_t(w, p0, p1);
}
constexpr float kLn2 = 0.6931471805599453f;

template <int NH, bool COS, bool TRAIN>
__global__ void __launch_bounds__(NH * 64, 1) attn_module_fwd_kernel(ModParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int C = 32 * NH, NT = NH * 64, KS = C / 16, NCH = C / 8;  // channels, threads, 16-deep k-steps, 16-byte chunks per row
    constexpr int W_OFF = 0, X_OFF = 3 * C * kRowB, O_OFF = X_OFF + 2 * kWs * kRowB, M_OFF = O_OFF + kWs * kRowB;
    constexpr int XP = (16 + NH - 1) / NH;  // x-tile DMA pieces (1 KB = 4 rows) per wave
    // + two 256-B label patches (64 bytes used) + fp32 parameter block: LayerNorm gamma | beta, qkv bias (q | k rows), proj bias
    constexpr int P_OFF = M_OFF + 2 * 256, P_LNG = 0, P_LNB = 128, P_BQ = 256, P_BK = 384, P_BP = 512, P_BV = 640, P_LN2G = 768,
                  P_LN2B = 896;  // float offsets, 128 each
    // + (training form) the natural-order token of each of the 64 window rows, current / next window: two 256-byte tables
    constexpr int T_OFF = P_OFF + 8 * 128 * 4;
    __shared__ __attribute__((aligned(16))) unsigned char smem[T_OFF + 2 * 256];
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;

    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // = head index h = proj channel block
    const int64_t N = p.N;
    const int nW = (int)(N / kWs);
    const int64_t total_windows = (int64_t)p.B * nW;
    const bool residual = (p.flags & kFlagResidual) != 0;
    const bool has_ln = p.ln_g != nullptr;
    if ((int64_t)blockIdx.x >= total_windows) return;

    // ---------------------------------------------------------------- one-off: weights, biases, LayerNorm parameters
    {
        const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)p.qkv_w, 0, 3 * C * C * 2, 0x00020000);
        for (int pc = wave; pc < 3 * C * 16 / 64; pc += NH) {  // 1-KB pieces: 4 weight rows each
            const int q = pc * 64 + lane, row = q >> 4, chunk = (q & 15) ^ (row & 15);
            const uint32_t voff = chunk < NCH ? (uint32_t)(row * C * 2 + chunk * 16) : kOob;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_void*)(smem + W_OFF + pc * 1024), 16, voff, 0, 0, 0);
        }
    }
    // small fp32 parameters -> LDS (read back per window with inline-asm LDS reads: registers are the scarce resource here)
    constexpr int CPQ = NCH / 4;  // LayerNorm: thread -> quarter row = CPQ chunks of 8 channels: 4 (C = 128) or 3 (C = 96)
    {
        float* ps = (float*)(smem + P_OFF);
        for (int i = tid; i < C; i += NT) {
            ps[P_LNG + i] = has_ln ? p.ln_g[i] : 1.f;
            ps[P_LNB + i] = has_ln ? p.ln_b[i] : 0.f;
            ps[P_BQ + i] = p.qkv_b ? p.qkv_b[i] : 0.f;
            ps[P_BK + i] = p.qkv_b ? p.qkv_b[C + i] : 0.f;
            ps[P_BP + i] = p.proj_b ? p.proj_b[i] : 0.f;
            ps[P_BV + i] = p.qkv_b ? p.qkv_b[2 * C + i] : 0.f;
            ps[P_LN2G + i] = p.ln2_g ? p.ln2_g[i] : 1.f;
            ps[P_LN2B + i] = p.ln2_b ? p.ln2_b[i] : 0.f;
        }
    }
    const uint32_t pbase = lds0 + P_OFF;
    // proj weights of this wave's 32 output channels, as A operands: lane (row n = 32 w + l31) holds k = 16 ks + 8 half ..
    bf16x8 wp[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) wp[ks] = *(const bf16x8*)(p.proj_w + (int64_t)(32 * wave + l31) * C + 16 * ks + 8 * half);
    const float bv = p.qkv_b ? p.qkv_b[2 * C + 32 * wave + l31] : 0.f;  // v bias: column d = l31 of the v accumulators
    // relative-position bias of this head (x log2 e) in the S^T layout: tile (kt, qt), register r: query qt*32 + l31, key kt*32 + d(r)
    float biasr[2][2][16];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int qt = 0; qt < 2; ++qt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half, qq = qt * 32 + l31;
                biasr[kt][qt][r] = p.bias ? p.bias[((int64_t)wave * kWs + qq) * kWs + key] * kLog2e : 0.f;
            }
    const float hscale = p.head_scale[wave];

    // ---------------------------------------------------------------- x-tile pieces of this wave: (row, logical chunk), token offsets
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
        (void*)p.x, 0, (int)(((int64_t)p.B * N * C * 2) > 0x7FFFFE00ll ? 0x7FFFFE00ll : (int64_t)p.B * N * C * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rl = __builtin_amdgcn_make_buffer_rsrc((void*)p.labels, 0, p.labels ? (int)N : 0, 0x00020000);
    // piece j of this wave: LDS position q = (wave + NH j) * 64 + lane -> (tile row q / 16, logical chunk (q % 16) ^ (row % 16)).
    // Recomputed from an opaque copy of the lane index that is refreshed once per window (two shifts and an xor per use): kept
    // in registers over the window loop -- with everything derived from them, hoisted -- they were spilled, and a reload at the
    // loop top waits for the previous window's stores
    int lane_o = lane;
    auto prow_of = [&](int j) { return ((wave + NH * j) * 64 + lane_o) >> 4; };
    auto pchunk_of = [&](int j) { return (lane_o & 15) ^ (prow_of(j) & 15); };
    uint32_t tok[XP], tok_next[XP];  // token (row of this launch's x) of this lane's piece rows: current / next window (< 2^30: the chunk limit)
    // a window = (image b, window r of that image); the loop advances the pair by `slots` windows without a division
    auto tokens_of = [&](int b, int r, uint32_t (&t)[XP]) {
        const int64_t j0 = (int64_t)r * kWs;
#pragma unroll
        for (int j = 0; j < XP; ++j) {
            const int64_t js = j0 + (prow_of(j) & 63);  // shifted position -> natural-order token (gather = scatter map)
            int64_t src;
            if (p.idx) src = p.idx[js];
            else {
                src = js + p.roll;
                if (src >= N) src -= N;
            }
            t[j] = (uint32_t)((int64_t)b * N + src);
        }
    };
    auto issue_x = [&](const uint32_t (&t)[XP], int r, int buf, int only = -1) {  // only >= 0: piece `only` (and the labels with piece 0)
#pragma unroll
        for (int j = 0; j < XP; ++j) {
            if (only >= 0 && j != only) continue;
            const int pc = wave + NH * j;
            if (pc < 16) {
                const uint32_t voff = pchunk_of(j) < NCH ? (uint32_t)(t[j] * (C * 2) + pchunk_of(j) * 16) : kOob;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_void*)(smem + X_OFF + buf * (kWs * kRowB) + pc * 1024), 16, voff, 0, 0, 0);
            }
        }
        if (only <= 0 && wave == 0 && p.labels) {  // 64 label bytes = 16 dwords; the other lanes read past the descriptor (zeros)
            const uint32_t voff = lane < 16 ? (uint32_t)(r * kWs + lane * 4) : kOob;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rl, (lds_void*)(smem + M_OFF + buf * 256), 4, voff, 0, 0, 0);
        }
    };

    auto note_tokens = [&](int b, int r, int buf) {  // (training form) row -> token table of window (b, r)
        if (tid < kWs) {
            const int64_t js = (int64_t)r * kWs + tid;
            int64_t src;
            if (p.idx) src = p.idx[js];
            else {
                src = js + p.roll;
                if (src >= N) src -= N;
            }
            const uint32_t t = (uint32_t)((int64_t)b * N + src);
            asm volatile("ds_write_b32 %0, %1" ::"v"(lds0 + T_OFF + buf * 256 + tid * 4), "v"(t) : "memory");
        }
    };
    // whole 2C-byte token rows of an LDS tile (x / O tile image) -> dst[token]: the scatter half of the shift
    auto store_rows = [&](uint32_t tile_base, uint16_t* dst) {
        u32x4 rows[XP];
#pragma unroll
        for (int j = 0; j < XP; ++j)
            asm volatile("ds_read_b128 %0, %1" : "=v"(rows[j]) : "v"(tile_base + (uint32_t)((wave + NH * j) * 1024 + lane * 16)));
#pragma unroll
        for (int j = 0; j < XP; ++j) {
            asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(rows[j]) : "n"(XP - 1 - j));
            if (wave + NH * j < 16 && pchunk_of(j) < NCH) *(u32x4*)(dst + (int64_t)tok[j] * C + pchunk_of(j) * 8) = rows[j];
        }
    };

#ifdef HS_MOD_TRACE
    int tr_w = 0;
    auto TR = [&](int ph) {
        if (p.trace && blockIdx.x == 0 && tr_w < 8 && lane == 0) p.trace[(wave * 8 + tr_w) * 16 + ph] = clock64();
    };
#else
    auto TR = [&](int) {};
#endif
    int64_t wi = blockIdx.x;
    int b_c = (int)((uint32_t)blockIdx.x / (uint32_t)nW), r_c = (int)((uint32_t)blockIdx.x - (uint32_t)b_c * (uint32_t)nW);
    tokens_of(b_c, r_c, tok);
    issue_x(tok, r_c, 0);
    if constexpr (TRAIN) note_tokens(b_c, r_c, 0);
    int cur = 0;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the weights and the first window's x rows have landed
    const uint32_t wbase = lds0 + W_OFF, obase = lds0 + O_OFF;

    for (; wi < total_windows; wi += p.slots) {
        asm volatile("" : "+v"(lane_o));
        const bool more = wi + p.slots < total_windows;
        int b_n = b_c, r_n = r_c + p.slots;
        while (r_n >= nW) {
            r_n -= nW;
            ++b_n;
        }
        if (more) tokens_of(b_n, r_n, tok_next);
        // (this wave's pieces of the window's x tile were waited for in front of the previous window's output stores -- not here,
        // where a vmcnt(0) would also wait for those stores' acknowledgements: 1-2 k cycles per window with nothing else to run)
        if constexpr (TRAIN) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (the first window's token table)
        __builtin_amdgcn_s_barrier();
        TR(0);
        const uint32_t xbase = lds0 + X_OFF + cur * (kWs * kRowB), tbase = lds0 + T_OFF + cur * 256;
        if constexpr (TRAIN) {
            if (more) note_tokens(b_n, r_n, cur ^ 1);  // read from the next window's first barrier on
        }
        // region labels of this window (fetched with its x tile into label patch `cur`): 16 words, every lane reads all
        // (the words are read again where a cut window needs them: 16 registers held across every phase cost the training form spills)
        bool mixed = false;
        auto read_labels = [&](uint32_t (&labw)[16]) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(labw[i]) : "v"(lds0 + M_OFF + cur * 256), "n"(4 * i));
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(labw[0]), "+v"(labw[1]), "+v"(labw[2]), "+v"(labw[3]), "+v"(labw[4]), "+v"(labw[5]), "+v"(labw[6]),
                           "+v"(labw[7]), "+v"(labw[8]), "+v"(labw[9]), "+v"(labw[10]), "+v"(labw[11]), "+v"(labw[12]),
                           "+v"(labw[13]), "+v"(labw[14]), "+v"(labw[15]));
        };
        if (p.labels) {
            uint32_t labw[16];
            read_labels(labw);
            const uint32_t first = (labw[0] & 0xffu) * 0x01010101u;
#pragma unroll
            for (int i = 0; i < 16; ++i) mixed |= labw[i] != first;
        }
        // (requested here, as early as the buffer is free.  Moving the ~100 issue cycles per LDS-DMA piece into the shadow of the qkv
        // product's MFMAs -- dma_mode 1 / 2 -- shortens this phase by 400 cycles in the in-kernel timeline and LENGTHENS the launch by
        // 3-6 %: the tile then arrives later than the stores of this window start to compete with it)
        if (more && p.dma_mode == 0) issue_x(tok_next, r_n, cur ^ 1);

        // ------------------------------------------------------------ optional LayerNorm of the 64 rows, in place
        if (has_ln) {
            // thread -> (row tid / 4 [+ NT / 4 per pass], quarter tid % 4 = CPQ chunks of 8 channels)
            // (opaque copy of the thread index: the row / chunk / parameter addresses below are loop-invariant, and hoisted out of the
            // window loop they were spilled -- reloaded here behind the next window's DMA, i.e. behind a full vmcnt(0))
            int tl = tid;
            asm volatile("" : "+v"(tl));
            for (int row = tl >> 2; row < kWs; row += NT / 4) {
                const int qd = tl & 3;
                // the row quarter and its gamma / beta are requested together: one LDS round trip for the whole phase
                u32x4 v[CPQ], gw[CPQ][2], bw[CPQ][2];
                uint32_t rtok = 0;  // (training form) token of this row; the oldest LDS read of the phase: returned before v[0]
                if constexpr (TRAIN) asm volatile("ds_read_b32 %0, %1" : "=v"(rtok) : "v"(tbase + row * 4));
#pragma unroll
                for (int c = 0; c < CPQ; ++c) v[c] = ld128(xbase + swz(row, qd * CPQ + c));
#pragma unroll
                for (int c = 0; c < CPQ; ++c) {
                    const uint32_t ga = pbase + (P_LNG + (qd * CPQ + c) * 8) * 4, ba = pbase + (P_LNB + (qd * CPQ + c) * 8) * 4;
                    gw[c][0] = ld128(ga);
                    gw[c][1] = ld128(ga + 16);
                    bw[c][0] = ld128(ba);
                    bw[c][1] = ld128(ba + 16);
                }
                if constexpr (CPQ == 4)
                    asm volatile("s_waitcnt lgkmcnt(15)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]));  // (4-bit counter: 15 of the 16 parameter reads may be pending)
                else
                    asm volatile("s_waitcnt lgkmcnt(12)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]));
                float s1 = 0.f, s2 = 0.f, f[CPQ][8];
#pragma unroll
                for (int c = 0; c < CPQ; ++c)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        f[c][2 * e] = __uint_as_float(v[c][e] << 16);
                        f[c][2 * e + 1] = __uint_as_float(v[c][e] & 0xffff0000u);
                        s1 += f[c][2 * e] + f[c][2 * e + 1];
                    }
                // sums over the 4 lanes of a row: DPP quad permutes (1 VALU each) instead of LDS-routed shuffles
                s1 += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, s1), 0xB1, 0xF, 0xF, true));
                s1 += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, s1), 0x4E, 0xF, 0xF, true));
                const float mean = s1 * (1.f / C);
#pragma unroll
                for (int c = 0; c < CPQ; ++c)
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        f[c][e] -= mean;
                        s2 += f[c][e] * f[c][e];
                    }
                s2 += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, s2), 0xB1, 0xF, 0xF, true));
                s2 += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, s2), 0x4E, 0xF, 0xF, true));
                const float rstd = rsqrtf(s2 * (1.f / C) + kLnEps);
                if constexpr (TRAIN) {
                    asm volatile("" : "+v"(rtok));
                    if (qd == 0) {
                        p.mean_out[rtok] = mean;
                        p.rstd_out[rtok] = rstd;
                    }
                }
#pragma unroll
                for (int c = 0; c < CPQ; ++c) {
                    asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(gw[c][0]), "+v"(gw[c][1]), "+v"(bw[c][0]), "+v"(bw[c][1]) : "n"(4 * (CPQ - 1 - c)));
                    u32x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float ga0 = __uint_as_float(gw[c][e >> 1][(2 * e) & 3]), ga1 = __uint_as_float(gw[c][e >> 1][(2 * e + 1) & 3]);
                        const float be0 = __uint_as_float(bw[c][e >> 1][(2 * e) & 3]), be1 = __uint_as_float(bw[c][e >> 1][(2 * e + 1) & 3]);
                        o[e] = pack_bf16x2(fmaf(f[c][2 * e] * rstd, ga0, be0), fmaf(f[c][2 * e + 1] * rstd, ga1, be1));
                    }
                    asm volatile("ds_write_b128 %0, %1" ::"v"(xbase + swz(row, qd * CPQ + c)), "v"(o) : "memory");
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        TR(1);
        if constexpr (TRAIN) {
            if (has_ln) store_rows(xbase, p.xn_out);  // LayerNorm(x), whole rows from the tile the qkv product reads
        }
        TR(2);

        // ------------------------------------------------------------ q^T, k^T, v of this head: 6 accumulators, KS k-steps
        f32x16 aq[2], ak[2], av[2];
        uint32_t ltok[4] = {0u, 0u, 0u, 0u};  // (training form) tokens of the rows lane / 4 + 16 i this lane stores qkv pieces of
        if constexpr (TRAIN) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(ltok[i]) : "v"(tbase + (lane >> 2) * 4), "n"(64 * i));
        }
        {   // accumulators start at the bias: q^T / k^T rows d = 8 g + 4 half + 0..3 (group g = r / 4), v column d = l31
            u32x4 bqv[4], bkv[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                bqv[g] = ld128(pbase + (P_BQ + 32 * wave + 8 * g + 4 * half) * 4);
                bkv[g] = ld128(pbase + (P_BK + 32 * wave + 8 * g + 4 * half) * 4);
            }
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(bqv[0]), "+v"(bqv[1]), "+v"(bqv[2]), "+v"(bqv[3]), "+v"(bkv[0]), "+v"(bkv[1]), "+v"(bkv[2]), "+v"(bkv[3]));
            if constexpr (TRAIN) asm volatile("" : "+v"(ltok[0]), "+v"(ltok[1]), "+v"(ltok[2]), "+v"(ltok[3]));
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    aq[t][r] = __uint_as_float(bqv[r >> 2][r & 3]);
                    ak[t][r] = __uint_as_float(bkv[r >> 2][r & 3]);
                    av[t][r] = bv;
                }
        }
        {
            const uint32_t xa0 = xbase + l31 * kRowB, xa1 = xbase + (32 + l31) * kRowB;
            const uint32_t wq = wbase + (32 * wave + l31) * kRowB, wk = wq + C * kRowB, wv = wk + C * kRowB;
            const int sx = l31 & 15;  // (row & 15) of every operand row of this lane: rows differ by multiples of 32 and 16 | C
            u32x4 fx[2][2], fw[2][3];
            auto reads = [&](int ks, int set) {
                const uint32_t co = (uint32_t)(((2 * ks + half) ^ sx) << 4);
                fx[set][0] = ld128(xa0 + co);
                fx[set][1] = ld128(xa1 + co);
                fw[set][0] = ld128(wq + co);
                fw[set][1] = ld128(wk + co);
                fw[set][2] = ld128(wv + co);
            };
            reads(0, 0);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int set = ks & 1;
                if (ks + 1 < KS) {
                    reads(ks + 1, set ^ 1);
                    asm volatile("s_waitcnt lgkmcnt(5)" : "+v"(fx[set][0]), "+v"(fx[set][1]), "+v"(fw[set][0]), "+v"(fw[set][1]), "+v"(fw[set][2]));
                } else {
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fx[set][0]), "+v"(fx[set][1]), "+v"(fw[set][0]), "+v"(fw[set][1]), "+v"(fw[set][2]));
                }
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    aq[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_frag(fw[set][0]), as_frag(fx[set][t]), aq[t], 0, 0, 0);
                    ak[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_frag(fw[set][1]), as_frag(fx[set][t]), ak[t], 0, 0, 0);
                    av[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_frag(fx[set][t]), as_frag(fw[set][2]), av[t], 0, 0, 0);
                }
                if (more) {  // in flight during everything below
                    if (p.dma_mode == 1 && ks == 1) issue_x(tok_next, r_n, cur ^ 1);
                    if (p.dma_mode == 2 && ks < XP) issue_x(tok_next, r_n, cur ^ 1, ks);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        TR(3);
        // ---- bf16 rounding of q^T, k^T, v: ONE packing serves the MFMA operands, the cosine norms and (training form) the stored rows;
        // the fp32 accumulators are dead behind it.  Word i of a tile = registers 2i, 2i+1.
        uint32_t qw[2][8], kw[2][8];
        bf16x8 qf[2][2], kf[2][2], vf[2][2];  // [token tile][8-register step]
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                qw[t][i] = pack_bf16x2(aq[t][2 * i], aq[t][2 * i + 1]);
                kw[t][i] = pack_bf16x2(ak[t][2 * i], ak[t][2 * i + 1]);
            }
#pragma unroll
            for (int c = 0; c < 2; ++c) vf[t][c] = pack8(av[t], 8 * c);
        }
        if constexpr (TRAIN) {
            // The q, k and v rows as the qkv Linear would have stored them.  Row-per-lane 16-byte stores (64 rows, 64 partial
            // lines per instruction) cost this phase 2100 + 1500 cycles of a 19 000-cycle window (in-kernel timeline, -DHS_MOD_TRACE):
            // the L1 -> L2 path takes a transaction per lane.  Each tensor therefore passes through the wave's OWN 64-byte column
            // block of the (idle) O tile -- written in the accumulator layout, read back with four adjacent lanes on the four
            // 16-byte units of one token's head slice -- so an instruction stores 16 tokens x 64 contiguous bytes.  No barrier:
            // nobody else touches these columns before the O barrier.  v comes from the [feature][token] accumulators by 2-byte
            // writes (one per packed half), which also replaces the second, transposed v product of the first version.
            const uint32_t rd = obase + (uint32_t)(lane >> 2) * kRowB;  // rows lane / 4 + 16 i, unit lane % 4 of this wave's block
            auto flush = [&](int sel) {
                u32x4 pc[4];
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int row = (lane >> 2) + 16 * i;
                    pc[i] = ld128(rd + (uint32_t)(16 * i) * kRowB + (uint32_t)(((4 * wave + (lane & 3)) ^ (row & 15)) << 4));
                }
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(pc[0]), "+v"(pc[1]), "+v"(pc[2]), "+v"(pc[3]));
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    *(u32x4*)(p.qkv_out + (int64_t)ltok[i] * (3 * C) + sel * C + 32 * wave + 8 * (lane & 3)) = pc[i];
            };
            auto stage_t = [&](const uint32_t (&w)[2][8]) {  // q^T / k^T: lane = token, word pair (2 g, 2 g + 1) = 4 features 8 g + 4 half ..
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int g = 0; g < 4; ++g) st64(obase + swz(t * 32 + l31, 4 * wave + g) + 8 * half, w[t][2 * g], w[t][2 * g + 1]);
            };
            stage_t(qw);
            flush(0);
            stage_t(kw);
            flush(1);
            // v: lane = feature d = l31 (+ half: tokens 4 half ..), word i of tile t = tokens (2 i & 3) + 8 (i >> 1) + 4 half and the next
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int tk = t * 32 + ((2 * i) & 3) + 8 * (i >> 1);  // + 4 half; the partner token is tk + 1
                    const uint32_t w = __builtin_bit_cast(u32x4, vf[t][i >> 2])[i & 3];
                    const int row0 = tk + 4 * half;  // (row & 15) of tk + 4 half and of tk + 1 + 4 half differ: two addresses
                    const uint32_t a0 = obase + swz(row0, 4 * wave + (l31 >> 3)) + (l31 & 7) * 2;
                    const uint32_t a1 = obase + swz(row0 + 1, 4 * wave + (l31 >> 3)) + (l31 & 7) * 2;
                    asm volatile("ds_write_b16 %0, %1" ::"v"(a0), "v"(w) : "memory");
                    asm volatile("ds_write_b16_d16_hi %0, %1" ::"v"(a1), "v"(w) : "memory");
                }
            flush(2);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        // cosine attention: k rows normalised, 1 / |q| folded into the per-query score factor.  The norms are those of the bf16
        // rows (what hs_window_attn_fwd / _bwd see in the stored qkv tensor), so that the saved log-sum-exp matches the scores
        // the backward recomputes
        float qinv[2] = {1.f, 1.f};
        if constexpr (COS) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                float sq = 0.f, sk = 0.f;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float q0 = __uint_as_float(qw[t][i] << 16), q1 = __uint_as_float(qw[t][i] & 0xffff0000u);
                    const float k0 = __uint_as_float(kw[t][i] << 16), k1 = __uint_as_float(kw[t][i] & 0xffff0000u);
                    sq = fmaf(q0, q0, fmaf(q1, q1, sq));
                    sk = fmaf(k0, k0, fmaf(k1, k1, sk));
                }
                sq += __shfl_xor(sq, 32, 64);
                sk += __shfl_xor(sk, 32, 64);
                qinv[t] = inv_norm(sq);
                const float kinv = inv_norm(sk);
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    kw[t][i] = pack_bf16x2(__uint_as_float(kw[t][i] << 16) * kinv, __uint_as_float(kw[t][i] & 0xffff0000u) * kinv);
            }
        }
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                qf[t][c] = __builtin_bit_cast(bf16x8, u32x4{qw[t][4 * c], qw[t][4 * c + 1], qw[t][4 * c + 2], qw[t][4 * c + 3]});
                kf[t][c] = __builtin_bit_cast(bf16x8, u32x4{kw[t][4 * c], kw[t][4 * c + 1], kw[t][4 * c + 2], kw[t][4 * c + 3]});
            }

        TR(4);
        TR(5);
        // ------------------------------------------------------------ S^T = k q^T, softmax over keys (log2 domain)
        f32x16 acc[2][2];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int qt = 0; qt < 2; ++qt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[kt][qt][r] = 0.f;
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int qt = 0; qt < 2; ++qt)
                    acc[kt][qt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[kt][c], qf[qt][c], acc[kt][qt], 0, 0, 0);
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            const float fq = hscale * kLog2e * qinv[qt];
            float m = -INFINITY;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float t = fmaf(acc[kt][qt][r], fq, biasr[kt][qt][r]);
                    acc[kt][qt][r] = t;
                    m = fmaxf(m, t);
                }
            if (mixed) {  // rare: windows cut by the shift boundary
                uint32_t labw[16];
                read_labels(labw);
                uint32_t mine = 0;  // the word holding this lane's query label: index qt * 8 + l31 / 4 is lane-dependent
#pragma unroll
                for (int i = 0; i < 8; ++i) mine = (l31 >> 2) == i ? labw[qt * 8 + i] : mine;
                const uint32_t mylab = (mine >> (8 * (l31 & 3))) & 0xffu;
                m = -INFINITY;
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        // key = kt * 32 + (r & 3) + 8 (r >> 2) + 4 half: word kt * 8 + 2 (r >> 2) + half, byte r & 3
                        const uint32_t word = half ? labw[kt * 8 + 2 * (r >> 2) + 1] : labw[kt * 8 + 2 * (r >> 2)];
                        const uint32_t klab = (word >> (8 * (r & 3))) & 0xffu;
                        float t = acc[kt][qt][r];
                        if (klab != mylab) t += kMaskLog2;
                        acc[kt][qt][r] = t;
                        m = fmaxf(m, t);
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
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[kt][qt][r] *= linv;
            if constexpr (TRAIN) {
                if (half == 0) {
                    p.lse_out[((int64_t)b_c * NH + wave) * N + (int64_t)r_c * kWs + qt * 32 + l31] = (m + __builtin_amdgcn_logf(l)) * kLn2;
                }
            }
        }

        TR(6);
        // ------------------------------------------------------------ O^T = v^T P^T  (rows = features, columns = queries)
        f32x16 ao[2];
#pragma unroll
        for (int qt = 0; qt < 2; ++qt)
#pragma unroll
            for (int r = 0; r < 16; ++r) ao[qt][r] = 0.f;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int qt = 0; qt < 2; ++qt)
                    ao[qt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[kt][c], pack8(acc[kt][qt], 8 * c), ao[qt], 0, 0, 0);
        // -> O tile [token][channel 32 h + d]: lane (token, half) writes 4 consecutive channels per register group
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            const int row = qt * 32 + l31;
#pragma unroll
            for (int g = 0; g < 4; ++g)
                st64(obase + swz(row, 4 * wave + g) + 8 * half, pack_bf16x2(ao[qt][4 * g], ao[qt][4 * g + 1]),
                     pack_bf16x2(ao[qt][4 * g + 2], ao[qt][4 * g + 3]));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        TR(7);
        __builtin_amdgcn_s_barrier();  // O tile complete; every wave is done with the x tile
        TR(8);
        if constexpr (TRAIN) store_rows(obase, p.o_out);
        // the residual operand (x rows of this wave's output pieces; L2 hits) is requested here, under the proj product
        // (unconditional, straight-line loads -- pieces that do not exist read row 0 -- pinned here by the scheduling barrier:
        // under per-piece conditions the compiler moved them back down to their use)
        u32x4v xres[XP];
        if (residual) {
#pragma unroll
            for (int j = 0; j < XP; ++j) {
                const bool ok = wave + NH * j < 16 && pchunk_of(j) < NCH;
                xres[j] = __builtin_nontemporal_load((const u32x4v*)(p.x + (ok ? (int64_t)tok[j] * C + pchunk_of(j) * 8 : (int64_t)0)));
            }
            __builtin_amdgcn_sched_barrier(0);
        }

        // ------------------------------------------------------------ Y^T = Wp O^T for this wave's 32 output channels
        f32x16 ay[2];
        {
            u32x4 bpv[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) bpv[g] = ld128(pbase + (P_BP + 32 * wave + 8 * g + 4 * half) * 4);
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bpv[0]), "+v"(bpv[1]), "+v"(bpv[2]), "+v"(bpv[3]));
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) ay[t][r] = __uint_as_float(bpv[r >> 2][r & 3]);
        }
        {
            const uint32_t oa0 = obase + l31 * kRowB, oa1 = obase + (32 + l31) * kRowB;
            const int sx = l31 & 15;
            u32x4 fo[2][2];
            fo[0][0] = ld128(oa0 + (uint32_t)(((0 + half) ^ sx) << 4));
            fo[0][1] = ld128(oa1 + (uint32_t)(((0 + half) ^ sx) << 4));
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int set = ks & 1;
                if (ks + 1 < KS) {
                    const uint32_t co = (uint32_t)(((2 * (ks + 1) + half) ^ sx) << 4);
                    fo[set ^ 1][0] = ld128(oa0 + co);
                    fo[set ^ 1][1] = ld128(oa1 + co);
                    asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(fo[set][0]), "+v"(fo[set][1]));
                } else {
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fo[set][0]), "+v"(fo[set][1]));
                }
                ay[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wp[ks], as_frag(fo[set][0]), ay[0], 0, 0, 0);
                ay[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wp[ks], as_frag(fo[set][1]), ay[1], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // -> staging in the (dead) x tile: lane (token, half), channels 32 w + 8 g + 4 half ..
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int row = t * 32 + l31;
#pragma unroll
            for (int g = 0; g < 4; ++g)
                st64(xbase + swz(row, 4 * wave + g) + 8 * half, pack_bf16x2(ay[t][4 * g], ay[t][4 * g + 1]),
                     pack_bf16x2(ay[t][4 * g + 2], ay[t][4 * g + 3]));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        TR(9);
        __builtin_amdgcn_s_barrier();  // staged tile complete; every wave is done with the O tile
        TR(10);

        // ------------------------------------------------------------ whole token rows -> out[token] (the scatter half of the shift)
        {
            // the NEXT window's x rows (requested a window ago) have landed, and with them everything this window has stored so far
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            TR(12);
            const bool norm2 = TRAIN && p.n2_out != nullptr;
            u32x4 rows[XP];
#pragma unroll
            for (int j = 0; j < XP; ++j)
                asm volatile("ds_read_b128 %0, %1" : "=v"(rows[j]) : "v"(xbase + (uint32_t)((wave + NH * j) * 1024 + lane * 16)));
            if (norm2) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (the parameter reads below would break the counted waits)
#pragma unroll
            for (int j = 0; j < XP; ++j) {
                asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(rows[j]) : "n"(XP - 1 - j));
                if (wave + NH * j < 16) {
                    const bool valid = pchunk_of(j) < NCH;
                    const int64_t e = (int64_t)tok[j] * C + pchunk_of(j) * 8;
                    u32x4 v = rows[j];
                    if (residual) {
                        const u32x4v xr = xres[j];
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            v[i] = pack_bf16x2(__uint_as_float(v[i] << 16) + __uint_as_float(xr[i] << 16),
                                               __uint_as_float(v[i] & 0xffff0000u) + __uint_as_float(xr[i] & 0xffff0000u));
                    }
                    if (valid) *(u32x4*)(p.out + e) = v;
                    if constexpr (TRAIN) {
                        if (norm2) {
                            // norm2 of the block (ref :337) on the row just formed: its 16 chunks sit in 16 adjacent lanes (in swizzled
                            // order), statistics by a 4-step xor tree inside the group, on the bf16 values the consumer of `out` would read
                            float f[8];
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                f[2 * i] = valid ? __uint_as_float(v[i] << 16) : 0.f;
                                f[2 * i + 1] = valid ? __uint_as_float(v[i] & 0xffff0000u) : 0.f;
                            }
                            float s1 = 0.f;
#pragma unroll
                            for (int i = 0; i < 8; ++i) s1 += f[i];
#pragma unroll
                            for (int o2 = 1; o2 < 16; o2 <<= 1) s1 += __shfl_xor(s1, o2, 64);
                            const float mean2 = s1 * (1.f / C);
                            float s2 = 0.f;
#pragma unroll
                            for (int i = 0; i < 8; ++i) {
                                f[i] = valid ? f[i] - mean2 : 0.f;
                                s2 = fmaf(f[i], f[i], s2);
                            }
#pragma unroll
                            for (int o2 = 1; o2 < 16; o2 <<= 1) s2 += __shfl_xor(s2, o2, 64);
                            const float rstd2 = rsqrtf(s2 * (1.f / C) + kLnEps);
                            const int pcx = valid ? pchunk_of(j) : 0;
                            const uint32_t ga = pbase + (P_LN2G + pcx * 8) * 4, ba = pbase + (P_LN2B + pcx * 8) * 4;
                            u32x4 gw0 = ld128(ga), gw1 = ld128(ga + 16), bw0 = ld128(ba), bw1 = ld128(ba + 16);
                            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(gw0), "+v"(gw1), "+v"(bw0), "+v"(bw1));
                            u32x4 o2v;
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                const float g0 = __uint_as_float(i < 2 ? gw0[2 * i] : gw1[2 * i - 4]), g1 = __uint_as_float(i < 2 ? gw0[2 * i + 1] : gw1[2 * i - 3]);
                                const float b0 = __uint_as_float(i < 2 ? bw0[2 * i] : bw1[2 * i - 4]), b1 = __uint_as_float(i < 2 ? bw0[2 * i + 1] : bw1[2 * i - 3]);
                                o2v[i] = pack_bf16x2(fmaf(f[2 * i] * rstd2, g0, b0), fmaf(f[2 * i + 1] * rstd2, g1, b1));
                            }
                            if (valid) *(u32x4*)(p.n2_out + e) = o2v;
                            if ((lane & 15) == 0) {
                                p.mean2_out[tok[j]] = mean2;
                                p.rstd2_out[tok[j]] = rstd2;
                            }
                        }
                    }
                }
            }
        }
        TR(11);
#ifdef HS_MOD_TRACE
        ++tr_w;
#endif
        b_c = b_n;
        r_c = r_n;
#pragma unroll
        for (int j = 0; j < XP; ++j) tok[j] = tok_next[j];
        cur ^= 1;
    }
#endif
}

template <int NH>
int launch_module(const ModParams& p0, bool cosine, hipStream_t stream) {
    ModParams p = p0;
    const int64_t windows = (int64_t)p.B * (p.N / kWs);
    const int cus = usable_cus();
    p.slots = (int)(windows < cus ? windows : cus);  // one persistent workgroup per CU (144 KB of LDS each)
    const bool train = p.qkv_out != nullptr;
#define HS_MOD_LAUNCH(COS, TRAIN) hipLaunchKernelGGL((attn_module_fwd_kernel<NH, COS, TRAIN>), dim3(p.slots), dim3(NH * 64), 0, stream, p)
    if (cosine) {
        if (train) HS_MOD_LAUNCH(true, true);
        else HS_MOD_LAUNCH(true, false);
    } else {
        if (train) HS_MOD_LAUNCH(false, true);
        else HS_MOD_LAUNCH(false, false);
    }
#undef HS_MOD_LAUNCH
    HS_LAUNCH_CHECK("attn_module_fwd");
    return HS_OK;
}

}  // namespace
}  // namespace hs

extern "C" {

int hs_window_attn_module_supported(int channels, int num_heads, int window_size, int dtype) {
    return dtype == HS_BF16 && window_size == hs::kWs && num_heads * 32 == channels && (channels == 96 || channels == 128);
}

#ifdef HS_MOD_TRACE
namespace {
unsigned long long* g_mod_trace = nullptr;
}
int hs_window_attn_module_set_trace(void* buf) {
    g_mod_trace = (unsigned long long*)buf;
    return 0;
}
#endif
namespace {
struct TrainOut {
    void *xn = nullptr, *qkv = nullptr, *o = nullptr;
    float *mean = nullptr, *rstd = nullptr, *lse = nullptr;
    const float *ln2_g = nullptr, *ln2_b = nullptr;  // optional norm2 behind the residual add
    void* n2 = nullptr;
    float *mean2 = nullptr, *rstd2 = nullptr;
};
int module_fwd_impl(const char* who, const void* x, void* out, const TrainOut& tr, const void* qkv_w, const float* qkv_b, const void* proj_w,
                    const float* proj_b, const float* ln_gamma, const float* ln_beta, const float* bias, const float* head_scale,
                    const int32_t* idx, int64_t roll, const uint8_t* labels, int batch, int64_t n_tokens, int channels, int num_heads,
                    int window_size, unsigned flags, int dtype, void* stream) {
    using namespace hs;
    HS_CHECK_ARG(x && out && qkv_w && proj_w && head_scale, "%s: null pointer", who);
    HS_CHECK_ARG(batch > 0 && n_tokens > 0 && n_tokens % kWs == 0, "%s: n_tokens must be a positive multiple of 64", who);
    HS_CHECK_ARG((ln_gamma == nullptr) == (ln_beta == nullptr), "%s: ln_gamma and ln_beta go together", who);
    HS_CHECK_ARG(roll >= 0 && roll < n_tokens, "%s: roll must be in [0, n_tokens)", who);
    if (!hs_window_attn_module_supported(channels, num_heads, window_size, dtype))
        return fail(HS_ERR_UNSUPPORTED, "%s: bf16, window 64, head_dim 32 and C = 96 or 128 only (got C = %d, heads %d, window %d): "
                    "the qkv weights must fit the LDS", who, channels, num_heads, window_size);
    // The x tiles are addressed through a buffer descriptor (32-bit byte offsets, < 2 GiB): larger activation tensors are
    // processed in batch chunks of whole images, one launch each (images are independent).
    const int64_t image_bytes = n_tokens * channels * 2;
    if (image_bytes > 0x7FFFFE00ll) return fail(HS_ERR_UNSUPPORTED, "%s: one image beyond the 2 GiB buffer-offset range", who);
    const int chunk = (int)std::min<int64_t>(batch, 0x7FFFFE00ll / image_bytes);
    const bool cosine = (flags & HS_ATTN_COSINE) != 0;
    for (int b0 = 0; b0 < batch; b0 += chunk) {
        ModParams p{};
        const int64_t t0 = (int64_t)b0 * n_tokens;
        p.x = (const uint16_t*)x + t0 * channels;
        p.out = (uint16_t*)out + t0 * channels;
        p.qkv_w = (const uint16_t*)qkv_w; p.qkv_b = qkv_b;
        p.proj_w = (const uint16_t*)proj_w; p.proj_b = proj_b; p.ln_g = ln_gamma; p.ln_b = ln_beta; p.bias = bias;
        p.head_scale = head_scale; p.idx = idx; p.roll = idx ? 0 : roll; p.labels = labels; p.B = std::min(chunk, batch - b0);
        p.N = n_tokens; p.flags = flags;
        p.dma_mode = 0;  // (1 / 2: the measured-and-rejected placements under the qkv product, profiles/archive_r01_r04/r04_attn_module_train.txt)
#ifdef HS_MOD_TRACE
        p.trace = g_mod_trace;
#endif
        if (tr.qkv) {
            p.xn_out = tr.xn ? (uint16_t*)tr.xn + t0 * channels : nullptr;
            p.qkv_out = (uint16_t*)tr.qkv + t0 * 3 * channels;
            p.o_out = (uint16_t*)tr.o + t0 * channels;
            p.mean_out = tr.mean ? tr.mean + t0 : nullptr;
            p.rstd_out = tr.rstd ? tr.rstd + t0 : nullptr;
            p.lse_out = tr.lse + t0 * num_heads;
            if (tr.n2) {
                p.ln2_g = tr.ln2_g; p.ln2_b = tr.ln2_b;
                p.n2_out = (uint16_t*)tr.n2 + t0 * channels;
                p.mean2_out = tr.mean2 + t0;
                p.rstd2_out = tr.rstd2 + t0;
            }
        }
        const int rc = num_heads == 4 ? launch_module<4>(p, cosine, (hipStream_t)stream) : launch_module<3>(p, cosine, (hipStream_t)stream);
        if (rc != HS_OK) return rc;
    }
    return HS_OK;
}
}  // namespace

int hs_window_attn_module_fwd(const void* x, void* out, const void* qkv_w, const float* qkv_b, const void* proj_w, const float* proj_b,
                              const float* ln_gamma, const float* ln_beta, const float* bias, const float* head_scale,
                              const int32_t* idx, int64_t roll, const uint8_t* labels, int batch, int64_t n_tokens, int channels,
                              int num_heads, int window_size, unsigned flags, int dtype, void* stream) {
    return module_fwd_impl("hs_window_attn_module_fwd", x, out, TrainOut{}, qkv_w, qkv_b, proj_w, proj_b, ln_gamma, ln_beta, bias,
                           head_scale, idx, roll, labels, batch, n_tokens, channels, num_heads, window_size, flags, dtype, stream);
}

int hs_window_attn_module_fwd_train(const void* x, void* out, void* xn_out, float* mean_out, float* rstd_out, void* qkv_out,
                                    void* attn_out, float* lse_out, const void* qkv_w, const float* qkv_b, const void* proj_w,
                                    const float* proj_b, const float* ln_gamma, const float* ln_beta, const float* bias,
                                    const float* head_scale, const int32_t* idx, int64_t roll, const uint8_t* labels,
                                    const float* norm2_gamma, const float* norm2_beta, void* n2_out, float* mean2_out, float* rstd2_out,
                                    int batch, int64_t n_tokens, int channels, int num_heads, int window_size, unsigned flags, int dtype,
                                    void* stream) {
    HS_CHECK_ARG(qkv_out && attn_out && lse_out, "hs_window_attn_module_fwd_train: null output");
    HS_CHECK_ARG((norm2_gamma != nullptr) == (norm2_beta != nullptr) && (norm2_gamma != nullptr) == (n2_out != nullptr) &&
                     (n2_out != nullptr) == (mean2_out != nullptr) && (n2_out != nullptr) == (rstd2_out != nullptr),
                 "hs_window_attn_module_fwd_train: norm2_gamma, norm2_beta, n2_out, mean2_out, rstd2_out go together");
    HS_CHECK_ARG(!ln_gamma || (xn_out && mean_out && rstd_out), "hs_window_attn_module_fwd_train: with a LayerNorm in front its output and statistics are saved too");
    TrainOut tr;
    tr.xn = xn_out; tr.qkv = qkv_out; tr.o = attn_out; tr.mean = mean_out; tr.rstd = rstd_out; tr.lse = lse_out;
    tr.ln2_g = norm2_gamma; tr.ln2_b = norm2_beta; tr.n2 = n2_out; tr.mean2 = mean2_out; tr.rstd2 = rstd2_out;
    return module_fwd_impl("hs_window_attn_module_fwd_train", x, out, tr, qkv_w, qkv_b, proj_w, proj_b, ln_gamma, ln_beta, bias,
                           head_scale, idx, roll, labels, batch, n_tokens, channels, num_heads, window_size, flags, dtype, stream);
}

}  // extern "C"
