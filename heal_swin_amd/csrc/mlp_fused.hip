// Fused Mlp block for the stages whose Mlp is HBM-bound (C = 96 / 128, hidden = 4 C: stage 0 of HEAL-SWIN-T / -B):
//
//   forward    out = x + fc2( gelu( fc1( LayerNorm(x) ) ) )          reference Mlp.forward, models_torch/swin_hp_transformer.py:38-44,
//                                                                     inside the block's second residual branch :337-338 (v1 placement)
//   backward   dh = (dy W2) o gelu'(h),  dn = dh W1                   the input-gradient half of its autograd
//
// ONE launch each.  Composed, the forward is LayerNorm -> hs_gemm_nt(GELU epilogue) -> hs_gemm_nt(residual epilogue): 17 tensor-units
// of HBM traffic per token row (unit = C bf16) -- x in, n out | n in, h + act out (4 + 4) | act in (4), x in, out -- and the backward's
// two products 14 (dy in, h in, dh out | dh in, dn out).  Fused: 11 and 10 -- the hidden activations are written once and never
// re-read in the same pass (7 and 10 when the caller does not keep gelu(h): `act_out = NULL`).
//
// Both directions are the SAME kernel: tile in -> product 1 (output = 4 C wide) -> elementwise -> LDS tile -> product 2 (output = C wide):
//   forward    product 1 = fc1 (A = W1 [4C, C]),     elementwise = bias + GELU,           product 2 = fc2 (A = W2   [C, 4C])
//   backward   product 1 = dy W2 (A = W2^T [4C, C]), elementwise = . gelu'(saved h),      product 2 = dh W1 (A = W1^T [C, 4C])
// (the transposed bf16 weight copies are the ones the input-gradient GEMMs already use, ops.ParamCastCache.get_t).
//
// Workgroup = C / 16 wavefronts (8 at C = 128), persistent, 32-token tiles.  THE WEIGHTS LIVE IN REGISTERS for the whole launch:
// wave w owns hidden rows 64 w .. 64 w + 63 of product 1 (2 x C/16 A fragments of v_mfma_f32_32x32x16_bf16 = 64 VGPRs) and output
// channels 16 w .. 16 w + 15 of product 2 (4C/32 A fragments of v_mfma_f32_16x16x32_bf16 = 64 VGPRs), so no weight byte moves after the
// prologue and the tile size is free of any weight-reuse consideration -- 256 KB of weights do not fit the LDS at C = 128.
//   * x tiles (32 rows) arrive by buffer_load ... lds into one of two buffers, a tile ahead; forward: LayerNorm in place by the wave
//     that loaded the rows (16 lanes per row, DPP reductions), LayerNorm(x) and the statistics leave from its registers;
//   * product 1 is formed TRANSPOSED (D[hidden][token]: lane = token, 4 consecutive hidden units per register group), so the
//     elementwise step is lane-local and its packed result goes to the LDS tile [token][hidden] with 8-byte writes; every wave
//     reads its own 128-byte column block back as whole line segments and stores h (and gelu(h)) / dh rows -- no barrier for that;
//   * one barrier, then product 2 reads the whole tile as B operands (each wave all of it: 256 KB of LDS reads per tile and CU
//     beside 2 x 90 KB of HBM traffic) and leaves D[channel][token] in an fp32 staging tile; the NEXT iteration's first barrier
//     releases it: every wave adds the residual it kept in registers to its 4 rows and stores whole 2C-byte rows.
// Two barriers per 32 tokens.  LDS traffic of the main phases is inline asm (a compiler-visible LDS access beside the DMA queue
// would be preceded by s_waitcnt vmcnt(0), csrc/window_attn_module.hip).
#include <algorithm>

#include "hs_gelu.h"

namespace hs {
HS_DEFINE_SEED_EPOCH_SETTER(set_seed_epoch_mlp_fused)
namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef unsigned int u32x2v __attribute__((__vector_size__(8)));
typedef __attribute__((address_space(3))) void lds_void;

constexpr int kT = 32;  // tokens per tile
constexpr float kLnEps = 1e-5f;
__device__ constexpr uint32_t kOob = 0x7FFFFF00u;  // beyond every descriptor: the DMA writes zeros

struct MlpParams {
    const uint16_t* x;    // [M, C]  forward: the residual stream (LayerNorm input);  backward: dy
    const uint16_t* wa;   // [4C, C] A operand of product 1
    const uint16_t* wb;   // [C, 4C] A operand of product 2
    const float* ba;      // [4C] or null (forward: fc1 bias)
    const float* bb;      // [C] or null (forward: fc2 bias)
    const float* ln_g;    // [C] or null: LayerNorm in front (forward)
    const float* ln_b;
    const uint16_t* hin;  // backward: saved pre-activation h [M, 4C]
    uint16_t* n_out;      // forward: LayerNorm(x) [M, C] (null: not kept)
    float* mean_out;      // forward: LayerNorm statistics [M] (null: not kept)
    float* rstd_out;
    uint16_t* h_out;      // forward: h [M, 4C] (null: not kept);  backward: dh [M, 4C]
    uint16_t* act_out;    // forward: gelu(h) [M, 4C] (null: not kept)
    uint16_t* out;        // forward: x + mlp(LayerNorm(x)) [M, C];  backward: dn [M, C]
    int64_t tiles;        // M / 32
    int residual;         // forward: add x to the result
    // v2 norm placement (reference :334-335): out = x + LayerNorm(mlp(x)) -- the LayerNorm sits BEHIND product 2
    int ln_after;         // forward: ln_g / ln_b apply to the product-2 rows (no LayerNorm in front)
    uint16_t* m_out;      // forward, ln_after: the un-normalised rows mlp(x) [M, C] (input of the LayerNorm backward; null: not kept)
    const uint16_t* res_in;  // backward: rows [M, C] added to dn (the residual path's gradient), or null
    // train-mode regularisers of the branch (DROP instantiations; v2 placement only): Mlp.drop after the activation and after fc2
    // (reference :41, :43; counter-based masks of hs_device.h keyed by the element index in the [M_total, 4C] / [M_total, C] tensor,
    // i.e. the masks hs_gemm_nt's GELU epilogue and hs_layernorm_drop_* generate for the same seeds) and DropPath (:335) as the
    // per-sample factor row_scale[row / rows_per_sample]
    float drop_p;
    uint64_t seed_h, seed_o;
    const float* row_scale;    // [M_total / rows_per_sample] or null
    int tiles_per_sample;      // rows_per_sample / 32
    int64_t row0;              // first row of this launch in the whole tensor (mask counters, sample index)
};

__device__ __forceinline__ u32x4 ld128(uint32_t addr) {
    u32x4 v;
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr));
    return v;
}
__device__ __forceinline__ u32x2v ld64(uint32_t addr) {
    u32x2v v;
    asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"(addr));
    return v;
}
// Loads that are consumed at once are ONE asm statement with their wait: between a separate load asm and its wait asm the compiler
// is free to copy the destination registers (it believes the value is there) -- and did, for the residual rows: `raw = v` was
// emitted between the ds_read and the s_waitcnt, i.e. it copied registers the LDS had not written yet.
__device__ __forceinline__ u32x4 ld128_now(uint32_t addr) {
    u32x4 v;
    asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(addr) : "memory");
    return v;
}
__device__ __forceinline__ void ld128x2_now(uint32_t a0, uint32_t a1, u32x4& v0, u32x4& v1) {
    asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %3\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v0), "=&v"(v1) : "v"(a0), "v"(a1) : "memory");
}
__device__ __forceinline__ void ld128x4_now(uint32_t a0, uint32_t a1, uint32_t a2, uint32_t a3, u32x4& v0, u32x4& v1, u32x4& v2, u32x4& v3) {
    asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %5\n\tds_read_b128 %2, %6\n\tds_read_b128 %3, %7\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3)
                 : "v"(a0), "v"(a1), "v"(a2), "v"(a3)
                 : "memory");
}
__device__ __forceinline__ void ld64x4_now(uint32_t a0, uint32_t a1, uint32_t a2, uint32_t a3, u32x2v& v0, u32x2v& v1, u32x2v& v2, u32x2v& v3) {
    asm volatile("ds_read_b64 %0, %4\n\tds_read_b64 %1, %5\n\tds_read_b64 %2, %6\n\tds_read_b64 %3, %7\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3)
                 : "v"(a0), "v"(a1), "v"(a2), "v"(a3)
                 : "memory");
}
__device__ __forceinline__ void st64(uint32_t addr, uint32_t a, uint32_t b) {
    const u32x2v v = {a, b};
    asm volatile("ds_write_b64 %0, %1" ::"v"(addr), "v"(v) : "memory");
}
__device__ __forceinline__ void st128(uint32_t addr, const u32x4& v) { asm volatile("ds_write_b128 %0, %1" ::"v"(addr), "v"(v) : "memory"); }
__device__ __forceinline__ bf16x8 as_frag(const u32x4& v) { return __builtin_bit_cast(bf16x8, v); }
__device__ __forceinline__ float lo_f(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float hi_f(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }
// sum over the 16 lanes of a DPP row: quad xor 1, xor 2, then the two mirror steps (sums are symmetric, any pairing works)
__device__ __forceinline__ float row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));  // row_half_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));  // row_mirror
    return v;
}

template <int C, bool BWD, bool DROP>
__global__ void __launch_bounds__(C * 4, 1) mlp_fused_kernel(MlpParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int H = 4 * C, NW = C / 16, KS1 = C / 16, KS2 = H / 32, NCH = C / 8, HCH = H / 8;
    constexpr int XP = (8 + NW - 1) / NW;             // 1-KB pieces of an x tile (8) per wave
    constexpr int HPIECES = kT * HCH / 64;            // 1-KB pieces of a saved-h tile: 32 (C = 128) / 24 (C = 96)
    constexpr int HP = (HPIECES + NW - 1) / NW;
    constexpr int XPITCH = 256, APITCH = H * 2, YPITCH = 512;
    constexpr int X_OFF = 0, A_OFF = 2 * kT * XPITCH, Y_OFF = A_OFF + kT * APITCH, P_OFF = Y_OFF + kT * YPITCH;
    constexpr int P_BA = 0, P_BB = H, P_LNG = H + 128, P_LNB = H + 256;  // float offsets into the parameter block
    constexpr int HB_OFF = P_OFF + (H + 384) * 4;
    constexpr int SMEM = BWD ? HB_OFF + 2 * kT * APITCH : HB_OFF + kT * APITCH;  // backward: two saved-h tiles; forward: the h tile
    __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM];
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;

    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, l31 = lane & 31, l15 = lane & 15, q4 = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if ((int64_t)blockIdx.x >= p.tiles) return;
    const bool has_ln = !BWD && p.ln_g != nullptr && !p.ln_after;
    const bool ln_post = !BWD && p.ln_g != nullptr && p.ln_after;
    const bool keep_raw = !BWD && p.residual;
    const bool add_res = BWD && p.res_in != nullptr;
    const ElemRng rng_h(DROP ? p.drop_p : 0.f, p.seed_h), rng_o(DROP ? p.drop_p : 0.f, p.seed_o);
    const bool dropping = DROP && p.drop_p > 0.f;

    // tile offsets: [token][16-byte chunk] with the chunk index xor-ed by (row & 15) inside its aligned group of 16
    auto aoff = [](int row, int chunk) { return (uint32_t)(row * APITCH + ((chunk ^ (row & 15)) << 4)); };
    auto yoff = [](int row, int c4) { return (uint32_t)(row * YPITCH + ((c4 ^ (row & 15)) << 4)); };

    // ---------------------------------------------------------------- one-off: first tile, parameters, weights
    const int64_t x_bytes = p.tiles * kT * (int64_t)(C * 2), h_bytes = p.tiles * kT * (int64_t)(H * 2);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)std::min<int64_t>(x_bytes, 0x7FFFFE00ll), 0x00020000);
    const __amdgpu_buffer_rsrc_t rh =
        __builtin_amdgcn_make_buffer_rsrc((void*)(BWD ? p.hin : p.x), 0, BWD ? (int)std::min<int64_t>(h_bytes, 0x7FFFFE00ll) : 0, 0x00020000);
    // Per-lane piece coordinates are recomputed from an OPAQUE copy of the lane index that is refreshed every iteration: as loop
    // invariants the compiler hoists them, spills them (the weights own half of the register file) and reloads them inside the
    // loop -- a scratch load that waits behind every outstanding store (csrc/window_attn_module.hip has the same note).
    int lane_o = lane;
    auto issue_x = [&](int64_t ti, int buf) {
        const int l15 = lane_o & 15, q4 = lane_o >> 4;
#pragma unroll
        for (int j = 0; j < XP; ++j) {
            const int pc = wave + NW * j;
            if (pc < 8) {
                const int row = 4 * pc + q4, lc = l15 ^ (row & 15);
                const uint32_t voff = lc < NCH ? (uint32_t)(((uint32_t)ti * kT + row) * (C * 2) + lc * 16) : kOob;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_void*)(smem + X_OFF + buf * (kT * XPITCH) + pc * 1024), 16, voff, 0, 0, 0);
            }
        }
    };
    auto issue_h = [&](int64_t ti, int buf) {
        if constexpr (BWD) {
#pragma unroll
            for (int j = 0; j < HP; ++j) {
                const int pc = wave + NW * j;
                if (pc < HPIECES) {
                    const int slot = pc * 64 + lane_o, row = slot / HCH, lc = (slot % HCH) ^ (row & 15);
                    const uint32_t voff = (uint32_t)(((uint32_t)ti * kT + row) * (H * 2) + lc * 16);
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rh, (lds_void*)(smem + HB_OFF + buf * (kT * APITCH) + pc * 1024), 16, voff, 0, 0, 0);
                }
            }
        }
    };
    int64_t ti = blockIdx.x;
    issue_x(ti, 0);
    issue_h(ti, 0);
    {
        float* ps = (float*)(smem + P_OFF);
        for (int i = tid; i < H; i += NW * 64) ps[P_BA + i] = p.ba ? p.ba[i] : 0.f;
        for (int i = tid; i < C; i += NW * 64) {
            ps[P_BB + i] = p.bb ? p.bb[i] : 0.f;
            ps[P_LNG + i] = (has_ln || ln_post) ? p.ln_g[i] : 1.f;
            ps[P_LNB + i] = (has_ln || ln_post) ? p.ln_b[i] : 0.f;
        }
    }
    bf16x8 wa[2][KS1], wb[KS2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int ks = 0; ks < KS1; ++ks) wa[i][ks] = *(const bf16x8*)(p.wa + (int64_t)(64 * wave + 32 * i + l31) * C + 16 * ks + 8 * half);
#pragma unroll
    for (int ks = 0; ks < KS2; ++ks) wb[ks] = *(const bf16x8*)(p.wb + (int64_t)(16 * wave + l15) * H + 32 * ks + 8 * q4);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // parameter block visible (the first tile's rows are waited for by their own waves)

    const uint32_t pbase = lds0 + P_OFF, abase = lds0 + A_OFF, ybase = lds0 + Y_OFF;
    u32x4 raw[XP], raw_prev[XP];  // forward: this wave's x rows (16 bytes per lane and piece) for the residual add
#pragma unroll
    for (int j = 0; j < XP; ++j) raw[j] = raw_prev[j] = u32x4{0u, 0u, 0u, 0u};
    int64_t ti_prev = -1;
    int buf = 0;

    // rows of this wave's pieces of the finished tile `tp`: staged product 2 (+ residual) -> whole 2C-byte rows of `out`
    auto epilogue = [&](int64_t tp, const u32x4 (&res)[XP]) {
        const int l15 = lane_o & 15, q4 = lane_o >> 4;
#pragma unroll
        for (int j = 0; j < XP; ++j) {
            const int pc = wave + NW * j;
            if (pc < 8) {
                const int row = 4 * pc + q4, lc = l15 ^ (row & 15);
                const int lcc = lc < NCH ? lc : 0;
                u32x4 y0, y1;
                ld128x2_now(ybase + yoff(row, 2 * lcc), ybase + yoff(row, 2 * lcc + 1), y0, y1);
                float f[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    f[e] = __uint_as_float(y0[e]);
                    f[4 + e] = __uint_as_float(y1[e]);
                }
                const bool valid = lc < NCH;
                const int64_t grow = tp * kT + row;
                if (ln_post) {
                    // v2 placement: LayerNorm of the row just formed, on its bf16 rounding (what the composed path's LayerNorm
                    // kernel reads, and what the LayerNorm backward will read from m_out); 16 lanes per row as in the prologue form
                    const u32x4 mw = {pack_bf16x2(f[0], f[1]), pack_bf16x2(f[2], f[3]), pack_bf16x2(f[4], f[5]), pack_bf16x2(f[6], f[7])};
                    if (valid && p.m_out) *(u32x4*)(p.m_out + grow * C + lc * 8) = mw;
                    const uint32_t ga = pbase + (P_LNG + lcc * 8) * 4, ba = pbase + (P_LNB + lcc * 8) * 4;
                    u32x4 g0, g1, b0, b1;
                    ld128x4_now(ga, ga + 16, ba, ba + 16, g0, g1, b0, b1);
                    float s1 = 0.f, s2 = 0.f;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        f[2 * e] = valid ? lo_f(mw[e]) : 0.f;
                        f[2 * e + 1] = valid ? hi_f(mw[e]) : 0.f;
                    }
                    if constexpr (DROP) {  // Mlp.drop behind fc2: y = x + rs * LN(drop(m)), as hs_layernorm_drop_fwd
                        if (dropping) {
                            const int64_t e0 = ((p.row0 + grow) * C) + lcc * 8;
                            uint32_t hp = 0;
                            const uint32_t ck = rng_o.template run_key<8>(e0);
#pragma unroll
                            for (int e = 0; e < 8; ++e) f[e] *= rng_o.template run_mult<8>(e0, e, ck, hp);
                        }
                    }
#pragma unroll
                    for (int e = 0; e < 8; ++e) s1 += f[e];
                    const float mean = row16_sum(s1) * (1.f / C);
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        f[e] = valid ? f[e] - mean : 0.f;
                        s2 = fmaf(f[e], f[e], s2);
                    }
                    const float rstd = rsqrtf(row16_sum(s2) * (1.f / C) + kLnEps);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        f[e] = fmaf(f[e] * rstd, __uint_as_float(g0[e]), __uint_as_float(b0[e]));
                        f[4 + e] = fmaf(f[4 + e] * rstd, __uint_as_float(g1[e]), __uint_as_float(b1[e]));
                    }
                    if ((lane_o & 15) == 0 && p.mean_out) {
                        p.mean_out[grow] = mean;
                        p.rstd_out[grow] = rstd;
                    }
                    if constexpr (DROP) {
                        if (p.row_scale) {  // DropPath: one factor per sample; a tile never straddles two samples
                            const float rs = p.row_scale[(uint32_t)((p.row0 >> 5) + tp) / (uint32_t)p.tiles_per_sample];
#pragma unroll
                            for (int e = 0; e < 8; ++e) f[e] *= rs;
                        }
                    }
                }
                if (keep_raw || add_res) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        f[2 * e] += lo_f(res[j][e]);
                        f[2 * e + 1] += hi_f(res[j][e]);
                    }
                }
                const u32x4 o = {pack_bf16x2(f[0], f[1]), pack_bf16x2(f[2], f[3]), pack_bf16x2(f[4], f[5]), pack_bf16x2(f[6], f[7])};
                if (valid) *(u32x4*)(p.out + grow * C + lc * 8) = o;
            }
        }
    };

    for (; ti < p.tiles; ti += gridDim.x) {
        asm volatile("" : "+v"(lane_o));
        const uint32_t xbase = lds0 + X_OFF + buf * (kT * XPITCH);
        const uint32_t hbase = lds0 + HB_OFF + buf * (kT * APITCH);
        const int64_t ti_next = ti + gridDim.x;
        const bool more = ti_next < p.tiles;
        // ------------------------------------------------------------ own rows of the x tile: residual copy, LayerNorm in place
        // (this wave's pieces landed before the previous iteration's row stores: vmcnt(0) below)
#pragma unroll
        for (int j = 0; j < XP; ++j) raw_prev[j] = raw[j];
        if constexpr (!BWD) {
            if (has_ln || keep_raw) {
#pragma unroll
                for (int j = 0; j < XP; ++j) {
                    const int pc = wave + NW * j;
                    if (pc < 8) {
                        const int l15 = lane_o & 15, q4 = lane_o >> 4;
                        const int row = 4 * pc + q4, lc = l15 ^ (row & 15);
                        const bool valid = lc < NCH;
                        const uint32_t addr = xbase + pc * 1024 + lane_o * 16;
                        const u32x4 v = ld128_now(addr);
                        raw[j] = v;
                        if (!has_ln) continue;
                        const int lcc = valid ? lc : 0;
                        const uint32_t ga = pbase + (P_LNG + lcc * 8) * 4, ba = pbase + (P_LNB + lcc * 8) * 4;
                        u32x4 g0, g1, b0, b1;
                        ld128x4_now(ga, ga + 16, ba, ba + 16, g0, g1, b0, b1);
                        float f[8], s1 = 0.f, s2 = 0.f;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            f[2 * e] = valid ? lo_f(v[e]) : 0.f;
                            f[2 * e + 1] = valid ? hi_f(v[e]) : 0.f;
                            s1 += f[2 * e] + f[2 * e + 1];
                        }
                        const float mean = row16_sum(s1) * (1.f / C);
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            f[e] = valid ? f[e] - mean : 0.f;
                            s2 = fmaf(f[e], f[e], s2);
                        }
                        const float rstd = rsqrtf(row16_sum(s2) * (1.f / C) + kLnEps);
                        u32x4 o;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float ga0 = __uint_as_float(e < 2 ? g0[2 * e] : g1[2 * e - 4]), ga1 = __uint_as_float(e < 2 ? g0[2 * e + 1] : g1[2 * e - 3]);
                            const float be0 = __uint_as_float(e < 2 ? b0[2 * e] : b1[2 * e - 4]), be1 = __uint_as_float(e < 2 ? b0[2 * e + 1] : b1[2 * e - 3]);
                            o[e] = pack_bf16x2(fmaf(f[2 * e] * rstd, ga0, be0), fmaf(f[2 * e + 1] * rstd, ga1, be1));
                        }
                        st128(addr, o);
                        const int64_t grow = ti * kT + row;
                        if (valid && p.n_out) *(u32x4*)(p.n_out + grow * C + lc * 8) = o;
                        if (l15 == 0 && p.mean_out) {
                            p.mean_out[grow] = mean;
                            p.rstd_out[grow] = rstd;
                        }
                    }
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();  // A: the x tile is complete (normalised); the previous tile's staged result is complete
        // (requested as early as its buffers are free: everybody passed the previous tile's product 1 / elementwise step)
        if (more) {
            issue_x(ti_next, buf ^ 1);
            issue_h(ti_next, buf ^ 1);
        }
        if (ti_prev >= 0) epilogue(ti_prev, raw_prev);
        if constexpr (BWD) {
            // the residual-path rows of THIS tile (added in the next iteration's epilogue): requested here, claimed by the vmcnt(0)
            // in front of the row stores below.  Inline asm, so that the compiler does not put a wait of its own in front of their
            // use -- it would stand right behind the next tile's DMA
            if (add_res) {
                const int l15o = lane_o & 15, q4o = lane_o >> 4;
#pragma unroll
                for (int j = 0; j < XP; ++j) {
                    const int pc = wave + NW * j;
                    const int row = 4 * (pc < 8 ? pc : 0) + q4o, lc = l15o ^ (row & 15);
                    const uint16_t* src = p.res_in + (ti * kT + row) * C + (lc < NCH ? lc : 0) * 8;
                    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(raw[j]) : "v"(src) : "memory");
                }
            }
        }

        // ------------------------------------------------------------ product 1, transposed: D[hidden 64 w + 32 i + ..][token]
        f32x16 acc[2];
        if constexpr (!BWD) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                u32x4 bv[4];
                const uint32_t a = pbase + (P_BA + 64 * wave + 32 * i + 4 * half) * 4;
                ld128x4_now(a, a + 32, a + 64, a + 96, bv[0], bv[1], bv[2], bv[3]);
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][r] = __uint_as_float(bv[r >> 2][r & 3]);
            }
        } else {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        }
        {
            const uint32_t xa = xbase + l31 * XPITCH;
            const int sx = l31 & 15;
            u32x4 fx[2];
            fx[0] = ld128(xa + (uint32_t)(((0 + half) ^ sx) << 4));
#pragma unroll
            for (int ks = 0; ks < KS1; ++ks) {
                const int set = ks & 1;
                if (ks + 1 < KS1) {
                    fx[set ^ 1] = ld128(xa + (uint32_t)(((2 * (ks + 1) + half) ^ sx) << 4));
                    asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(fx[set]));
                } else {
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fx[set]));
                }
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[0][ks], as_frag(fx[set]), acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[1][ks], as_frag(fx[set]), acc[1], 0, 0, 0);
            }
        }
        // ------------------------------------------------------------ elementwise, lane-local (lane = token, 4 hidden units per group)
        // The first result (forward: h, backward: dh) goes to the LDS tile [token][hidden] as it is formed -- 8 bytes per lane and
        // register group; the tile is free since barrier A -- so only gelu(h) stays in registers (16) across the step.
        u32x2v hw[2][4];  // backward: the saved h of this lane's 32 accumulator positions
        const uint32_t h2base = lds0 + HB_OFF;  // forward: h goes to a tile of its own, gelu(h) to the tile product 2 reads
        // (DROP) generator chunk (8 elements) of this lane's token in the [M_total, 4C] tensor, at the wave's first hidden unit
        const uint64_t cbase = DROP ? (uint64_t)(p.row0 + ti * kT + l31) * (uint64_t)(H / 8) + (uint64_t)(8 * wave) : 0ull;
        if constexpr (BWD) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
                ld64x4_now(hbase + aoff(l31, 8 * wave + 4 * i + 0) + 8 * half, hbase + aoff(l31, 8 * wave + 4 * i + 1) + 8 * half,
                           hbase + aoff(l31, 8 * wave + 4 * i + 2) + 8 * half, hbase + aoff(l31, 8 * wave + 4 * i + 3) + 8 * half, hw[i][0],
                           hw[i][1], hw[i][2], hw[i][3]);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x2 v[2] = {f32x2{acc[i][4 * g], acc[i][4 * g + 1]}, f32x2{acc[i][4 * g + 2], acc[i][4 * g + 3]}};
                const uint32_t off = aoff(l31, 8 * wave + 4 * i + g) + 8 * half, dst = abase + off;
                // Mlp.drop behind the activation: the lane's four consecutive hidden units are the first (half = 0) or second half of
                // a chunk of the generator; one chunk key, one multiply per element pair, formed where the pair is used
                uint32_t ck = 0;
                if constexpr (DROP) {
                    if (dropping) ck = rng_h.chunk_key(cbase + (uint32_t)(4 * i + g));
                }
                auto drop2 = [&](int t) {
                    const uint32_t hh = ElemRng::pair_bits(ck, t == 0 ? (half ? ElemRng::kM2 : ElemRng::kM0) : (half ? ElemRng::kM3 : ElemRng::kM1));
                    return f32x2{rng_h.keep_lo(hh), rng_h.keep_hi(hh)};
                };
                if constexpr (!BWD) {
                    st64(h2base + off, pack_bf16x2(v[0].x, v[0].y), pack_bf16x2(v[1].x, v[1].y));
                    v[0] = gelu2(v[0]);
                    v[1] = gelu2(v[1]);
                    if constexpr (DROP) {
                        if (dropping) {
                            v[0] *= drop2(0);
                            v[1] *= drop2(1);
                        }
                    }
                    st64(dst, pack_bf16x2(v[0].x, v[0].y), pack_bf16x2(v[1].x, v[1].y));
                } else {
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        const f32x2 hx = {lo_f(hw[i][g][t]), hi_f(hw[i][g][t])};
                        v[t] *= gelu_grad2(hx);
                        if constexpr (DROP) {
                            if (dropping) v[t] *= drop2(t);
                        }
                    }
                    st64(dst, pack_bf16x2(v[0].x, v[0].y), pack_bf16x2(v[1].x, v[1].y));
                }
            }
        // the NEXT tile's rows (requested behind barrier A) have landed -- waited for here, in front of this tile's row stores,
        // because vmcnt counts loads and stores together
        if constexpr (BWD && XP == 1) asm volatile("s_waitcnt vmcnt(0)" : "+v"(raw[0])::"memory");
        else if constexpr (BWD) asm volatile("s_waitcnt vmcnt(0)" : "+v"(raw[0]), "+v"(raw[XP - 1])::"memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        // ------------------------------------------------------------ own column block of the tile back out as whole row segments
        auto rows_out = [&](uint32_t tile, uint16_t* dst) {  // 32 rows x 128 bytes of this wave's block: 4 instructions of 8 rows x 8 chunks
            u32x4 pc[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) pc[j] = ld128(tile + aoff(8 * j + (lane_o >> 3), 8 * wave + (lane_o & 7)));
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(pc[j]) : "n"(3 - j));
                *(u32x4*)(dst + (ti * kT + 8 * j + (lane_o >> 3)) * H + 64 * wave + 8 * (lane_o & 7)) = pc[j];
            }
        };
        if constexpr (!BWD) {
            if (p.h_out) rows_out(h2base, p.h_out);
            if (p.act_out) rows_out(abase, p.act_out);
        } else {
            rows_out(abase, p.h_out);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();  // B: the tile is complete; everybody is done with the x tile and the staged previous result

        // ------------------------------------------------------------ product 2: D[channel 16 w + ..][token], K = 4 C
        f32x4 y[2];
        {
            const u32x4 bbv = ld128_now(pbase + (P_BB + 16 * wave + 4 * q4) * 4);
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int e = 0; e < 4; ++e) y[t][e] = BWD ? 0.f : __uint_as_float(bbv[e]);
        }
        {
            const uint32_t ra0 = abase + l15 * APITCH, ra1 = abase + (16 + l15) * APITCH;
            u32x4 fb[2][2];
            auto reads = [&](int ks, int set) {
                const uint32_t co = (uint32_t)(((4 * ks + q4) ^ l15) << 4);
                fb[set][0] = ld128(ra0 + co);
                fb[set][1] = ld128(ra1 + co);
            };
            reads(0, 0);
#pragma unroll
            for (int ks = 0; ks < KS2; ++ks) {
                const int set = ks & 1;
                if (ks + 1 < KS2) {
                    reads(ks + 1, set ^ 1);
                    asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(fb[set][0]), "+v"(fb[set][1]));
                } else {
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fb[set][0]), "+v"(fb[set][1]));
                }
                y[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[ks], as_frag(fb[set][0]), y[0], 0, 0, 0);
                y[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[ks], as_frag(fb[set][1]), y[1], 0, 0, 0);
            }
        }
        // -> fp32 staging [token][channel]: lane (token 16 t + l15) writes its 4 consecutive channels 16 w + 4 q4 ..
        // (the writes are inline asm and read the MFMA destinations directly: the compiler's hazard recogniser does not look
        // inside asm statements, so the wait states between the last v_mfma and its first reader are spelled out -- without them
        // the staged values missed the last k-steps)
        asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" : "+v"(y[0]), "+v"(y[1])::"memory");  // (tied to y: the MFMAs cannot sink below it)
#pragma unroll
        for (int t = 0; t < 2; ++t) st128(ybase + yoff(16 * t + l15, 4 * wave + q4), __builtin_bit_cast(u32x4, y[t]));
        ti_prev = ti;
        buf ^= 1;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int j = 0; j < XP; ++j) raw_prev[j] = raw[j];
    epilogue(ti_prev, raw_prev);
#endif
}

template <int C>
int launch_mlp(const MlpParams& p, bool bwd, bool drop, hipStream_t stream) {
    const int64_t grid = std::min<int64_t>(p.tiles, usable_cus());  // one persistent workgroup per CU (weights in registers)
    if (bwd && drop) hipLaunchKernelGGL((mlp_fused_kernel<C, true, true>), dim3((unsigned)grid), dim3(C * 4), 0, stream, p);
    else if (bwd) hipLaunchKernelGGL((mlp_fused_kernel<C, true, false>), dim3((unsigned)grid), dim3(C * 4), 0, stream, p);
    else if (drop) hipLaunchKernelGGL((mlp_fused_kernel<C, false, true>), dim3((unsigned)grid), dim3(C * 4), 0, stream, p);
    else hipLaunchKernelGGL((mlp_fused_kernel<C, false, false>), dim3((unsigned)grid), dim3(C * 4), 0, stream, p);
    HS_LAUNCH_CHECK("mlp_fused");
    return HS_OK;
}

// rows per launch: the tiles are addressed through buffer descriptors with 32-bit byte offsets (< 2 GiB; the widest tensor is [rows, 4 C])
int64_t rows_per_launch(int C) { return (0x7FFFFE00ll / (8 * C)) / kT * kT; }

}  // namespace
}  // namespace hs

extern "C" {

int hs_mlp_fused_supported(int channels, int hidden, int dtype) {
    return dtype == HS_BF16 && (channels == 96 || channels == 128) && hidden == 4 * channels;
}

static int mlp_fused_fwd_impl(const void* x, const float* ln_gamma, const float* ln_beta, const void* w1, const float* b1, const void* w2,
                              const float* b2, void* n_out, float* mean_out, float* rstd_out, void* h_out, void* act_out, void* out,
                              int64_t rows, int channels, int hidden, unsigned flags, int dtype, void* stream, bool drop,
                              const float* row_scale, int64_t rows_per_sample, float drop_p, uint64_t seed_h, uint64_t seed_o) {
    using namespace hs;
    HS_CHECK_ARG(x && w1 && w2 && out, "hs_mlp_fused_fwd: null pointer");
    HS_CHECK_ARG(!(flags & HS_MLP_NORM_AFTER) || ln_gamma, "hs_mlp_fused_fwd: HS_MLP_NORM_AFTER needs ln_gamma / ln_beta");
    HS_CHECK_ARG((ln_gamma == nullptr) == (ln_beta == nullptr), "hs_mlp_fused_fwd: ln_gamma and ln_beta go together");
    HS_CHECK_ARG((mean_out == nullptr) == (rstd_out == nullptr), "hs_mlp_fused_fwd: mean_out and rstd_out go together");
    HS_CHECK_ARG(rows > 0 && rows % kT == 0, "hs_mlp_fused_fwd: rows must be a positive multiple of 32");
    if (!hs_mlp_fused_supported(channels, hidden, dtype))
        return fail(HS_ERR_UNSUPPORTED, "hs_mlp_fused_fwd: bf16, C = 96 or 128 and hidden = 4 C only (got C = %d, hidden %d): the weights must fit the registers",
                    channels, hidden);
    const int64_t step = rows_per_launch(channels);
    for (int64_t r0 = 0; r0 < rows; r0 += step) {
        const int64_t n = std::min(step, rows - r0);
        MlpParams p{};
        p.x = (const uint16_t*)x + r0 * channels;
        p.wa = (const uint16_t*)w1; p.wb = (const uint16_t*)w2; p.ba = b1; p.bb = b2; p.ln_g = ln_gamma; p.ln_b = ln_beta;
        p.ln_after = (flags & HS_MLP_NORM_AFTER) ? 1 : 0;
        // (v2 placement: the kept [rows, C] tensor is the LayerNorm's INPUT mlp(x); v1: its output LayerNorm(x))
        p.n_out = (n_out && !p.ln_after) ? (uint16_t*)n_out + r0 * channels : nullptr;
        p.m_out = (n_out && p.ln_after) ? (uint16_t*)n_out + r0 * channels : nullptr;
        p.mean_out = mean_out ? mean_out + r0 : nullptr;
        p.rstd_out = rstd_out ? rstd_out + r0 : nullptr;
        p.h_out = h_out ? (uint16_t*)h_out + r0 * hidden : nullptr;
        p.act_out = act_out ? (uint16_t*)act_out + r0 * hidden : nullptr;
        p.out = (uint16_t*)out + r0 * channels;
        p.tiles = n / kT;
        p.residual = (flags & HS_ATTN_RESIDUAL) ? 1 : 0;
        p.drop_p = drop_p; p.seed_h = seed_h; p.seed_o = seed_o; p.row_scale = row_scale;
        p.tiles_per_sample = (int)(rows_per_sample / kT); p.row0 = r0;
        const int rc = channels == 128 ? launch_mlp<128>(p, false, drop, (hipStream_t)stream) : launch_mlp<96>(p, false, drop, (hipStream_t)stream);
        if (rc != HS_OK) return rc;
    }
    return HS_OK;
}

int hs_mlp_fused_fwd(const void* x, const float* ln_gamma, const float* ln_beta, const void* w1, const float* b1, const void* w2,
                     const float* b2, void* n_out, float* mean_out, float* rstd_out, void* h_out, void* act_out, void* out, int64_t rows,
                     int channels, int hidden, unsigned flags, int dtype, void* stream) {
    return mlp_fused_fwd_impl(x, ln_gamma, ln_beta, w1, b1, w2, b2, n_out, mean_out, rstd_out, h_out, act_out, out, rows, channels, hidden,
                              flags, dtype, stream, false, nullptr, 32, 0.f, 0, 0);
}

int hs_mlp_fused_drop_fwd(const void* x, const float* ln_gamma, const float* ln_beta, const void* w1, const float* b1, const void* w2,
                          const float* b2, void* m_out, float* mean_out, float* rstd_out, void* h_out, void* act_out, void* out,
                          const float* row_scale, int64_t rows_per_sample, float drop_p, uint64_t seed_hidden, uint64_t seed_out,
                          int64_t rows, int channels, int hidden, unsigned flags, int dtype, void* stream) {
    using namespace hs;
    HS_CHECK_ARG(flags & HS_MLP_NORM_AFTER, "hs_mlp_fused_drop_fwd: the stochastic form exists for the v2 placement (HS_MLP_NORM_AFTER) only");
    HS_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f, "hs_mlp_fused_drop_fwd: drop_p must be in [0, 1)");
    HS_CHECK_ARG(!row_scale || (rows_per_sample > 0 && rows_per_sample % kT == 0 && rows % rows_per_sample == 0),
                 "hs_mlp_fused_drop_fwd: rows_per_sample must be a multiple of 32 that divides rows");
    return mlp_fused_fwd_impl(x, ln_gamma, ln_beta, w1, b1, w2, b2, m_out, mean_out, rstd_out, h_out, act_out, out, rows, channels, hidden,
                              flags, dtype, stream, true, row_scale, row_scale ? rows_per_sample : 32, drop_p, seed_hidden, seed_out);
}

static int mlp_fused_bwd_impl(const void* dy, const void* h, const void* w2_t, const void* w1_t, const void* dres, void* dh, void* dn,
                              int64_t rows, int channels, int hidden, int dtype, void* stream, bool drop, float drop_p, uint64_t seed_h) {
    using namespace hs;
    HS_CHECK_ARG(dy && h && w2_t && w1_t && dh && dn, "hs_mlp_fused_bwd: null pointer");
    HS_CHECK_ARG(rows > 0 && rows % kT == 0, "hs_mlp_fused_bwd: rows must be a positive multiple of 32");
    if (!hs_mlp_fused_supported(channels, hidden, dtype))
        return fail(HS_ERR_UNSUPPORTED, "hs_mlp_fused_bwd: bf16, C = 96 or 128 and hidden = 4 C only (got C = %d, hidden %d)", channels, hidden);
    const int64_t step = rows_per_launch(channels);
    for (int64_t r0 = 0; r0 < rows; r0 += step) {
        const int64_t n = std::min(step, rows - r0);
        MlpParams p{};
        p.x = (const uint16_t*)dy + r0 * channels;
        p.wa = (const uint16_t*)w2_t; p.wb = (const uint16_t*)w1_t;
        p.hin = (const uint16_t*)h + r0 * hidden;
        p.h_out = (uint16_t*)dh + r0 * hidden;
        p.res_in = dres ? (const uint16_t*)dres + r0 * channels : nullptr;
        p.out = (uint16_t*)dn + r0 * channels;
        p.tiles = n / kT;
        p.drop_p = drop_p; p.seed_h = seed_h; p.row0 = r0; p.tiles_per_sample = 1;
        const int rc = channels == 128 ? launch_mlp<128>(p, true, drop, (hipStream_t)stream) : launch_mlp<96>(p, true, drop, (hipStream_t)stream);
        if (rc != HS_OK) return rc;
    }
    return HS_OK;
}

int hs_mlp_fused_bwd(const void* dy, const void* h, const void* w2_t, const void* w1_t, const void* dres, void* dh, void* dn, int64_t rows,
                     int channels, int hidden, int dtype, void* stream) {
    return mlp_fused_bwd_impl(dy, h, w2_t, w1_t, dres, dh, dn, rows, channels, hidden, dtype, stream, false, 0.f, 0);
}

int hs_mlp_fused_drop_bwd(const void* dy, const void* h, const void* w2_t, const void* w1_t, const void* dres, void* dh, void* dn,
                          float drop_p, uint64_t seed_hidden, int64_t rows, int channels, int hidden, int dtype, void* stream) {
    using namespace hs;
    HS_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f, "hs_mlp_fused_drop_bwd: drop_p must be in [0, 1)");
    return mlp_fused_bwd_impl(dy, h, w2_t, w1_t, dres, dh, dn, rows, channels, hidden, dtype, stream, true, drop_p, seed_hidden);
}

}  // extern "C"
