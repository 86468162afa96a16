// Internal helpers shared by the libhealswin translation units (not part of the ABI).
#pragma once
#include <cstdarg>
#include <cstdio>
#include <cstdint>

#include "../../include/healswin.h"

namespace hs {

// thread-local last-error text, returned by hs_last_error()
char* error_buffer();
int fail(int status, const char* fmt, ...);

inline int isqrt_pow2_window(int ws) {
    // window_size must be 4^k (a sqrt(Ws) x sqrt(Ws) nested block, hp_windowing.py:16); returns side or -1
    int side = 1;
    while (side * side < ws) side <<= 1;
    return (side * side == ws) ? side : -1;
}

inline bool is_pow2(int64_t v) { return v > 0 && (v & (v - 1)) == 0; }

// Compute units the persistent / one-resident-round launches may plan for: 256 minus hs_set_reserved_cus(), spread evenly over
// the 8 XCDs.  The kernels whose grids are sized to exactly the resident capacity of the chip (hs_linear_wgrad, hs_gemm_nt,
// the attention kernels, the module kernel) leave the reserved CUs free for kernels of OTHER streams that must make progress
// at the same time -- RCCL's all-reduce kernels during the backward of a data-parallel run: a persistent grid that fills
// every CU either delays them to its end (no overlap) or, if they were resident first, runs its last workgroups as a second
// round (profiles/archive_r01_r04/r03_cu_contention.json).
int reserved_cus();
inline int usable_cus_per_xcd() { return 32 - reserved_cus() / 8; }
inline int usable_cus() { return 8 * usable_cus_per_xcd(); }

// per translation unit with stochastic kernels (HS_DEFINE_SEED_EPOCH_SETTER); hs_set_seed_epoch (hs_core.cpp) calls them all
int set_seed_epoch_gelu(const void* counter);
int set_seed_epoch_layernorm(const void* counter);
int set_seed_epoch_gemm_nt(const void* counter);
int set_seed_epoch_mlp_fused(const void* counter);
int set_seed_epoch_attn_generic(const void* counter);
int set_seed_epoch_attn_mfma(const void* counter);
int set_seed_epoch_attn_mfma_f32(const void* counter);

}  // namespace hs

#define HS_CHECK_ARG(cond, ...)                                        \
    do {                                                               \
        if (!(cond)) return hs::fail(HS_ERR_INVALID_ARG, __VA_ARGS__); \
    } while (0)
