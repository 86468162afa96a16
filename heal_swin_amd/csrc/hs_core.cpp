// Status / error plumbing of libhealswin.
#include <hip/hip_runtime_api.h>

#include <atomic>
#include <cstdlib>
#include <cstring>

#include "hs_common.h"

namespace hs {
namespace {
std::atomic<int>& reserved_slot() {
    static std::atomic<int> v{[] {
        const char* e = getenv("HS_RESERVED_CUS");
        const int n = e ? atoi(e) : 0;
        return n < 0 ? 0 : (n > 128 ? 128 : (n / 8) * 8);
    }()};
    return v;
}
}  // namespace
int reserved_cus() { return reserved_slot().load(std::memory_order_relaxed); }

char* error_buffer() {
    static thread_local char buf[512] = {0};
    return buf;
}
int fail(int status, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(error_buffer(), 512, fmt, ap);
    va_end(ap);
    return status;
}
}  // namespace hs

static const void* g_seed_epoch_host = nullptr;

extern "C" {
const char* hs_version(void) { return "healswin 0.1 (gfx950)"; }
const char* hs_last_error(void) { return hs::error_buffer(); }
const char* hs_status_string(int status) {
    switch (status) {
        case HS_OK: return "ok";
        case HS_ERR_INVALID_ARG: return "invalid argument";
        case HS_ERR_UNSUPPORTED: return "unsupported shape or dtype";
        case HS_ERR_HIP: return "HIP runtime error";
        case HS_ERR_NOT_PERMUTATION: return "shift is not a permutation";
        default: return "unknown status";
    }
}
int hs_set_reserved_cus(int n) {
    HS_CHECK_ARG(n >= 0 && n <= 128 && n % 8 == 0, "hs_set_reserved_cus: a multiple of 8 in [0, 128] (one CU per XCD at a time)");
    hs::reserved_slot().store(n, std::memory_order_relaxed);
    return HS_OK;
}
int hs_get_reserved_cus(void) { return hs::reserved_cus(); }
int hs_set_seed_epoch(const void* counter) {
    // (one static __device__ pointer per translation unit with stochastic kernels: hs_device.h)
    int (*const setters[])(const void*) = {hs::set_seed_epoch_gelu, hs::set_seed_epoch_layernorm, hs::set_seed_epoch_gemm_nt, hs::set_seed_epoch_mlp_fused,
                                           hs::set_seed_epoch_attn_generic, hs::set_seed_epoch_attn_mfma, hs::set_seed_epoch_attn_mfma_f32};
    for (auto set : setters)
        if (set(counter) != HS_OK) {
            (void)hipGetLastError();
            return hs::fail(HS_ERR_HIP, "hs_set_seed_epoch: hipMemcpyToSymbol failed");
        }
    g_seed_epoch_host = counter;
    return HS_OK;
}
const void* hs_get_seed_epoch(void) { return g_seed_epoch_host; }
int hs_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}
}
