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

}  // namespace hs

#define HS_CHECK_ARG(cond, ...)                                        \
    do {                                                               \
        if (!(cond)) return hs::fail(HS_ERR_INVALID_ARG, __VA_ARGS__); \
    } while (0)
