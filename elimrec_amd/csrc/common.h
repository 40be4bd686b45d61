// Shared helpers for the gfx950 kernels behind include/elimrec_hip.h.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include "../../include/elimrec_hip.h"

namespace elimrec {

void set_error(const char *fmt, ...);

inline int check_hip(hipError_t e, const char *what) {
    if (e == hipSuccess) return 0;
    set_error("%s: %s", what, hipGetErrorString(e));
    return (int)e;
}

#define ELIMREC_LAUNCH_CHECK(name)                                   \
    do {                                                             \
        int _rc = ::elimrec::check_hip(hipGetLastError(), name);     \
        if (_rc) return _rc;                                         \
    } while (0)

#define ELIMREC_REQUIRE(cond, ...)                                   \
    do {                                                             \
        if (!(cond)) {                                               \
            ::elimrec::set_error(__VA_ARGS__);                       \
            return ELIMREC_E_BADARG;                                 \
        }                                                            \
    } while (0)

constexpr int kWave = 64;  // gfx950 wavefront

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

}  // namespace elimrec
