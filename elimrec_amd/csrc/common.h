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

// Layout of the packed head weights (head.hip packs, head.hip and bpr.hip read): the forward operands of the fused head
// (feature Linears, fusion Linear for users / items, single-modal heads), then -- 16-row forms -- the operands of the head
// BACKWARD dOut = dY_f W_side + dY_m Ws_m, i.e. B[k][c] = W[k][c]: the fusion weights as [C columns x 64], the heads' as
// [64 x 64]. recdim 64.
struct HeadPackLayout { int64_t Wm[3], Wf[2], Ws[3], Bf[2], Bs[3], fwd_total, total; };
static inline HeadPackLayout head_pack_layout(int n_mod, const int *D) {
    HeadPackLayout L = {};
    int64_t off = 0;
    const int64_t C = (int64_t)(1 + n_mod) * 64;
    for (int m = 0; m < n_mod; ++m) { L.Wm[m] = off; off += 64 * (int64_t)D[m]; }
    for (int sd = 0; sd < 2; ++sd) { L.Wf[sd] = off; off += 64 * C; }
    for (int m = 0; m < n_mod; ++m) { L.Ws[m] = off; off += 64 * 64; }
    L.fwd_total = off;
    for (int sd = 0; sd < 2; ++sd) { L.Bf[sd] = off; off += C * 64; }
    for (int m = 0; m < n_mod; ++m) { L.Bs[m] = off; off += 64 * 64; }
    L.total = off;
    return L;
}

}  // namespace elimrec
