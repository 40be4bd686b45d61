// The folded propagation for adjacencies WITH a diagonal (adj_type = norm: D^-1 (A + I); the reference's fall-through
// branch mean + I: /root/reference/models/EliMRec.py:332-335,349-352).
//
// With a bipartite A-hat the part of every table that comes from E_u alone and the part that comes from E_i alone sit in
// alternating layers (users at even k / items at odd k), so ONE [N x d] table carries both (slab.hip, elimrec_slab_rows). A
// diagonal mixes them on every row. The folded algebra itself does not need the parity:
//     out_0 = mean_k A^k [E_u ; 0] + mean_k A^k [0 ; E_i],      out_m = mean_k A^k [E_u ; 0] + S_m W_m^T + c b_m^T  (m >= 1)
// so the graph carries TWO column blocks side by side -- a "wide" slab table [N x 2 dl]: left = the E_u-borne part, right =
// the E_i-borne part -- through the same hop kernels; layer 0 is [E_u | 0] on user rows and [0 | E_i] on item rows. The
// adjoint's source is the same at every layer, [H | G] (H = block sum of dOut -> left, G = dOut's block 0 -> right), and
// the parameters' gradient is the left half on user rows and the right half on item rows.
// Slab-major layout as everywhere (include/elimrec_hip.h): the wide table has 2 ns slabs of w floats, slabs [0, ns) = left.
#include "common.h"

namespace elimrec {

__global__ __launch_bounds__(256) void wide_from_master_kernel(const float4 *__restrict__ master, int64_t U, int64_t N, int ns, int w4,
                                                               float4 *__restrict__ wide) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;        // one float4 of the narrow table
    const int64_t per_slab = N * w4;
    if (t >= (int64_t)ns * per_slab) return;
    const int64_t row = (t % per_slab) / w4;
    const float4 v = master[t], z = make_float4(0.f, 0.f, 0.f, 0.f);
    const bool user = row < U;
    wide[t] = user ? v : z;                                            // left slabs: the E_u-borne part
    wide[(int64_t)ns * per_slab + t] = user ? z : v;                   // right slabs: the E_i-borne part
}

constexpr int kWideMaxLayers = 8;
struct WideRowsArgs {
    const float4 *x[kWideMaxLayers + 1];
    int L;
    int64_t N;
    int ns, w4;
    const int32_t *rows;
    int64_t total;
    float *out0; int64_t ld_out0;
    float *narrow; int64_t ld_narrow;
    float inv;
};

// out0[r] = inv * sum_k (left_k + right_k)[row], narrow[r] = inv * sum_k left_k[row]; a lane per float4 column of the row
__global__ __launch_bounds__(256) void wide_rows_kernel(WideRowsArgs a) {
    const int nc4 = a.ns * a.w4;
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= a.total * nc4) return;
    const int64_t r = t / nc4;
    const int c = (int)(t % nc4);
    const int64_t row = a.rows ? (int64_t)a.rows[r] : r;
    if (row < 0) return;                                               // padding of a gathered list
    const int slab = c / a.w4, c4 = c % a.w4;
    const int64_t li = ((int64_t)slab * a.N + row) * a.w4 + c4, ri = li + (int64_t)a.ns * a.N * a.w4;
    float4 l = make_float4(0.f, 0.f, 0.f, 0.f), s = l;
    for (int k = 0; k <= a.L; ++k) {
        const float4 x = a.x[k][li], y = a.x[k][ri];
        l.x += x.x; l.y += x.y; l.z += x.z; l.w += x.w;
        s.x += x.x + y.x; s.y += x.y + y.y; s.z += x.z + y.z; s.w += x.w + y.w;
    }
    *reinterpret_cast<float4 *>(a.out0 + r * a.ld_out0 + 4 * c) = make_float4(s.x * a.inv, s.y * a.inv, s.z * a.inv, s.w * a.inv);
    *reinterpret_cast<float4 *>(a.narrow + r * a.ld_narrow + 4 * c) = make_float4(l.x * a.inv, l.y * a.inv, l.z * a.inv, l.w * a.inv);
}

__global__ __launch_bounds__(256) void wide_grad_kernel(const float4 *__restrict__ gw, int64_t U, int64_t N, int ns, int w4, float scale,
                                                        float4 *__restrict__ grad) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t per_slab = N * w4;
    if (t >= (int64_t)ns * per_slab) return;
    const int64_t row = (t % per_slab) / w4;
    const float4 v = row < U ? gw[t] : gw[(int64_t)ns * per_slab + t];
    grad[t] = make_float4(v.x * scale, v.y * scale, v.z * scale, v.w * scale);
}

}  // namespace elimrec

using namespace elimrec;

static int wide_geometry(const char *what, int ns, int w) {
    ELIMREC_REQUIRE(ns >= 1 && w >= 4 && w % 4 == 0, "%s: bad slab geometry (%d x %d)", what, ns, w);
    return 0;
}

extern "C" int elimrec_wide_from_master(const float *d_master, int64_t U, int64_t N, int ns, int w, float *d_wide, void *stream) {
    ELIMREC_REQUIRE(d_master && d_wide && U >= 0 && U <= N, "wide_from_master: bad arguments");
    int rc = wide_geometry("wide_from_master", ns, w);
    if (rc) return rc;
    const int64_t n4 = (int64_t)ns * N * (w / 4);
    if (n4 == 0) return 0;
    hipLaunchKernelGGL(wide_from_master_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const float4 *)d_master, U,
                       N, ns, w / 4, (float4 *)d_wide);
    ELIMREC_LAUNCH_CHECK("wide_from_master");
    return 0;
}

extern "C" int elimrec_wide_rows(const float *const *layers, int L, int64_t N, int ns, int w, const int32_t *d_rows, int64_t total,
                                 float *d_out0, int64_t ld_out0, float *d_narrow, int64_t ld_narrow, void *stream) {
    ELIMREC_REQUIRE(layers && d_out0 && d_narrow && L >= 1 && L <= kWideMaxLayers, "wide_rows: 1 <= L <= %d", kWideMaxLayers);
    ELIMREC_REQUIRE(ld_out0 % 4 == 0 && ld_narrow % 4 == 0, "wide_rows: leading dimensions must be multiples of 4");
    int rc = wide_geometry("wide_rows", ns, w);
    if (rc) return rc;
    if (total <= 0) return 0;
    WideRowsArgs a = {};
    for (int k = 0; k <= L; ++k) { ELIMREC_REQUIRE(layers[k], "wide_rows: layer table %d missing", k); a.x[k] = (const float4 *)layers[k]; }
    a.L = L; a.N = N; a.ns = ns; a.w4 = w / 4; a.rows = d_rows; a.total = total;
    a.out0 = d_out0; a.ld_out0 = ld_out0; a.narrow = d_narrow; a.ld_narrow = ld_narrow; a.inv = 1.0f / (float)(L + 1);
    const int64_t threads = total * ns * (w / 4);
    hipLaunchKernelGGL(wide_rows_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
    ELIMREC_LAUNCH_CHECK("wide_rows");
    return 0;
}

extern "C" int elimrec_wide_grad(const float *d_wide_grad, int64_t U, int64_t N, int ns, int w, float scale, float *d_grad, void *stream) {
    ELIMREC_REQUIRE(d_wide_grad && d_grad && U >= 0 && U <= N, "wide_grad: bad arguments");
    int rc = wide_geometry("wide_grad", ns, w);
    if (rc) return rc;
    const int64_t n4 = (int64_t)ns * N * (w / 4);
    if (n4 == 0) return 0;
    hipLaunchKernelGGL(wide_grad_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const float4 *)d_wide_grad, U, N,
                       ns, w / 4, scale, (float4 *)d_grad);
    ELIMREC_LAUNCH_CHECK("wide_grad");
    return 0;
}
