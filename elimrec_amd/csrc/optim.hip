// Dense Adam with coupled L2 (torch.optim.Adam as main.py:49,101 configures it) and the
// CatBackward of the layer-0 table assembly (embedding gradients).
#include "common.h"

namespace elimrec {

__global__ void adam_kernel(float *__restrict__ p, const float *__restrict__ g, float *__restrict__ m,
                            float *__restrict__ v, int64_t n, float step_size, float beta1, float beta2,
                            float inv_sqrt_bc2, float eps, float wd) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float pi = p[i];
        const float gi = fmaf(wd, pi, g[i]);                 // grad.add(param, alpha=weight_decay)
        const float mi = m[i] + (1.f - beta1) * (gi - m[i]);  // exp_avg.lerp_(grad, 1-beta1)
        const float vi = fmaf(1.f - beta2, gi * gi, beta2 * v[i]);
        const float denom = sqrtf(vi) * inv_sqrt_bc2 + eps;
        m[i] = mi;
        v[i] = vi;
        p[i] = pi - step_size * (mi / denom);                // addcdiv_(exp_avg, denom, value=-step_size)
    }
}

// dE_user[u, :] = sum_m G[u, m*d : (m+1)*d];  dE_item[i, :] = G[U+i, 0:d]
__global__ void embed_grad_kernel(const float4 *__restrict__ G, int64_t U, int64_t I, int d4, int M,
                                  float4 *__restrict__ gu, float4 *__restrict__ gi) {
    const int64_t total = (U + I) * d4;
    const int C4 = d4 * M;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = t / d4;
        const int c = (int)(t - row * d4);
        if (row < U) {
            float4 acc = G[row * C4 + c];
            for (int m = 1; m < M; ++m) {
                const float4 x = G[row * C4 + m * d4 + c];
                acc.x += x.x; acc.y += x.y; acc.z += x.z; acc.w += x.w;
            }
            gu[row * d4 + c] = acc;
        } else {
            gi[(row - U) * d4 + c] = G[row * C4 + c];
        }
    }
}

}  // namespace elimrec

using namespace elimrec;

extern "C" int elimrec_adam_step(float *d_p, const float *d_g, float *d_m, float *d_v, int64_t n, float lr,
                                 float beta1, float beta2, float eps, float weight_decay, int64_t step,
                                 void *stream) {
    ELIMREC_REQUIRE(d_p && d_g && d_m && d_v, "adam_step: null pointer");
    ELIMREC_REQUIRE(step >= 1, "adam_step: step is 1-based");
    if (n <= 0) return 0;
    // bias corrections in double on the host, exactly as torch's python scalars
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    const float step_size = (float)((double)lr / bc1);
    const float inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
    int64_t blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, d_p, d_g, d_m, d_v, n,
                       step_size, beta1, beta2, inv_sqrt_bc2, eps, weight_decay);
    ELIMREC_LAUNCH_CHECK("adam_step");
    return 0;
}

extern "C" int elimrec_embed_grad(const float *d_G, int64_t U, int64_t I, int d, int M, float *d_grad_user,
                                  float *d_grad_item, void *stream) {
    ELIMREC_REQUIRE(d_G && d_grad_user && d_grad_item, "embed_grad: null pointer");
    ELIMREC_REQUIRE(d > 0 && d % 4 == 0 && M >= 1, "embed_grad: bad d/M");
    const int64_t total = (U + I) * (d / 4);
    if (total == 0) return 0;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(embed_grad_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                       (const float4 *)d_G, U, I, d / 4, M, (float4 *)d_grad_user, (float4 *)d_grad_item);
    ELIMREC_LAUNCH_CHECK("embed_grad");
    return 0;
}
