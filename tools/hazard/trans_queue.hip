// N back-to-back transcendental ops, GAP independent VALU instructions, then a consumer of the LAST result: is one wait
// state (what the compiler leaves) enough when the transcendental pipe is backed up?  KIND 0: v_mul_f32 consumer,
// 1: v_pk_mul_f32 consumer of the last two results, 2: v_pk_fma_f32 consumer.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>

template <int KIND, int NT, int GAP>
__global__ void probe(const float *in, float *out, int iters) {
    const int t = threadIdx.x + blockIdx.x * blockDim.x;
    float a = in[t], r0 = 0.f;
    for (int i = 0; i < iters; ++i) {
        float e0;
        asm volatile(
            "v_mov_b32 v20, %1\n"
            "v_mov_b32 v30, 1.0\n v_mov_b32 v31, 1.0\n v_mov_b32 v24, 2.0\n"
            "v_mov_b32 v22, 0\n v_mov_b32 v23, 0\n"
            "s_nop 7\n"
            ".if %2 >= 6\n v_exp_f32 v40, v20\n .endif\n"
            ".if %2 >= 5\n v_exp_f32 v41, v20\n .endif\n"
            ".if %2 >= 4\n v_exp_f32 v42, v20\n .endif\n"
            ".if %2 >= 3\n v_exp_f32 v43, v20\n .endif\n"
            ".if %2 >= 2\n v_exp_f32 v22, v20\n .endif\n"
            "v_exp_f32 v23, v20\n"
            ".if %3 >= 1\n v_add_f32 v26, v24, v24\n .endif\n"
            ".if %3 >= 2\n v_add_f32 v27, v24, v24\n .endif\n"
            ".if %3 >= 3\n v_add_f32 v28, v24, v24\n .endif\n"
            ".if %4 == 0\n v_mul_f32 v23, v23, v30\n .endif\n"
            ".if %4 == 1\n v_pk_mul_f32 v[22:23], v[22:23], v[30:31]\n .endif\n"
            ".if %4 == 2\n v_pk_fma_f32 v[22:23], v[22:23], v[30:31], v[22:23]\n .endif\n"
            "s_nop 7\n s_nop 7\n"
            "v_mov_b32 %0, v23\n"
            : "=v"(e0) : "v"(a), "n"(NT), "n"(GAP), "n"(KIND)
            : "v20", "v22", "v23", "v24", "v26", "v27", "v28", "v30", "v31", "v40", "v41", "v42", "v43");
        r0 += e0;
        a += 0.f;
    }
    out[t] = r0;
}

template <int KIND, int NT, int GAP>
static void run(const char *name, float *d_in, float *d_out, int n, int iters) {
    hipLaunchKernelGGL((probe<KIND, NT, GAP>), dim3(n / 256), dim3(256), 0, 0, d_in, d_out, iters);
    std::vector<float> out(n), in(n);
    hipMemcpy(out.data(), d_out, sizeof(float) * n, hipMemcpyDeviceToHost);
    hipMemcpy(in.data(), d_in, sizeof(float) * n, hipMemcpyDeviceToHost);
    long bad[4] = {0, 0, 0, 0};
    for (int t = 0; t < n; ++t) {
        const float w = iters * exp2f(in[t]) * (KIND == 2 ? 2.f : 1.f);
        if (!(fabsf(out[t] - w) <= 1e-3f * w)) bad[(t & 63) >> 4]++;
    }
    printf("%-8s trans x%d gap %d: wrong lanes by quarter = %ld %ld %ld %ld\n", name, NT, GAP, bad[0], bad[1], bad[2], bad[3]);
}

int main() {
    const int n = 256 * 2048, iters = 64;
    std::vector<float> in(n);
    for (int i = 0; i < n; ++i) in[i] = -1.f + 2.f * (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.f;
    float *d_in, *d_out;
    hipMalloc(&d_in, sizeof(float) * n); hipMalloc(&d_out, sizeof(float) * n);
    hipMemcpy(d_in, in.data(), sizeof(float) * n, hipMemcpyHostToDevice);
#define ROW(K, NAME) run<K, 1, 1>(NAME, d_in, d_out, n, iters); run<K, 2, 1>(NAME, d_in, d_out, n, iters); run<K, 3, 1>(NAME, d_in, d_out, n, iters); \
    run<K, 4, 1>(NAME, d_in, d_out, n, iters); run<K, 6, 1>(NAME, d_in, d_out, n, iters); run<K, 6, 2>(NAME, d_in, d_out, n, iters); run<K, 6, 3>(NAME, d_in, d_out, n, iters);
    ROW(0, "v_mul") ROW(1, "pk_mul") ROW(2, "pk_fma")
    return 0;
}
