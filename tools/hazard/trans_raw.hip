// Does a VALU write to a VGPR that a transcendental op issued just before still reads (quarter-rate: lanes 48-63 last)
// corrupt the transcendental's result? Raw instruction sequences in inline asm (the compiler's hazard recognizer does not
// look inside): GAP independent VALU instructions between the second v_exp_f32 and the overwriting instruction.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>

template <int KIND, int GAP>
__global__ void probe(const float *in, float *out, int iters) {
    const int t = threadIdx.x + blockIdx.x * blockDim.x;
    float a = in[t], b = in[t + 1], r0 = 0.f, r1 = 0.f;
    for (int i = 0; i < iters; ++i) {
        float e0, e1;
        asm volatile(
            "v_mov_b32 v20, %2\n"
            "v_mov_b32 v21, %3\n"
            "v_mov_b32 v24, 0x40a00000\n"          // 5.0
            "v_mov_b32 v25, 0x40a00000\n"
            "v_mov_b32 v30, 1.0\n v_mov_b32 v31, 1.0\n"
            "s_nop 7\n"
            "v_exp_f32 v22, v20\n"
            "v_exp_f32 v23, v21\n"
            ".if %4 >= 1\n v_add_f32 v26, v24, v24\n .endif\n"
            ".if %4 >= 2\n v_add_f32 v27, v24, v24\n .endif\n"
            ".if %4 >= 3\n v_add_f32 v28, v24, v24\n .endif\n"
            ".if %4 >= 4\n v_add_f32 v29, v24, v24\n .endif\n"
                        ".if %5 == 0\n v_pk_mul_f32 v[22:23], v[22:23], v[30:31]\n .endif\n"      // packed fp32 consumes both results (RAW)
            ".if %5 == 1\n v_mul_f32 v23, v23, v30\n .endif\n"                         // plain VALU consumes the 2nd
            ".if %5 == 2\n v_pk_fma_f32 v[22:23], v[22:23], v[30:31], v[22:23] neg_lo:[0,0,1] neg_hi:[0,0,1]\n v_pk_add_f32 v[22:23], v[22:23], v[22:23]\n .endif\n"
            "s_nop 7\n"
            "s_nop 7\n"
            "v_mov_b32 %0, v22\n"
            "v_mov_b32 %1, v23\n"
            : "=v"(e0), "=v"(e1) : "v"(a), "v"(b), "n"(GAP), "n"(KIND)
            : "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31");
        r0 += e0; r1 += e1;
        a += 0.f; b += 0.f;
    }
    out[2 * t] = r0; out[2 * t + 1] = r1;
}

template <int KIND, int GAP>
static void run(const char *name, float *d_in, float *d_out, int n, int iters) {
    hipLaunchKernelGGL((probe<KIND, GAP>), dim3(n / 256), dim3(256), 0, 0, d_in, d_out, iters);
    std::vector<float> out(2 * n), in(n + 1);
    hipMemcpy(out.data(), d_out, sizeof(float) * 2 * n, hipMemcpyDeviceToHost);
    hipMemcpy(in.data(), d_in, sizeof(float) * (n + 1), hipMemcpyDeviceToHost);
    long bad[4] = {0, 0, 0, 0};
    for (int t = 0; t < n; ++t) {
        float w0 = iters * exp2f(in[t]), w1 = iters * exp2f(in[t + 1]);
        if (KIND == 2) { if (out[2 * t] != 0.f || out[2 * t + 1] != 0.f) bad[(t & 63) >> 4]++; continue; }
        if (fabsf(out[2 * t] - w0) > 1e-3f * w0 || fabsf(out[2 * t + 1] - w1) > 1e-3f * w1) bad[(t & 63) >> 4]++;
    }
    printf("%-12s gap %d: wrong lanes by quarter [0-15 | 16-31 | 32-47 | 48-63] = %ld %ld %ld %ld\n", name, GAP, bad[0], bad[1], bad[2], bad[3]);
}

int main() {
    const int n = 256 * 1024, iters = 64;
    std::vector<float> in(n + 1);
    for (int i = 0; i <= n; ++i) in[i] = -1.f + 2.f * (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.f;
    float *d_in, *d_out;
    hipMalloc(&d_in, sizeof(float) * (n + 1)); hipMalloc(&d_out, sizeof(float) * 2 * n);
    hipMemcpy(d_in, in.data(), sizeof(float) * (n + 1), hipMemcpyHostToDevice);
    run<0, 0>("pk_mul", d_in, d_out, n, iters); run<0, 1>("pk_mul", d_in, d_out, n, iters); run<0, 2>("pk_mul", d_in, d_out, n, iters);
    run<0, 3>("pk_mul", d_in, d_out, n, iters); run<0, 4>("pk_mul", d_in, d_out, n, iters);
    run<2, 0>("pk_fma", d_in, d_out, n, iters); run<2, 1>("pk_fma", d_in, d_out, n, iters); run<2, 2>("pk_fma", d_in, d_out, n, iters);
    run<2, 3>("pk_fma", d_in, d_out, n, iters); run<2, 4>("pk_fma", d_in, d_out, n, iters);
    run<1, 0>("v_mul", d_in, d_out, n, iters); run<1, 1>("v_mul", d_in, d_out, n, iters); run<1, 2>("v_mul", d_in, d_out, n, iters);
    run<1, 3>("v_mul", d_in, d_out, n, iters); run<1, 4>("v_mul", d_in, d_out, n, iters);
    return 0;
}
