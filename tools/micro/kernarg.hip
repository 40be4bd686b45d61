// Does the size of a by-value kernel argument block change a kernel's duration? (dev micro-benchmark)
#include <hip/hip_runtime.h>
#include <cstdio>
template <int N> struct Blob { float *out; int n; int pad[N]; };
template <int N> __global__ __launch_bounds__(256) void k(Blob<N> b) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < b.n) b.out[i] = (float)i;
}
// same, but the kernel reads one word from the END of the block (forces the last kernarg line in)
template <int N> __global__ __launch_bounds__(256) void k2(Blob<N> b) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < b.n) b.out[i] = (float)(i + b.pad[N - 1]);
}
template <int N> void run(float *out, int n, hipStream_t s) {
    Blob<N> b; b.out = out; b.n = n; for (int i = 0; i < N; ++i) b.pad[i] = i;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int form = 0; form < 2; ++form) {
        for (int i = 0; i < 20; ++i) { if (form) hipLaunchKernelGGL(k2<N>, dim3(n / 256), dim3(256), 0, s, b); else hipLaunchKernelGGL(k<N>, dim3(n / 256), dim3(256), 0, s, b); }
        hipStreamSynchronize(s);
        hipEventRecord(e0, s);
        for (int i = 0; i < 200; ++i) { if (form) hipLaunchKernelGGL(k2<N>, dim3(n / 256), dim3(256), 0, s, b); else hipLaunchKernelGGL(k<N>, dim3(n / 256), dim3(256), 0, s, b); }
        hipEventRecord(e1, s); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("kernarg %5zu B, %s: %.2f us per launch (back-to-back, %d workgroups)\n", sizeof(Blob<N>), form ? "reads last word " : "reads first line", ms * 1e3 / 200, n / 256);
    }
}
int main() {
    float *out; const int n = 256 * 2048;
    hipMalloc(&out, n * 4);
    hipStream_t s; hipStreamCreate(&s);
    run<4>(out, n, s); run<60>(out, n, s); run<124>(out, n, s); run<252>(out, n, s); run<508>(out, n, s); run<1000>(out, n, s);
    return 0;
}
