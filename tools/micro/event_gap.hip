// Dev micro-benchmark: what an event record / a cross-stream wait between two kernels of one stream costs on this box.
// hipcc --offload-arch=gfx950 -O2 tools/micro/event_gap.hip -o gpurun_out/event_gap && gpurun_out/event_gap
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
__global__ void spin(long long cycles, int *sink) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) {}
    if (sink && threadIdx.x == 12345) *sink = 1;
}
int main() {
    hipStream_t s, aux;
    hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    hipStreamCreateWithFlags(&aux, hipStreamNonBlocking);
    hipEvent_t ev, ev2;
    hipEventCreateWithFlags(&ev, hipEventDisableTiming | hipEventDisableSystemFence);
    hipEventCreateWithFlags(&ev2, hipEventDisableTiming | hipEventDisableSystemFence);
    const long long cyc = 2000;          // wall_clock64 ticks at 100 MHz: 20 us
    const int N = 2000;
    for (int mode = 0; mode < 6; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            hipDeviceSynchronize();
            auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < N; ++i) {
                if (mode == 2) hipExtLaunchKernelGGL(spin, dim3(256), dim3(256), 0, s, nullptr, ev, 0, cyc, (int *)nullptr);
                else hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, s, cyc, (int *)nullptr);
                if (mode == 1 || mode == 3) hipEventRecord(ev, s);                    // record between the two kernels
                if (mode == 3 || mode == 2) hipStreamWaitEvent(aux, ev, 0);           // ... that another stream waits for
                if (mode == 3 || mode == 2) hipLaunchKernelGGL(spin, dim3(8), dim3(64), 0, aux, 200LL, (int *)nullptr);
                if (mode == 4 || mode == 5) {                                         // join: main waits for an event of aux
                    hipLaunchKernelGGL(spin, dim3(8), dim3(64), 0, aux, 200LL, (int *)nullptr);
                    if (mode == 4) hipEventRecord(ev2, aux);
                    else hipExtLaunchKernelGGL(spin, dim3(8), dim3(64), 0, aux, nullptr, ev2, 0, 100LL, (int *)nullptr);
                    hipStreamWaitEvent(s, ev2, 0);
                }
                hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, s, cyc, (int *)nullptr);
            }
            hipDeviceSynchronize();
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / N;
            if (rep) printf("mode %d: %.2f us per pair of 20-us kernels (%s)\n", mode, us,
                            mode == 0 ? "back to back" : mode == 1 ? "hipEventRecord between" : mode == 2 ? "stop event on the first kernel, aux waits"
                            : mode == 3 ? "hipEventRecord between, aux waits" : mode == 4 ? "main waits for aux's recorded event" : "main waits for aux's stop event");
        }
    }
    return 0;
}
