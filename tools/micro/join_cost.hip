// What a cross-stream join costs the WAITING stream when the other stream finished long ago (the training step's joins, DESIGN.md
// section 3): a chain of short kernels on stream A with, between every two of them, (0) nothing, (1) a wait for an event recorded
// on stream B behind a kernel there (default event), (2) the same with hipEventDisableSystemFence, (3) hipStreamWaitValue32 on a
// value stream B wrote with hipStreamWriteValue32 (signal memory), (4) only a RECORD on stream A (what a fork costs its stream).
//   hipcc -O2 --offload-arch=gfx950 join_cost.hip -o join_cost && timeout 60 ./join_cost          (dev tool)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void spin(float *p, int n) {
    float v = p[threadIdx.x];
    for (int i = 0; i < n; ++i) v = v * 1.0001f + 0.5f;
    p[threadIdx.x] = v;
}

int main() {
    hipStream_t A, B;
    CK(hipStreamCreate(&A)); CK(hipStreamCreate(&B));
    float *pa, *pb;
    CK(hipMalloc(&pa, 4096)); CK(hipMalloc(&pb, 4096));
    CK(hipMemset(pa, 0, 4096)); CK(hipMemset(pb, 0, 4096));
    int can = 0;
    CK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
    uint32_t *sig = nullptr;
    if (can) CK(hipExtMallocWithFlags((void **)&sig, 8, hipMallocSignalMemory));
    const int N = 300, SPIN = 600;
    hipEvent_t t0, t1;
    CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
    std::vector<hipEvent_t> ev(N), evn(N);
    for (int i = 0; i < N; ++i) {
        CK(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming));
        CK(hipEventCreateWithFlags(&evn[i], hipEventDisableTiming | hipEventDisableSystemFence));
    }
    const char *names[5] = {"back to back", "wait(default event)", "wait(event, no system fence)", "hipStreamWaitValue32", "record on the chain's own stream"};
    for (int mode = 0; mode < 5; ++mode) {
        if (mode == 3 && !can) { printf("%-36s not supported on this device\n", names[mode]); continue; }
        for (int rep = 0; rep < 2; ++rep) {
            // enqueued interleaved, as a host that runs ahead of the device does: when the wait is ENQUEUED its event is still pending
            // (a wait for an event that is already complete is dropped by the runtime), when stream A REACHES it the event has
            // long been signalled -- stream B's kernels are a tenth of stream A's
            if (mode == 3) CK(hipMemset(sig, 0, 8));
            CK(hipDeviceSynchronize());
            hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, A, pa, 200 * SPIN);          // a head start for the host
            CK(hipEventRecord(t0, A));
            for (int i = 0; i < N; ++i) {
                hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, B, pb, SPIN / 10);
                if (mode == 1) CK(hipEventRecord(ev[i], B));
                if (mode == 2) CK(hipEventRecord(evn[i], B));
                if (mode == 3) CK(hipStreamWriteValue32(B, sig, (uint32_t)(i + 1), 0));
                hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, A, pa, SPIN);
                if (mode == 1) CK(hipStreamWaitEvent(A, ev[i], 0));
                if (mode == 2) CK(hipStreamWaitEvent(A, evn[i], 0));
                if (mode == 3) CK(hipStreamWaitValue32(A, sig, (uint32_t)(i + 1), hipStreamWaitValueGte, 0xFFFFFFFFu));
                if (mode == 4) CK(hipEventRecord(evn[i], A));
            }
            CK(hipEventRecord(t1, A));
            CK(hipDeviceSynchronize());
            float ms = 0;
            CK(hipEventElapsedTime(&ms, t0, t1));
            if (rep == 1) printf("%-36s %7.2f us per link\n", names[mode], ms * 1e3 / N);
        }
    }
    return 0;
}
