// Stand-alone repeat-launch harness for the evaluator's chunked top-K call (elimrec_score_topk), used while chasing the
// intermittent wrong tile of score_t16b_kernel<2, 4, 2, 1, 64>. It dlopens ONE build of the library (so
// that several builds of eval.hip can be compared in one GPU session), runs the same top-20 call `reps` times on workspaces
// refilled with a byte pattern that changes every launch, and reports every launch whose (ids, scores) differ from the first.
//   hipcc -O2 scorer_repro.cpp -o scorer_repro -ldl && ./scorer_repro <libelimrec_hip.so> [reps] [mode 0|1|2] [ptype 0|1|2]
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

typedef size_t (*ws_for_t)(int, int64_t, int64_t, int, int, int, int);
typedef int (*topk_t)(const float *, int64_t, int64_t, int64_t, const int64_t *, int, int, int, uint32_t, int, int, const float *,
                      const int64_t *, const int32_t *, float *, int64_t, int, int32_t *, float *, void *, size_t, void *);
typedef void (*set_t)(int);
typedef int (*haz_t)(unsigned *, float *, int);
typedef const char *(*err_t)(void);

int main(int argc, char **argv) {
    if (argc < 2) { printf("usage: %s lib.so [reps] [mode] [ptype] [d]\n", argv[0]); return 2; }
    const int reps = argc > 2 ? atoi(argv[2]) : 400, mode = argc > 3 ? atoi(argv[3]) : 1, ptype = argc > 4 ? atoi(argv[4]) : 2;
    const int d = argc > 5 ? atoi(argv[5]) : 64;
    void *h = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL);
    if (!h) { printf("dlopen: %s\n", dlerror()); return 2; }
    ws_for_t ws_for = (ws_for_t)dlsym(h, "elimrec_score_workspace_for");
    topk_t topk = (topk_t)dlsym(h, "elimrec_score_topk");
    set_t set_math = (set_t)dlsym(h, "elimrec_score_set_math"), set_b3 = (set_t)dlsym(h, "elimrec_score_set_bf16x3");
    haz_t haz = (haz_t)dlsym(h, "elimrec_haz_read");
    err_t last_error = (err_t)dlsym(h, "elimrec_last_error");
    if (!ws_for || !topk || !set_math || !set_b3) { printf("missing symbols\n"); return 2; }
    const int U = 300, S = 3, K = 20, B = 200;
    const int64_t I = 40000;
    const int C = (1 + S) * d;
    std::mt19937 rng(d);
    std::normal_distribution<float> nd(0.f, 0.4f);
    std::vector<float> Y((size_t)(U + I) * C);
    for (auto &v : Y) v = nd(rng);
    std::vector<int64_t> users(B);
    { std::vector<int> perm(U); for (int i = 0; i < U; ++i) perm[i] = i; std::shuffle(perm.begin(), perm.end(), rng); for (int b = 0; b < B; ++b) users[b] = perm[b]; }
    std::vector<int64_t> ptr(B + 1, 0);
    std::vector<int32_t> items;
    for (int b = 0; b < B; ++b) {
        const int n = (int)(rng() % 50);
        std::vector<int32_t> l;
        for (int j = 0; j < n; ++j) l.push_back((int32_t)(rng() % I));
        std::sort(l.begin(), l.end()); l.erase(std::unique(l.begin(), l.end()), l.end());
        items.insert(items.end(), l.begin(), l.end());
        ptr[b + 1] = (int64_t)items.size();
    }
    if (items.empty()) items.push_back(0);
    float *dY; int64_t *dU, *dP; int32_t *dI;
    CK(hipMalloc(&dY, Y.size() * 4)); CK(hipMemcpy(dY, Y.data(), Y.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&dU, B * 8)); CK(hipMemcpy(dU, users.data(), B * 8, hipMemcpyHostToDevice));
    CK(hipMalloc(&dP, (B + 1) * 8)); CK(hipMemcpy(dP, ptr.data(), (B + 1) * 8, hipMemcpyHostToDevice));
    CK(hipMalloc(&dI, items.size() * 4)); CK(hipMemcpy(dI, items.data(), items.size() * 4, hipMemcpyHostToDevice));
    set_math(1); set_b3(1);
    const size_t nbytes = ws_for(B, U, I, S, K, d, 0);
    void *ws[2]; CK(hipMalloc(&ws[0], nbytes)); CK(hipMalloc(&ws[1], nbytes));
    // argv[6]: 0 = a host synchronisation after every launch (results copied back one by one), 1 = all launches enqueued back to
    // back on the NULL stream with fresh workspaces (what a torch process does), results kept on the device and compared at the end
    const int async = argc > 6 ? atoi(argv[6]) : 1;
    hipStream_t st = nullptr;
    if (!async) CK(hipStreamCreate(&st));
    int32_t *allIdx; float *allVal;
    CK(hipMalloc(&allIdx, (size_t)reps * B * K * 4)); CK(hipMalloc(&allVal, (size_t)reps * B * K * 4));
    for (int rep = 0; rep < reps; ++rep) {
        void *w = ws[rep & 1];
        if (!async) CK(hipMemsetAsync(w, rep % 3 == 0 ? 0xFF : (rep % 3 == 1 ? 0x00 : 0x7F), nbytes, st));     // NaN / zero / 3.4e38 fill
        const int rc = topk(dY, C, U, I, dU, B, d, S, 7u, mode, ptype, nullptr, dP, dI, nullptr, 0, K, allIdx + (size_t)rep * B * K,
                            allVal + (size_t)rep * B * K, w, nbytes, st);
        if (rc) { printf("rc %d: %s\n", rc, last_error ? last_error() : "?"); return 2; }
        if (!async) CK(hipStreamSynchronize(st));
    }
    CK(hipDeviceSynchronize());
    std::vector<int32_t> hI((size_t)reps * B * K); std::vector<float> hV((size_t)reps * B * K);
    CK(hipMemcpy(hI.data(), allIdx, hI.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hV.data(), allVal, hV.size() * 4, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int rep = 1; rep < reps; ++rep) {
        const int32_t *idx = &hI[(size_t)rep * B * K]; const float *val = &hV[(size_t)rep * B * K];
        const int32_t *idx0 = &hI[0]; const float *val0 = &hV[0];
        if (memcmp(idx, idx0, B * K * 4) || memcmp(val, val0, B * K * 4)) {
            ++bad;
            if (bad <= 6)
                for (int b = 0; b < B; ++b)
                    if (memcmp(&idx[b * K], &idx0[b * K], K * 4) || memcmp(&val[b * K], &val0[b * K], K * 4)) {
                        printf("  rep %d user row %d (wave %d, row-in-wave %d):", rep, b, (b % 128) / 16, b % 16);
                        for (int k = 0; k < 4; ++k) printf(" (%d, %.9g | first %d, %.9g)", idx[b * K + k], val[b * K + k], idx0[b * K + k], val0[b * K + k]);
                        printf("\n");
                    }
        }
    }
    printf("%s: mode %d ptype %d d %d async %d: %d of %d launches differ from the first\n", argv[1], mode, ptype, d, async, bad, reps - 1);
    if (haz) {
        unsigned cnt = 0; std::vector<float> rec(4096 * 16);
        haz(&cnt, rec.data(), 1);
        printf("  detector: %u scores of exactly 1.0\n", cnt);
        for (unsigned i = 0; i < cnt && i < 40; ++i) {
            const float *r = &rec[i * 16];
            printf("  bx %g by %g tid %g (wave %d lane %d) r %g tile %g first %g | umean(LDS now) %.9g row_mean(global) %.9g | acc %.6g %.6g %.6g %.6g | unorm %.6g %.6g inorm %.6g cur %g\n",
                   r[0], r[1], r[2], (int)r[2] / 64, (int)r[2] % 64, r[3], r[4], r[5], r[6], r[7], r[8], r[9], r[10], r[11], r[12], r[13], r[14], r[15]);
        }
    }
    return bad ? 1 : 0;
}
