// On-device pairwise (BPR) sampler: the epoch-level contract of PairwiseSamplerV2
// (data/sampler.py:93-126,297-351; util/cython/random_choice.pyx:20-62) with a counter-based
// generator instead of libc rand().
#include "common.h"

namespace elimrec {

struct Philox {
    uint32_t c[4];
    uint32_t k[2];
    __device__ static inline void mulhilo(uint32_t a, uint32_t b, uint32_t &hi, uint32_t &lo) {
        const uint64_t p = (uint64_t)a * b;
        hi = (uint32_t)(p >> 32);
        lo = (uint32_t)p;
    }
    // Philox4x32-10 (Salmon et al., SC'11)
    __device__ inline void generate(uint32_t out[4]) const {
        uint32_t c0 = c[0], c1 = c[1], c2 = c[2], c3 = c[3], k0 = k[0], k1 = k[1];
#pragma unroll
        for (int r = 0; r < 10; ++r) {
            uint32_t h0, l0, h1, l1;
            mulhilo(0xD2511F53u, c0, h0, l0);
            mulhilo(0xCD9E8D57u, c2, h1, l1);
            const uint32_t n0 = h1 ^ c1 ^ k0, n1 = l1, n2 = h0 ^ c3 ^ k1, n3 = l0;
            c0 = n0; c1 = n1; c2 = n2; c3 = n3;
            k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
        }
        out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
    }
};

__global__ void sample_triplets_kernel(const int32_t *__restrict__ user_ids, const int64_t *__restrict__ ptr,
                                       const int32_t *__restrict__ items, int64_t n_train_users, int64_t I, int64_t n,
                                       uint64_t seed, uint64_t epoch, int64_t *__restrict__ users,
                                       int64_t *__restrict__ pos, int64_t *__restrict__ neg) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Philox ph;
    ph.k[0] = (uint32_t)seed; ph.k[1] = (uint32_t)(seed >> 32);
    ph.c[0] = (uint32_t)i; ph.c[1] = (uint32_t)((uint64_t)i >> 32);
    ph.c[2] = (uint32_t)epoch; ph.c[3] = (uint32_t)(epoch >> 32) & 0x00FFFFFFu;   // top byte = draw round
    uint32_t r[4];
    ph.generate(r);
    const uint64_t ru = ((uint64_t)r[0] << 32) | r[1];
    const uint64_t rp = ((uint64_t)r[2] << 32) | r[3];
    const int64_t ui = (int64_t)(ru % (uint64_t)n_train_users);
    const int64_t beg = ptr[ui], end = ptr[ui + 1];
    const int64_t cnt = end - beg;
    users[i] = user_ids[ui];
    pos[i] = items[beg + (int64_t)(rp % (uint64_t)cnt)];
    // negatives: uniform over [0, I), rejected while in the user's (sorted) training items
    int64_t cand = -1;
    for (uint32_t round = 1; round < 256 && cand < 0; ++round) {
        ph.c[3] = ((uint32_t)(epoch >> 32) & 0x00FFFFFFu) | (round << 24);
        ph.generate(r);
#pragma unroll
        for (int t = 0; t < 2 && cand < 0; ++t) {
            const uint64_t x = ((uint64_t)r[2 * t] << 32) | r[2 * t + 1];
            const int32_t a = (int32_t)(x % (uint64_t)I);
            int64_t lo = beg, hi = end;
            bool found = false;
            while (lo < hi) {
                const int64_t mid = (lo + hi) >> 1;
                const int32_t v = items[mid];
                if (v == a) { found = true; break; }
                if (v < a) lo = mid + 1; else hi = mid;
            }
            if (!found) cand = a;
        }
    }
    if (cand < 0) {   // a user who interacted with almost every item: the first id missing from the sorted list
        int64_t k = 0;
        while (k < cnt && items[beg + k] == (int32_t)k) ++k;
        cand = k;      // < I: the host rejects users with >= I training items
    }
    neg[i] = cand;
}

}  // namespace elimrec

using namespace elimrec;

extern "C" int elimrec_sample_triplets(const int32_t *d_user_ids, const int64_t *d_ptr, const int32_t *d_items,
                                       int64_t n_train_users, int64_t I, int64_t n, uint64_t seed, uint64_t epoch,
                                       int64_t *d_users, int64_t *d_pos, int64_t *d_neg, void *stream) {
    ELIMREC_REQUIRE(d_user_ids && d_ptr && d_items && d_users && d_pos && d_neg, "sample_triplets: null pointer");
    ELIMREC_REQUIRE(n_train_users > 0 && I > 0, "sample_triplets: 'user_pos_dict' cannot be empty.");
    if (n <= 0) return 0;
    hipLaunchKernelGGL(sample_triplets_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       d_user_ids, d_ptr, d_items, n_train_users, I, n, seed, epoch, d_users, d_pos, d_neg);
    ELIMREC_LAUNCH_CHECK("sample_triplets");
    return 0;
}
