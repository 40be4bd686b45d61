// Row-sharded constant tables with an id lookup (the "all-to-all index lookup" of the row partition).
//
// What is sharded: the folded constants S_m = mean_k A^k [0 ; F_m] ([N x D_m], the propagated form of the V/A/T feature
// tables of /root/reference/models/EliMRec.py:233-236,366-381) and c = mean_k A^k [0 ; 1] -- parameter-free, read at the
// batch's <= 3B active rows only, and by far the largest tables of the model (N x sum(D_m)). Rank o owns the rows of the
// users [ub[o], ub[o+1]) and of the items [ib[o], ib[o+1]) ("GPU g owns users [gU/8,(g+1)U/8) and items likewise"),
// stored as ONE local table [own users ; own items] x row_elems in fp32, fp16 or bf16: columns [0, sumD) = the S_m side
// by side, then c (fp32: one column; 16-bit storage: c = hi + lo in two columns, 22 bits), zero padding up to a multiple
// of 16 bytes.
//
// A step, per rank (ids of every rank's active rows are all-gathered anyway, shard.py):
//   lookup_counts   counts[r][o] = how many of rank r's active rows rank o owns  -> the split sizes of the exchange
//   lookup_pack     owner side: my rows of every requester's list, requester by requester (users, then items, ascending)
//   (all_to_all with those split sizes: RCCL over xGMI -- torch.distributed, shard.py)
//   lookup_unpack   requester side: received rows -> compact fp32 [R x sumD] in ACTIVE-ROW order + c [R]; the head kernels
//                   then read row r of the compact table where they read row act[r] of the replicated one.
// Active-row lists are sorted ascending with negative padding behind the valid prefix (elimrec_batch_plan), so every
// (list, owner) pair is two contiguous ranges found by binary search: no atomics, no sort, deterministic layout.
#include "common.h"
#include <hip/hip_fp16.h>
#include <hip/hip_bf16.h>

namespace elimrec {

struct OwnerMap {
    int world;
    int64_t U;
    int64_t ub[ELIMREC_MAX_RANKS + 1];
    int64_t ib[ELIMREC_MAX_RANKS + 1];
};

// first index in [0, n) whose entry is padding (< 0) or >= key: the valid prefix is ascending, the padding sits behind it
__device__ __forceinline__ int first_at_least(const int32_t *__restrict__ a, int n, int64_t key) {
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        const int32_t v = a[mid];
        if (v < 0 || (int64_t)v >= key) hi = mid; else lo = mid + 1;
    }
    return lo;
}

struct Ranges { int ulo, ucnt, ilo, icnt; };
__device__ __forceinline__ Ranges owner_ranges(const int32_t *__restrict__ list, int R, const OwnerMap &m, int o) {
    Ranges g;
    g.ulo = first_at_least(list, R, m.ub[o]);
    g.ucnt = first_at_least(list, R, m.ub[o + 1]) - g.ulo;
    g.ilo = first_at_least(list, R, m.U + m.ib[o]);
    g.icnt = first_at_least(list, R, m.U + m.ib[o + 1]) - g.ilo;
    return g;
}

__global__ __launch_bounds__(256) void lookup_counts_kernel(const int32_t *__restrict__ acts, int R, OwnerMap m,
                                                            int32_t *__restrict__ counts) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= m.world * m.world) return;
    const int r = t / m.world, o = t % m.world;
    const Ranges g = owner_ranges(acts + (int64_t)r * R, R, m, o);
    counts[t] = g.ucnt + g.icnt;
}

// Owner `me`: send[g] = shard[local row of the g-th requested node], requests taken requester by requester.
__global__ __launch_bounds__(256) void lookup_pack_kernel(const int32_t *__restrict__ acts, int R, OwnerMap m, int me,
                                                          const uint4 *__restrict__ shard, int row16,
                                                          uint4 *__restrict__ send, int32_t *__restrict__ send_off) {
    __shared__ Ranges s_rg[ELIMREC_MAX_RANKS];
    __shared__ int s_off[ELIMREC_MAX_RANKS + 1];
    const int W = m.world;
    if ((int)threadIdx.x < W) s_rg[threadIdx.x] = owner_ranges(acts + (int64_t)threadIdx.x * R, R, m, me);
    __syncthreads();
    if (threadIdx.x == 0) {
        int off = 0;
        for (int r = 0; r < W; ++r) { s_off[r] = off; off += s_rg[r].ucnt + s_rg[r].icnt; }
        s_off[W] = off;
    }
    __syncthreads();
    if (send_off && blockIdx.x == 0 && (int)threadIdx.x <= W) send_off[threadIdx.x] = s_off[threadIdx.x];
    const int total = s_off[W];
    const int lane = threadIdx.x & 63;
    const int64_t n_users_loc = m.ub[me + 1] - m.ub[me];
    for (int g = blockIdx.x * 4 + (threadIdx.x >> 6); g < total; g += gridDim.x * 4) {
        int r = 0;
        while (r + 1 < W && g >= s_off[r + 1]) ++r;
        const int k = g - s_off[r];
        const Ranges rg = s_rg[r];
        const bool user = k < rg.ucnt;
        const int j = user ? rg.ulo + k : rg.ilo + (k - rg.ucnt);
        const int64_t node = acts[(int64_t)r * R + j];
        const int64_t loc = user ? node - m.ub[me] : n_users_loc + (node - m.U - m.ib[me]);
        const uint4 *src = shard + loc * row16;
        uint4 *dst = send + (int64_t)g * row16;
        for (int c = lane; c < row16; c += 64) dst[c] = src[c];
    }
}

template <int DT> struct Elem;
template <> struct Elem<0> {       // fp32
    static __device__ __forceinline__ float4 load4(const void *row, int c4) { return reinterpret_cast<const float4 *>(row)[c4]; }
    static __device__ __forceinline__ float cval(const void *row, int c_col) { return reinterpret_cast<const float *>(row)[c_col]; }
};
template <> struct Elem<1> {       // fp16
    static __device__ __forceinline__ float4 load4(const void *row, int c4) {
        const uint2 v = reinterpret_cast<const uint2 *>(row)[c4];
        const __half2 a = *reinterpret_cast<const __half2 *>(&v.x), b = *reinterpret_cast<const __half2 *>(&v.y);
        const float2 fa = __half22float2(a), fb = __half22float2(b);
        return make_float4(fa.x, fa.y, fb.x, fb.y);
    }
    static __device__ __forceinline__ float cval(const void *row, int c_col) {
        const __half *h = reinterpret_cast<const __half *>(row) + c_col;
        return __half2float(h[0]) + __half2float(h[1]);
    }
};
template <> struct Elem<2> {       // bf16: the upper 16 bits of an fp32
    static __device__ __forceinline__ float4 load4(const void *row, int c4) {
        const uint2 v = reinterpret_cast<const uint2 *>(row)[c4];
        return make_float4(__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xffff0000u), __uint_as_float(v.y << 16),
                           __uint_as_float(v.y & 0xffff0000u));
    }
    static __device__ __forceinline__ float cval(const void *row, int c_col) {
        const uint16_t *h = reinterpret_cast<const uint16_t *>(row) + c_col;
        return __uint_as_float((uint32_t)h[0] << 16) + __uint_as_float((uint32_t)h[1] << 16);
    }
};

// Requester `me` (direct = 0): recv holds, owner by owner, the rows pack_kernel of that owner wrote for my list.
// direct = 1 (one rank, or rows this rank owns itself): rows are read from the local shard.
template <int DT>
__global__ __launch_bounds__(256) void lookup_unpack_kernel(const int32_t *__restrict__ act, int R, OwnerMap m, int me,
                                                            const char *__restrict__ rows_in, int64_t row_bytes, int sumD,
                                                            int direct, float *__restrict__ S_out, int64_t ldS,
                                                            float *__restrict__ c_out) {
    __shared__ Ranges s_rg[ELIMREC_MAX_RANKS];
    __shared__ int s_off[ELIMREC_MAX_RANKS + 1];
    __shared__ int s_count;
    const int W = m.world;
    if ((int)threadIdx.x < W) s_rg[threadIdx.x] = owner_ranges(act, R, m, threadIdx.x);
    if (threadIdx.x == 64) s_count = first_at_least(act, R, (int64_t)1 << 40);
    __syncthreads();
    if (threadIdx.x == 0) {
        int off = 0;
        for (int o = 0; o < W; ++o) { s_off[o] = off; off += s_rg[o].ucnt + s_rg[o].icnt; }
        s_off[W] = off;
    }
    __syncthreads();
    const int count = s_count;
    const int lane = threadIdx.x & 63;
    const int d4 = sumD / 4;
    for (int j = blockIdx.x * 4 + (threadIdx.x >> 6); j < count; j += gridDim.x * 4) {
        const int64_t node = act[j];
        const bool user = node < m.U;
        int o = 0;
        if (user) { while (o + 1 < W && node >= m.ub[o + 1]) ++o; }
        else { while (o + 1 < W && node - m.U >= m.ib[o + 1]) ++o; }
        int64_t pos;
        if (direct) pos = user ? node - m.ub[o] : (m.ub[o + 1] - m.ub[o]) + (node - m.U - m.ib[o]);
        else pos = s_off[o] + (user ? j - s_rg[o].ulo : s_rg[o].ucnt + (j - s_rg[o].ilo));
        const char *src = rows_in + pos * row_bytes;
        float4 *dst = reinterpret_cast<float4 *>(S_out + (int64_t)j * ldS);
        for (int c = lane; c < d4; c += 64) dst[c] = Elem<DT>::load4(src, c);
        if (lane == 0) c_out[j] = Elem<DT>::cval(src, sumD);
    }
}

// The column shards' forward exchange leaves, per peer q, the (layer mean | shared part) of MY active rows in q's columns:
// recv [W x R x 2 x dl]. The head wants rows: out0[r, q*dl + c] = recv[q, r, 0, c], out1[r, q*dl + c] = recv[q, r, 1, c].
__global__ __launch_bounds__(256) void peer_cols_to_rows_kernel(const float4 *__restrict__ recv, int W, int64_t R, int dl4,
                                                                float *__restrict__ out0, int64_t ld0, float *__restrict__ out1, int64_t ld1) {
    const int64_t per_row = (int64_t)W * 2 * dl4;
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= R * per_row) return;
    const int64_t r = t / per_row;
    const int k = (int)(t % per_row);
    const int q = k / (2 * dl4), h = (k / dl4) & 1, c4 = k % dl4;
    const float4 v = recv[(((int64_t)q * R + r) * 2 + h) * dl4 + c4];
    float *dst = (h ? out1 + r * ld1 : out0 + r * ld0) + ((int64_t)q * dl4 + c4) * 4;
    *reinterpret_cast<float4 *>(dst) = v;
}

static int fill_owner_map(OwnerMap &m, int world, int64_t U, int64_t I, const int64_t *ub, const int64_t *ib) {
    ELIMREC_REQUIRE(world >= 1 && world <= ELIMREC_MAX_RANKS, "lookup: 1..%d ranks", ELIMREC_MAX_RANKS);
    ELIMREC_REQUIRE(ub && ib, "lookup: owner bounds missing");
    ELIMREC_REQUIRE(ub[0] == 0 && ib[0] == 0 && ub[world] == U && ib[world] == I, "lookup: bounds must cover [0,U) and [0,I)");
    ELIMREC_REQUIRE(U + I < (int64_t)INT32_MAX, "lookup: node ids must fit int32");
    m.world = world; m.U = U;
    for (int o = 0; o <= world; ++o) {
        ELIMREC_REQUIRE(o == 0 || (ub[o] >= ub[o - 1] && ib[o] >= ib[o - 1]), "lookup: bounds must be ascending");
        m.ub[o] = ub[o]; m.ib[o] = ib[o];
    }
    return 0;
}

}  // namespace elimrec

using namespace elimrec;

extern "C" int elimrec_lookup_counts(const int32_t *d_acts, int world, int64_t R, int64_t U, int64_t I, const int64_t *ub,
                                     const int64_t *ib, int32_t *d_counts, void *stream) {
    ELIMREC_REQUIRE(d_acts && d_counts && R >= 0 && R < INT32_MAX, "lookup_counts: bad arguments");
    OwnerMap m;
    int rc = fill_owner_map(m, world, U, I, ub, ib);
    if (rc) return rc;
    hipLaunchKernelGGL(lookup_counts_kernel, dim3((world * world + 255) / 256), dim3(256), 0, (hipStream_t)stream, d_acts, (int)R, m,
                       d_counts);
    ELIMREC_LAUNCH_CHECK("lookup_counts");
    return 0;
}

extern "C" int elimrec_lookup_pack(const int32_t *d_acts, int world, int64_t R, int64_t U, int64_t I, const int64_t *ub,
                                   const int64_t *ib, int me, const void *d_shard, int64_t row_bytes, void *d_send,
                                   int32_t *d_send_off, void *stream) {
    ELIMREC_REQUIRE(d_acts && d_shard && d_send && R >= 0 && R < INT32_MAX, "lookup_pack: bad arguments");
    ELIMREC_REQUIRE(row_bytes > 0 && row_bytes % 16 == 0, "lookup_pack: rows must be a multiple of 16 bytes");
    ELIMREC_REQUIRE(me >= 0 && me < world, "lookup_pack: rank out of range");
    OwnerMap m;
    int rc = fill_owner_map(m, world, U, I, ub, ib);
    if (rc) return rc;
    if (R == 0) return 0;
    int64_t blocks = ((int64_t)world * R + 3) / 4;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(lookup_pack_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, d_acts, (int)R, m, me,
                       (const uint4 *)d_shard, (int)(row_bytes / 16), (uint4 *)d_send, d_send_off);
    ELIMREC_LAUNCH_CHECK("lookup_pack");
    return 0;
}

extern "C" int elimrec_lookup_unpack(const int32_t *d_act, int world, int64_t R, int64_t U, int64_t I, const int64_t *ub,
                                     const int64_t *ib, int me, const void *d_rows, int64_t row_bytes, int dtype, int sum_d,
                                     int direct, float *d_S, int64_t ldS, float *d_c, void *stream) {
    ELIMREC_REQUIRE(d_act && d_rows && d_S && d_c && R >= 0 && R < INT32_MAX, "lookup_unpack: bad arguments");
    ELIMREC_REQUIRE(dtype >= 0 && dtype <= 2, "lookup_unpack: dtype 0 (f32), 1 (f16) or 2 (bf16)");
    const int es = dtype == 0 ? 4 : 2;
    ELIMREC_REQUIRE(sum_d > 0 && sum_d % 4 == 0 && ldS >= sum_d && ldS % 4 == 0, "lookup_unpack: widths must be multiples of 4");
    ELIMREC_REQUIRE(row_bytes % 16 == 0 && row_bytes >= (int64_t)(sum_d + (dtype == 0 ? 1 : 2)) * es,
                    "lookup_unpack: row_bytes too small for sum_d + c");
    ELIMREC_REQUIRE(me >= 0 && me < world, "lookup_unpack: rank out of range");
    OwnerMap m;
    int rc = fill_owner_map(m, world, U, I, ub, ib);
    if (rc) return rc;
    if (R == 0) return 0;
    int64_t blocks = (R + 3) / 4;
    if (blocks > 2048) blocks = 2048;
    const dim3 grid((unsigned)blocks), blk(256);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == 0)
        hipLaunchKernelGGL(lookup_unpack_kernel<0>, grid, blk, 0, s, d_act, (int)R, m, me, (const char *)d_rows, row_bytes, sum_d, direct,
                           d_S, ldS, d_c);
    else if (dtype == 1)
        hipLaunchKernelGGL(lookup_unpack_kernel<1>, grid, blk, 0, s, d_act, (int)R, m, me, (const char *)d_rows, row_bytes, sum_d, direct,
                           d_S, ldS, d_c);
    else
        hipLaunchKernelGGL(lookup_unpack_kernel<2>, grid, blk, 0, s, d_act, (int)R, m, me, (const char *)d_rows, row_bytes, sum_d, direct,
                           d_S, ldS, d_c);
    ELIMREC_LAUNCH_CHECK("lookup_unpack");
    return 0;
}

extern "C" int elimrec_peer_cols_to_rows(const float *d_recv, int world, int64_t R, int dl, float *d_out0, int64_t ld0,
                                         float *d_out1, int64_t ld1, void *stream) {
    ELIMREC_REQUIRE(d_recv && d_out0 && d_out1 && world >= 1 && dl > 0 && dl % 4 == 0 && ld0 % 4 == 0 && ld1 % 4 == 0,
                    "peer_cols_to_rows: bad arguments");
    if (R <= 0) return 0;
    const int64_t total = R * world * 2 * (dl / 4);
    hipLaunchKernelGGL(peer_cols_to_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const float4 *)d_recv, world, R, dl / 4, d_out0, ld0, d_out1, ld1);
    ELIMREC_LAUNCH_CHECK("peer_cols_to_rows");
    return 0;
}
