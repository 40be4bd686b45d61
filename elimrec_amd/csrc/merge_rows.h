// Merge of the gathered adjoint-source rows into the slab-major source tables (elimrec_slab_merge_rows), as a device
// function: the launch of its own (slab.hip) and the leading / trailing workgroups of the weight-gradient launch
// (gemm.hip, elimrec_linear_bwd_w_batched_merge) run the same body.
#pragma once
#include "common.h"

namespace elimrec {

constexpr int kSlabMaxRanks = 64;
constexpr int kMergeSeenWords = 4096;      // LDS words a fused launch can lend to the chunk's row bits (16 KB: one GEMM stage pair)

struct MergeArgs {
    const float *rows;
    const int32_t *keys;
    int W, R;
    int64_t U, N;
    int nc4, w4, w4_shift, chunk, M;
    float *SrcA, *SrcB;
    uint32_t *mask;
    int plain;                             // 1: H always to SrcA, G always to SrcB (the wide form's [H | G] source, wide.hip); 0: by side
};

// rows per workgroup: about 1024 workgroups over the N rows, whole bitmap words
static inline int merge_rows_chunk(int64_t N) {
    int chunk = (int)((N + 1023) / 1024);
    chunk = (chunk + 31) / 32 * 32;
    return chunk < 32 ? 32 : chunk;
}

__device__ __forceinline__ int slab_merge_key(const int32_t *keys, int i) {
    const int k = keys[i];
    return k < 0 ? INT32_MAX : k;
}

// s_beg / s_end: kSlabMaxRanks ints each; seen: chunk / 32 words (LDS)
__device__ __forceinline__ void slab_merge_rows_body(const MergeArgs &a, int block, int *s_beg, int *s_end, uint32_t *seen) {
    const float *__restrict__ rows = a.rows;
    const int32_t *__restrict__ keys = a.keys;
    const int W = a.W, R = a.R, nc4 = a.nc4, w4 = a.w4, w4_shift = a.w4_shift, chunk = a.chunk, M = a.M;
    const int64_t U = a.U, N = a.N;
    float *SrcA = a.SrcA, *SrcB = a.SrcB;
    uint32_t *__restrict__ mask = a.mask;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t lo = (int64_t)block * chunk, hi = min(lo + chunk, N);
    for (int w = tid; w < chunk / 32; w += 256) seen[w] = 0u;
    if (tid < W) {
        const int32_t *kr = keys + (int64_t)tid * R;
        int a0 = 0, a1 = R, b0 = 0, b1 = R;
        while (a0 < a1 || b0 < b1) {
            if (a0 < a1) { const int m = (a0 + a1) >> 1; if (slab_merge_key(kr, m) < lo) a0 = m + 1; else a1 = m; }
            if (b0 < b1) { const int m = (b0 + b1) >> 1; if (slab_merge_key(kr, m) < hi) b0 = m + 1; else b1 = m; }
        }
        s_beg[tid] = a0; s_end[tid] = b0;
    }
    __syncthreads();
    // lanes per row = the power of two >= the row's float4 count: a wave takes 64 / lg rows of a rank at a time (a column
    // shard's rows are 2-4 float4 wide; one row per wave would leave 60 lanes idle and cost a round trip per row)
    int lg = 1;
    while (lg < nc4 && lg < 64) lg <<= 1;
    const int rpw = 64 / lg, sub = lane / lg, cl = lane % lg;
    for (int r = 0; r < W; ++r) {
        for (int s = s_beg[r] + wave * rpw + sub; s < s_end[r]; s += 4 * rpw) {
            const int64_t node = keys[(int64_t)r * R + s];
            const int bit = (int)(node - lo);
            const bool was = (seen[bit >> 5] >> (bit & 31)) & 1u;
            const float4 *g = reinterpret_cast<const float4 *>(rows) + ((int64_t)r * R + s) * (M ? M : 2) * nc4;
            float *hT = (a.plain || node < U) ? SrcA : SrcB;       // H lives in SrcA on user rows, SrcB on item rows
            float *gT = (a.plain || node < U) ? SrcB : SrcA;
            for (int c = cl; c < nc4; c += lg) {
                const int64_t idx = (((int64_t)(c >> w4_shift) * N + node) * w4 + (c & (w4 - 1))) * 4;
                float4 h, g0;
                if (M) {                                 // dOut rows: G = block 0, H = sum of the M blocks (block order)
                    g0 = g[c];
                    h = g0;
                    for (int mb = 1; mb < M; ++mb) {
                        const float4 x = g[mb * nc4 + c];
                        h.x += x.x; h.y += x.y; h.z += x.z; h.w += x.w;
                    }
                } else { h = g[c]; g0 = g[nc4 + c]; }
                if (was) {                               // written by an earlier rank of this workgroup: read through L2
                    float *hp = hT + idx, *gp = gT + idx;
                    float4 x, y;
                    x.x = __hip_atomic_load(hp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    x.y = __hip_atomic_load(hp + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    x.z = __hip_atomic_load(hp + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    x.w = __hip_atomic_load(hp + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    y.x = __hip_atomic_load(gp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    y.y = __hip_atomic_load(gp + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    y.z = __hip_atomic_load(gp + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    y.w = __hip_atomic_load(gp + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    h = make_float4(x.x + h.x, x.y + h.y, x.z + h.z, x.w + h.w);
                    g0 = make_float4(y.x + g0.x, y.y + g0.y, y.z + g0.z, y.w + g0.w);
                }
                *reinterpret_cast<float4 *>(hT + idx) = h;
                *reinterpret_cast<float4 *>(gT + idx) = g0;
            }
            if (cl == 0 && !was) atomicOr(&seen[bit >> 5], 1u << (bit & 31));
        }
        __syncthreads();
    }
    for (int w = tid; w < chunk / 32; w += 256)
        if (lo + 32 * (int64_t)w < ((N + 31) / 32) * 32) mask[lo / 32 + w] = seen[w];
}

}  // namespace elimrec
