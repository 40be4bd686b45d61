// Weight-gradient contraction (gemm.hip): problem table, decomposition and the fixed-order slab reduce, shared with the
// hop launches that can carry either phase as extra workgroups (slab.hip, elimrec_slab_hop_bwd_w).
#pragma once
#include "common.h"
#include <cstdlib>

namespace elimrec {

constexpr int kMaxBatch = 8;
typedef float v16f __attribute__((ext_vector_type(16)));
#ifndef ELIMREC_BWDW_TN2
#define ELIMREC_BWDW_TN2 64
#endif
constexpr int TN1 = 64, TN2 = ELIMREC_BWDW_TN2, TRB = 32;      // TN2 = 64 (one MFMA tile per wave; measured 3 us faster
                                                                  // per step than 128 = two tiles per wave: more, smaller workgroups)
constexpr int BNJ = TN2 / 64, BC4 = TN2 / 4, BRP = 256 / BC4, BPS = TRB / BRP;   // tiles per wave; B loader geometry

struct BwdProblem {
    elimrec_linear_bwd_desc d;
    int chunk_rows, chunks, t1, t2;       // decomposition
    int first_block;                      // prefix of (chunks * t1 * t2) over the problems before this one
    float *slabs, *cslabs;
};
struct BwdBatch { BwdProblem p[kMaxBatch]; int n; };

// One workgroup of the partial launch: a chunk of rows x one 64 x TN2 output tile (see gemm.hip, "weight grad").
// As / Bs: the caller's LDS stages ([2][TRB * TN1], [2][TRB * TN2] floats = 32 KB: five workgroups per CU). The per-row
// weights of the weighted column sum stay in registers: lane k of wave 0 holds row k's, the sum reads it with v_readlane.
__device__ __forceinline__ void bwd_w_partial_body(const BwdBatch &batch, int block, float (*As)[TRB * TN1], float (*Bs)[TRB * TN2]) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int pi = 0;
    while (pi + 1 < batch.n && block >= batch.p[pi + 1].first_block) ++pi;
    const BwdProblem &pb = batch.p[pi];
    const int local = block - pb.first_block;
    const int n1_tiles = pb.t1, n2_tiles = pb.t2;
    const int chunk = local / (n1_tiles * n2_tiles);
    const int tile = local - chunk * (n1_tiles * n2_tiles);
    const int tile_y = tile / n2_tiles, tile_z = tile - tile_y * n2_tiles;
    const float *__restrict__ A = pb.d.d_A;
    const float *__restrict__ B = pb.d.d_B;
    const int32_t *__restrict__ row_index = pb.d.d_row_index;
    const int32_t *__restrict__ range = pb.d.d_range;
    const int64_t lda = pb.d.lda, ldb = pb.d.ldb, R = pb.d.R;
    const int n1 = pb.d.n1, n2 = pb.d.n2, chunk_rows = pb.chunk_rows;
    float *__restrict__ slabs = pb.slabs;
    float *__restrict__ colsum_slabs = pb.d.d_colsum ? pb.cslabs : nullptr;
    const float *__restrict__ cw = pb.d.d_colsum ? pb.d.d_colsum_weight : nullptr;
    int64_t rb = 0, re = R;
    if (range) { rb = range[0]; re = range[1]; }
    const int64_t r0 = rb + (int64_t)chunk * chunk_rows;
    const int64_t r1 = (r0 + chunk_rows < re) ? r0 + chunk_rows : re;
    const int i_base = tile_y * TN1, j_base = tile_z * TN2;

    // loader geometry: A block = 32 rows x 16 float4 (2 per thread); B block = 32 rows x 32 float4 (4 per thread)
    const int a_row = tid >> 4, a_c4 = tid & 15;          // + 16 rows on the second pass
    const int b_row = tid / BC4, b_c4 = tid % BC4;        // + BRP rows per pass, BPS passes
    const bool a_col_ok = (i_base + a_c4 * 4) < n1;       // n1, n2 are multiples of 4
    const bool b_col_ok = (j_base + b_c4 * 4) < n2;
    float4 ra[2], rbv[BPS];
    float rw = 1.f, rw_cur = 1.f;                          // the block in flight / the block in LDS
    auto load_block = [&](int64_t row0) {
        if (cw && tid < TRB) {
            const int64_t r = row0 + tid;
            rw = r < r1 ? cw[row_index ? (int64_t)row_index[r] : r] : 0.f;
        }
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int64_t r = row0 + a_row + 16 * p;
            ra[p] = (a_col_ok && r < r1) ? *reinterpret_cast<const float4 *>(A + r * lda + i_base + a_c4 * 4)
                                         : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int p = 0; p < BPS; ++p) {
            const int64_t r = row0 + b_row + BRP * p;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (b_col_ok && r < r1) {
                const int64_t br = row_index ? (int64_t)row_index[r] : r;
                v = *reinterpret_cast<const float4 *>(B + br * ldb + j_base + b_c4 * 4);
            }
            rbv[p] = v;
        }
    };
    auto store_block = [&](int buf) {
        rw_cur = rw;
#pragma unroll
        for (int p = 0; p < 2; ++p)
            *reinterpret_cast<float4 *>(&As[buf][(a_row + 16 * p) * TN1 + a_c4 * 4]) = ra[p];
#pragma unroll
        for (int p = 0; p < BPS; ++p)
            *reinterpret_cast<float4 *>(&Bs[buf][(b_row + BRP * p) * TN2 + b_c4 * 4]) = rbv[p];
    };

    const int wi = (wave & 1) * 32, wj = (wave >> 1) * (TN2 / 2);
    const int li = lane & 31, lk = lane >> 5;
    // 64-column problems (the single-modal heads) fill half of the 128-column tile: the waves of the empty half do not
    // multiply zeros or write them (the reduce never reads the padding columns)
    const bool wave_on = (j_base + wj) < n2;
    const bool half1_on = BNJ > 1 && (j_base + wj + 32) < n2;
    v16f acc0 = {0}, acc1 = {0};
    float csum = 0.f;                                      // threads 0..63: column sum of A[:, i_base + tid]
    int buf = 0;
    if (r0 < r1) {
        load_block(r0);
        store_block(0);
    }
    __syncthreads();
    for (int64_t row0 = r0; row0 < r1; row0 += TRB) {
        const bool more = (row0 + TRB) < r1;
        if (more) load_block(row0 + TRB);                  // in flight during the MFMAs below
        const float *as = As[buf], *bs = Bs[buf];
        if (wave_on) {
#pragma unroll
            for (int kk = 0; kk < TRB; kk += 2) {
                const float a = as[(kk + lk) * TN1 + wi + li];
                const float b0 = bs[(kk + lk) * TN2 + wj + li];
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0, acc0, 0, 0, 0);
                if (BNJ > 1) {
                    const float b1 = bs[(kk + lk) * TN2 + wj + (BNJ > 1 ? 32 : 0) + li];
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b1, acc1, 0, 0, 0);
                }
            }
        }
        if (colsum_slabs && tid < TN1) {
            if (cw) {
#pragma unroll
                for (int k = 0; k < TRB; ++k)
                    csum += as[k * TN1 + tid] * __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, rw_cur), k));
            } else {
#pragma unroll 8
                for (int k = 0; k < TRB; ++k) csum += as[k * TN1 + tid];
            }
        }
        if (more) store_block(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }
    const int n1_pad = n1_tiles * TN1, n2_pad = n2_tiles * TN2;
    float *slab = slabs + (size_t)chunk * n1_pad * n2_pad;
    if (wave_on) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = i_base + wi + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            slab[(size_t)row * n2_pad + j_base + wj + li] = acc0[r];
            if (half1_on) slab[(size_t)row * n2_pad + j_base + wj + 32 + li] = acc1[r];
        }
    }
    if (colsum_slabs && tile_z == 0 && tid < TN1) colsum_slabs[(size_t)chunk * n1_pad + i_base + tid] = csum;
}

// out[e] (+)= sum over chunks of slab[chunk][e] in a FIXED order: four adjacent lanes share one output
// element, lane q adds chunks q, q+4, q+8, ... (4 loads in flight each), then (q0+q1)+(q2+q3).
// by selects the problem; 256 threads per workgroup.
__device__ __forceinline__ void reduce_slabs_body(const BwdBatch &batch, int bx, int by) {
    const BwdProblem &pb = batch.p[by];
    const int n1 = pb.d.n1, n2 = pb.d.n2;
    const int n1_pad = pb.t1 * TN1, n2_pad = pb.t2 * TN2;
    const int gid = bx * 256 + (int)threadIdx.x;
    const int idx = gid >> 2, q = gid & 3;
    const int total = n1 * n2;
    int64_t rows = pb.d.d_range ? (int64_t)pb.d.d_range[1] - pb.d.d_range[0] : pb.d.R;
    if (rows < 0) rows = 0;
    const int chunks = (int)((rows + pb.chunk_rows - 1) / pb.chunk_rows);
    const float *src = nullptr;
    size_t stride = 0;
    float *dst = nullptr;
    if (idx < total) {
        const int i = idx / n2, j = idx - i * n2;
        src = pb.slabs + (size_t)i * n2_pad + j;
        stride = (size_t)n1_pad * n2_pad;
        dst = pb.d.d_out + (int64_t)i * pb.d.ldo + j;
    } else if (pb.d.d_colsum && idx < total + n1) {
        src = pb.cslabs + (idx - total);
        stride = (size_t)n1_pad;
        dst = pb.d.d_colsum + (idx - total);
    }
    float s = 0.f;
    if (src) {
        int c = q;
        for (; c + 12 < chunks; c += 16) {
            const float v0 = src[(size_t)c * stride], v1 = src[(size_t)(c + 4) * stride];
            const float v2 = src[(size_t)(c + 8) * stride], v3 = src[(size_t)(c + 12) * stride];
            s += v0; s += v1; s += v2; s += v3;
        }
        for (; c < chunks; c += 4) s += src[(size_t)c * stride];
    }
    s += __shfl_xor(s, 1, 64);       // (q0+q1), (q2+q3)
    s += __shfl_xor(s, 2, 64);       // sum of the two pairs
    if (src && q == 0) *dst = pb.d.accumulate ? (*dst + s) : s;
}


// ---- host side
// Row-chunk size: the partial kernel holds 48 KB of LDS, i.e. 3 workgroups per CU = 768 resident at once;
// chunks are sized so that one problem's workgroups fill about a third of that (batches hold ~3 problems
// of equal weight) in ONE round, between 64 and 512 rows.
static inline void bwd_w_dims(int64_t R, int n1, int n2, int &chunk_rows, int &chunks, int &t1, int &t2) {
    const int tiles = ((n1 + TN1 - 1) / TN1) * ((n2 + TN2 - 1) / TN2);
    static int target = 0;
    if (!target) target = 480;
    int64_t want = (R * tiles + target - 1) / target;       // rows per workgroup for ~`target` workgroups per problem
    want = (want + TRB - 1) / TRB * TRB;
    chunk_rows = (int)(want < 64 ? 64 : (want > 512 ? 512 : want));
    chunks = (int)((R + chunk_rows - 1) / chunk_rows);
    if (chunks < 1) chunks = 1;
    t1 = (n1 + TN1 - 1) / TN1;
    t2 = (n2 + TN2 - 1) / TN2;
}

static inline size_t bwd_w_bytes(int64_t R, int n1, int n2) {
    int cr, chunks, t1, t2;
    bwd_w_dims(R, n1, n2, cr, chunks, t1, t2);
    return align_up(((size_t)chunks * t1 * TN1 * t2 * TN2 + (size_t)chunks * t1 * TN1) * sizeof(float), 256);
}


// fills the problem table of `n` contractions whose slabs live in d_workspace; blocks = workgroups of the partial launch,
// max_out = the largest problem's output elements (the reduce runs a (4 * max_out / 256) x n grid)
static inline int bwd_w_build_batch(const elimrec_linear_bwd_desc *descs, int n, void *d_workspace, BwdBatch &batch, int &blocks,
                                    int &max_out) {
    batch.n = n;
    char *ws = (char *)d_workspace;
    blocks = 0; max_out = 0;
    for (int i = 0; i < n; ++i) {
        const elimrec_linear_bwd_desc &d = descs[i];
        ELIMREC_REQUIRE(d.d_A && d.d_B && d.d_out, "linear_bwd_w: null pointer");
        ELIMREC_REQUIRE(d.R >= 0 && d.n1 > 0 && d.n2 > 0, "linear_bwd_w: bad shape");
        ELIMREC_REQUIRE(d.n1 % 4 == 0 && d.n2 % 4 == 0 && d.lda % 4 == 0 && d.ldb % 4 == 0,
                        "linear_bwd_w: n1, n2, lda, ldb must be multiples of 4");
        ELIMREC_REQUIRE(((uintptr_t)d.d_A % 16) == 0 && ((uintptr_t)d.d_B % 16) == 0,
                        "linear_bwd_w: A and B must be 16-byte aligned");
        BwdProblem &pb = batch.p[i];
        pb.d = d;
        bwd_w_dims(d.R, d.n1, d.n2, pb.chunk_rows, pb.chunks, pb.t1, pb.t2);
        pb.first_block = blocks;
        blocks += pb.chunks * pb.t1 * pb.t2;
        pb.slabs = (float *)ws;
        pb.cslabs = pb.slabs + (size_t)pb.chunks * pb.t1 * TN1 * pb.t2 * TN2;
        ws += bwd_w_bytes(d.R, d.n1, d.n2);
        const int out_elems = d.n1 * d.n2 + (d.d_colsum ? d.n1 : 0);
        if (out_elems > max_out) max_out = out_elems;
    }
    return 0;
}

static inline size_t bwd_w_batched_bytes(const elimrec_linear_bwd_desc *descs, int n) {
    size_t total = 0;
    for (int i = 0; i < n; ++i) total += bwd_w_bytes(descs[i].R, descs[i].n1, descs[i].n2);
    return total;
}

}  // namespace elimrec
