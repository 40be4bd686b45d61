// The hop's wave-tile plan built on the device (models/EliMRec.py:309-354 hands the adjacency over once; slab.SellPlan restates
// it as the index the hop kernels walk). The host form (slab.SellPlan._wave_tiles, numpy argsort / lexsort / cumsum over all
// non-zeros) is the definition; this file reproduces its arrays bit for bit from a device CSR, for graphs whose plan a host
// cannot build in reasonable time or memory (BASELINE.json configs[3]: 18 M non-zeros, configs[4]: 2 x 10^9):
//   rows   -> a class per row (a workgroup / a wave / segments / one of G rows of a tile) and ONE stable radix sort by
//             (class, side, descending length): the four row orders of the plan are the four stretches of its result;
//             the long rows' ranks and the segment slots by two scans in row order;
//   tiles  -> the segments sorted by descending length (stable), then a record per (tile, lane group): first CSR position,
//             count, stride, destination; a tile's steps = its longest group, the tiles' offsets an exclusive scan;
//   scatter-> every non-zero to tile_off[tile] + step * G + group.
// rocPRIM does the sorts and scans; everything else is one thread per row / segment / (tile, group).
#include <cstring>
#include "common.h"
#include <rocprim/rocprim.hpp>

namespace elimrec {

struct PlanShape {
    int64_t n_rows;
    int T, T1, T2, TS, G;
    int64_t side_split;      // rows below are one side (users): inside the tiles of short rows the OTHER side comes first; < 0: none
    int64_t rows_from;       // rows below take no part (they are another plan's)
};

enum { CLS_W4 = 0, CLS_W1 = 1, CLS_SPLIT = 2, CLS_SHORT = 3, CLS_NONE = 4 };

__device__ __forceinline__ int row_class(const PlanShape &p, int64_t r, int64_t deg) {
    if (r < p.rows_from) return CLS_NONE;
    if (deg <= p.T) return CLS_SHORT;
    if (deg <= p.T1) return CLS_W1;
    if (deg <= p.T2) return CLS_W4;
    return CLS_SPLIT;
}

// key = class | side | (2^32 - 1 - length): ascending keys = classes in plan order, inside the short rows the side at or above
// side_split first, longer rows first, equal rows by row number (the sort is stable). Split rows keep their row order.
__global__ void plan_keys_kernel(PlanShape p, const int64_t *__restrict__ rowptr, uint64_t *__restrict__ keys, int32_t *__restrict__ rows,
                                 int32_t *__restrict__ is_long, int32_t *__restrict__ nseg) {
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < p.n_rows; r += (int64_t)gridDim.x * blockDim.x) {
        const int64_t deg = rowptr[r + 1] - rowptr[r];
        const int cls = row_class(p, r, deg);
        const uint64_t side = (cls == CLS_SHORT && p.side_split >= 0 && r < p.side_split) ? 1u : 0u;
        const uint64_t sec = (cls == CLS_SPLIT || cls == CLS_NONE) ? 0u : (0xFFFFFFFFull - (uint64_t)deg);
        keys[r] = ((uint64_t)cls << 34) | (side << 33) | sec;
        rows[r] = (int32_t)r;
        const bool lng = cls != CLS_NONE && deg > p.T;
        is_long[r] = lng ? 1 : 0;
        nseg[r] = cls == CLS_SPLIT ? (int32_t)((deg + p.TS - 1) / p.TS) : 0;
    }
}

// counts[0..4] = rows of class 0..4 (from the sorted keys), counts[5] = long rows, counts[6] = segments
__global__ void plan_counts_kernel(PlanShape p, const uint64_t *__restrict__ sorted_keys, const int32_t *__restrict__ long_scan,
                                   const int32_t *__restrict__ is_long, const int32_t *__restrict__ seg_scan, const int32_t *__restrict__ nseg,
                                   int64_t *__restrict__ counts) {
    if (blockIdx.x != 0 || threadIdx.x > 5) return;
    const int c = threadIdx.x;
    if (c < 5) {
        // first key of class c and of class c + 1
        auto lower = [&](uint64_t want) { int64_t lo = 0, hi = p.n_rows; while (lo < hi) { const int64_t m = (lo + hi) >> 1; if (sorted_keys[m] < want) lo = m + 1; else hi = m; } return lo; };
        counts[c] = lower((uint64_t)(c + 1) << 34) - lower((uint64_t)c << 34);
    } else {
        counts[5] = p.n_rows ? (int64_t)long_scan[p.n_rows - 1] + is_long[p.n_rows - 1] : 0;
        counts[6] = p.n_rows ? (int64_t)seg_scan[p.n_rows - 1] + nseg[p.n_rows - 1] : 0;
    }
}

__global__ void plan_long_kernel(PlanShape p, const int32_t *__restrict__ is_long, const int32_t *__restrict__ long_scan,
                                 const int32_t *__restrict__ seg_scan, int32_t *__restrict__ long_rows, int32_t *__restrict__ long_index,
                                 int32_t *__restrict__ long_seg_ptr, const int64_t *__restrict__ counts) {
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < p.n_rows; r += (int64_t)gridDim.x * blockDim.x) {
        if (is_long[r]) {
            const int32_t k = long_scan[r];
            long_rows[k] = (int32_t)r;
            long_index[r] = k;
            long_seg_ptr[k] = seg_scan[r];
        } else long_index[r] = -1;
        if (r == 0) long_seg_ptr[counts[5]] = (int32_t)counts[6];      // (counts: written by the launch before this one)
    }
}

// segment k of the split rows, in (long row, piece) order: its first CSR position, its length, its long row's rank
__global__ void plan_segments_kernel(PlanShape p, const int64_t *__restrict__ rowptr, const int32_t *__restrict__ long_rows,
                                     const int32_t *__restrict__ long_seg_ptr, int64_t n_long, int64_t n_seg, int64_t *__restrict__ seg_beg,
                                     int32_t *__restrict__ seg_len, int32_t *__restrict__ seg_rank, uint32_t *__restrict__ seg_key,
                                     int32_t *__restrict__ seg_id) {
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n_seg; k += (int64_t)gridDim.x * blockDim.x) {
        int64_t lo = 0, hi = n_long;                      // the last long row whose first slot is <= k
        while (lo < hi) { const int64_t m = (lo + hi) >> 1; if ((int64_t)long_seg_ptr[m + 1] <= k) lo = m + 1; else hi = m; }
        const int64_t row = long_rows[lo];
        const int64_t beg = rowptr[row] + (k - long_seg_ptr[lo]) * p.TS;
        const int64_t len = rowptr[row + 1] - beg < p.TS ? rowptr[row + 1] - beg : p.TS;
        seg_beg[k] = beg; seg_len[k] = (int32_t)len; seg_rank[k] = (int32_t)lo;
        seg_key[k] = (uint32_t)(p.TS - len);              // longest first, equal ones by slot (stable sort)
        seg_id[k] = (int32_t)k;
    }
}

struct TileCounts { int64_t n_w4, n_w1, n_split, n_short, n_seg, n_t4, n_t1, n_tseg, n_tfin; };

__host__ __device__ inline int64_t pad_to(int64_t k, int64_t m) { return (k + m - 1) / m * m; }

// One thread per (tile, lane group): the records of slab.SellPlan._wave_tiles -- A: a quarter of a workgroup row per tile, its
// neighbours dealt round-robin to the groups; B: a wave row per tile, the same; C: G segments per tile; D: G short rows per tile.
// gb: first CSR position, gl: count, gs: stride, gd: destination (row, or segment slot; -1: padding).
__global__ void plan_tiles_kernel(PlanShape p, TileCounts c, const int64_t *__restrict__ rowptr, const int32_t *__restrict__ order,
                                  const int64_t *__restrict__ seg_beg, const int32_t *__restrict__ seg_len, const int32_t *__restrict__ seg_rank,
                                  const int32_t *__restrict__ seg_sorted, int64_t *__restrict__ gb, int32_t *__restrict__ gl,
                                  int32_t *__restrict__ gs, int32_t *__restrict__ gd, int32_t *__restrict__ tile_long) {
    const int64_t G = p.G, nt = c.n_t4 + c.n_t1 + c.n_tseg + c.n_tfin;
    for (int64_t x = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; x < nt * G; x += (int64_t)gridDim.x * blockDim.x) {
        const int64_t tile = x / G, g = x % G;
        int64_t B = 0, L = 0, S = 1, D = -1;
        if (tile < c.n_t4) {                                           // A
            const int64_t row = order[tile / 4], q4 = tile % 4;
            const int64_t n4 = rowptr[row + 1] - rowptr[row];
            const int64_t q = pad_to((n4 + 3) / 4, G);                  // quarter length, a multiple of G
            int64_t cn = n4 - q4 * q;
            cn = cn < 0 ? 0 : (cn > q ? q : cn);
            B = rowptr[row] + q4 * q + g; L = (cn - g + G - 1) / G; S = G; D = row;
        } else if (tile < c.n_t4 + c.n_t1) {                           // B (+ empty tiles up to a multiple of 4)
            const int64_t k = tile - c.n_t4;
            if (k < c.n_w1) {
                const int64_t row = order[c.n_w4 + k];
                const int64_t n = rowptr[row + 1] - rowptr[row];
                B = rowptr[row] + g; L = (n - g + G - 1) / G; S = G; D = row;
            }
        } else if (tile < c.n_t4 + c.n_t1 + c.n_tseg) {                // C
            const int64_t k = (tile - c.n_t4 - c.n_t1) * G + g;
            if (k < c.n_seg) {
                const int32_t sg = seg_sorted[k];
                B = seg_beg[sg]; L = seg_len[sg]; D = sg;
                tile_long[k] = seg_rank[sg];
            } else tile_long[k] = 0;
        } else {                                                       // D
            const int64_t k = (tile - c.n_t4 - c.n_t1 - c.n_tseg) * G + g;
            if (k < c.n_short) {
                const int64_t row = order[c.n_w4 + c.n_w1 + c.n_split + k];
                B = rowptr[row]; L = rowptr[row + 1] - rowptr[row]; D = row;
            }
        }
        gb[x] = B; gl[x] = (int32_t)L; gs[x] = (int32_t)S; gd[x] = (int32_t)D;
    }
}

// steps of a tile = its longest group; entries = steps * G (the exclusive scan of these is tile_off)
__global__ void plan_steps_kernel(int64_t n_tiles, int G, const int32_t *__restrict__ gl, int64_t *__restrict__ entries) {
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n_tiles; t += (int64_t)gridDim.x * blockDim.x) {
        int32_t m = 0;
        for (int g = 0; g < G; ++g) m = gl[t * G + g] > m ? gl[t * G + g] : m;
        entries[t] = (int64_t)m * G;
    }
}

// totals[0] = all entries, totals[1] = entries of the tiles before the short rows' (A, B, C), totals[2] = most 64-entry lines of a tile
__global__ void plan_totals_kernel(int64_t n_tiles, int64_t seg_end_tile, const int64_t *__restrict__ tile_off, const int64_t *__restrict__ entries,
                                   int64_t *__restrict__ tile_off_last, int64_t *__restrict__ totals) {
    __shared__ int64_t best[256];
    int64_t m = 0;
    for (int64_t t = threadIdx.x; t < n_tiles; t += 256) { const int64_t l = (entries[t] + 63) / 64; m = l > m ? l : m; }
    best[threadIdx.x] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < 256; ++i) m = best[i] > m ? best[i] : m;
        const int64_t total = n_tiles ? tile_off[n_tiles - 1] + entries[n_tiles - 1] : 0;
        *tile_off_last = total;
        totals[0] = total;
        totals[1] = seg_end_tile < n_tiles ? tile_off[seg_end_tile] : total;
        totals[2] = n_tiles ? m : 1;
    }
}

__global__ void plan_scatter_kernel(int64_t n_groups, int G, const int64_t *__restrict__ tile_off, const int64_t *__restrict__ gb,
                                    const int32_t *__restrict__ gl, const int32_t *__restrict__ gs, const int32_t *__restrict__ col,
                                    const float *__restrict__ val, int32_t *__restrict__ tcol, float *__restrict__ tval) {
    for (int64_t x = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; x < n_groups; x += (int64_t)gridDim.x * blockDim.x) {
        const int64_t base = tile_off[x / G] + x % G, b = gb[x], st = gs[x];
        const int32_t n = gl[x];
        for (int32_t j = 0; j < n; ++j) {
            tcol[base + (int64_t)j * G] = col[b + (int64_t)j * st];
            tval[base + (int64_t)j * G] = val[b + (int64_t)j * st];
        }
    }
}

}  // namespace elimrec

using namespace elimrec;

static size_t plan_scan_bytes(size_t n) {
    size_t a = 0, b = 0;
    (void)rocprim::exclusive_scan(nullptr, a, (const int32_t *)nullptr, (int32_t *)nullptr, (int32_t)0, n, rocprim::plus<int32_t>(), 0, false);
    (void)rocprim::exclusive_scan(nullptr, b, (const int64_t *)nullptr, (int64_t *)nullptr, (int64_t)0, n, rocprim::plus<int64_t>(), 0, false);
    return a > b ? a : b;
}

extern "C" size_t elimrec_plan_workspace(int64_t n_rows, int64_t n_seg, int64_t n_tiles) {
    size_t s1 = 0, s2 = 0;
    (void)rocprim::radix_sort_pairs(nullptr, s1, (const uint64_t *)nullptr, (uint64_t *)nullptr, (const int32_t *)nullptr, (int32_t *)nullptr,
                                    (size_t)n_rows, 0, 37, 0, false);
    (void)rocprim::radix_sort_pairs(nullptr, s2, (const uint32_t *)nullptr, (uint32_t *)nullptr, (const int32_t *)nullptr, (int32_t *)nullptr,
                                    (size_t)(n_seg > 0 ? n_seg : 1), 0, 32, 0, false);
    const size_t n = (size_t)(n_rows > n_tiles ? n_rows : n_tiles) + 1;
    size_t tmp = s1 > s2 ? s1 : s2;
    const size_t sc = plan_scan_bytes(n);
    tmp = tmp > sc ? tmp : sc;
    // keys in / out, rows in, two flag arrays, two scans (rows stage); segment keys in / out + ids in (tiles stage)
    return align_up(tmp, 256) + 2 * align_up((size_t)n_rows * 8, 256) + 5 * align_up((size_t)n_rows * 4, 256) +
           3 * align_up((size_t)(n_seg > 0 ? n_seg : 1) * 4, 256) + 1024;
}

// Stage 1: the rows. d_order int32 [n_rows]: rows by (class, side, descending length) -- workgroup rows, wave rows, split rows (by
// row), short rows, rows below rows_from; d_long_index int32 [n_rows]; d_long_rows / d_long_seg_ptr int32 [n_rows + 1] (the first
// n_long / n_long + 1 entries are used); d_counts int64 [8] (device): rows per class 0..4, n_long, n_seg.
extern "C" int elimrec_plan_rows(const int64_t *d_rowptr, int64_t n_rows, int T, int T1, int T2, int TS, int G, int64_t side_split,
                                 int64_t rows_from, int32_t *d_order, int32_t *d_long_rows, int32_t *d_long_index, int32_t *d_long_seg_ptr,
                                 int64_t *d_counts, void *d_workspace, size_t workspace_bytes, void *stream) {
    ELIMREC_REQUIRE(d_rowptr && d_order && d_long_rows && d_long_index && d_long_seg_ptr && d_counts && d_workspace, "plan_rows: null pointer");
    ELIMREC_REQUIRE(n_rows > 0 && n_rows < INT32_MAX && T >= 1 && T1 >= T && T2 >= T1 && TS >= 1 && G >= 1 && G <= 64, "plan_rows: bad shape");
    if (workspace_bytes < elimrec_plan_workspace(n_rows, 1, 1)) { set_error("plan_rows: workspace too small"); return ELIMREC_E_WORKSPACE; }
    const PlanShape p = {n_rows, T, T1, T2, TS, G, side_split, rows_from};
    hipStream_t s = (hipStream_t)stream;
    char *ws = (char *)d_workspace;
    size_t at = 0;
    auto take = [&](size_t bytes) { char *q = ws + at; at += align_up(bytes, 256); return q; };
    uint64_t *keys = (uint64_t *)take((size_t)n_rows * 8), *keys_out = (uint64_t *)take((size_t)n_rows * 8);
    int32_t *rows = (int32_t *)take((size_t)n_rows * 4), *is_long = (int32_t *)take((size_t)n_rows * 4), *nseg = (int32_t *)take((size_t)n_rows * 4);
    int32_t *long_scan = (int32_t *)take((size_t)n_rows * 4), *seg_scan = (int32_t *)take((size_t)n_rows * 4);
    void *tmp = ws + at;
    size_t tmp_bytes = workspace_bytes - at;
    const unsigned blocks = (unsigned)((n_rows + 255) / 256 < 4096 ? (n_rows + 255) / 256 : 4096);
    hipLaunchKernelGGL(plan_keys_kernel, dim3(blocks), dim3(256), 0, s, p, d_rowptr, keys, rows, is_long, nseg);
    ELIMREC_LAUNCH_CHECK("plan_keys");
    size_t b = tmp_bytes;
    hipError_t e = rocprim::radix_sort_pairs(tmp, b, (const uint64_t *)keys, keys_out, (const int32_t *)rows, d_order, (size_t)n_rows, 0, 37, s, false);
    if (e != hipSuccess) return check_hip(e, "radix_sort_pairs(rows)");
    b = tmp_bytes;
    e = rocprim::exclusive_scan(tmp, b, (const int32_t *)is_long, long_scan, (int32_t)0, (size_t)n_rows, rocprim::plus<int32_t>(), s, false);
    if (e != hipSuccess) return check_hip(e, "exclusive_scan(long rows)");
    b = tmp_bytes;
    e = rocprim::exclusive_scan(tmp, b, (const int32_t *)nseg, seg_scan, (int32_t)0, (size_t)n_rows, rocprim::plus<int32_t>(), s, false);
    if (e != hipSuccess) return check_hip(e, "exclusive_scan(segments)");
    hipLaunchKernelGGL(plan_counts_kernel, dim3(1), dim3(64), 0, s, p, (const uint64_t *)keys_out, (const int32_t *)long_scan, (const int32_t *)is_long,
                       (const int32_t *)seg_scan, (const int32_t *)nseg, d_counts);
    ELIMREC_LAUNCH_CHECK("plan_counts");
    hipLaunchKernelGGL(plan_long_kernel, dim3(blocks), dim3(256), 0, s, p, (const int32_t *)is_long, (const int32_t *)long_scan,
                       (const int32_t *)seg_scan, d_long_rows, d_long_index, d_long_seg_ptr, (const int64_t *)d_counts);
    ELIMREC_LAUNCH_CHECK("plan_long");
    return 0;
}

// Stage 2: the tiles, given stage 1's counts on the host. Outputs: d_tile_off int64 [n_tiles + 1], d_tile_len / d_tile_dst int32
// [n_tiles * G], d_tile_long int32 [max(n_tseg * G, 1)], and for stage 3 d_gb int64 / d_gs int32 [n_tiles * G]; d_totals int64 [4]
// (device): all entries, the entries of the tiles before the short rows', the most 64-entry lines of a tile.
// n_tiles = 4 n_w4 + pad4(n_w1) + pad(n_seg, 4 G) / G + pad(n_short, 4 G) / G  (elimrec_plan_tile_count).
extern "C" int64_t elimrec_plan_tile_count(int64_t n_w4, int64_t n_w1, int64_t n_seg, int64_t n_short, int G) {
    return 4 * n_w4 + pad_to(n_w1, 4) + pad_to(n_seg, 4 * (int64_t)G) / G + pad_to(n_short, 4 * (int64_t)G) / G;
}

extern "C" int elimrec_plan_tiles(const int64_t *d_rowptr, int64_t n_rows, int T, int T1, int T2, int TS, int G, int64_t side_split,
                                  int64_t rows_from, const int32_t *d_order, const int32_t *d_long_rows, const int32_t *d_long_seg_ptr,
                                  int64_t n_w4, int64_t n_w1, int64_t n_split, int64_t n_short, int64_t n_long, int64_t n_seg,
                                  int64_t *d_tile_off, int32_t *d_tile_len, int32_t *d_tile_dst, int32_t *d_tile_long, int64_t *d_gb,
                                  int32_t *d_gs, int64_t *d_totals, void *d_workspace, size_t workspace_bytes, void *stream) {
    ELIMREC_REQUIRE(d_rowptr && d_order && d_long_rows && d_long_seg_ptr && d_tile_off && d_tile_len && d_tile_dst && d_tile_long && d_gb && d_gs &&
                    d_totals && d_workspace, "plan_tiles: null pointer");
    const PlanShape p = {n_rows, T, T1, T2, TS, G, side_split, rows_from};
    TileCounts c;
    c.n_w4 = n_w4; c.n_w1 = n_w1; c.n_split = n_split; c.n_short = n_short; c.n_seg = n_seg;
    c.n_t4 = 4 * n_w4; c.n_t1 = pad_to(n_w1, 4); c.n_tseg = pad_to(n_seg, 4 * (int64_t)G) / G; c.n_tfin = pad_to(n_short, 4 * (int64_t)G) / G;
    const int64_t n_tiles = c.n_t4 + c.n_t1 + c.n_tseg + c.n_tfin;
    if (workspace_bytes < elimrec_plan_workspace(n_rows, n_seg, n_tiles)) { set_error("plan_tiles: workspace too small"); return ELIMREC_E_WORKSPACE; }
    hipStream_t s = (hipStream_t)stream;
    char *ws = (char *)d_workspace;
    size_t at = 0;
    auto take = [&](size_t bytes) { char *q = ws + at; at += align_up(bytes, 256); return q; };
    const size_t ns = (size_t)(n_seg > 0 ? n_seg : 1);
    // (the rows stage's arrays are dead: its space holds the segments' and the tiles')
    int64_t *seg_beg = (int64_t *)take(ns * 8);
    int32_t *seg_len = (int32_t *)take(ns * 4), *seg_rank = (int32_t *)take(ns * 4), *seg_id = (int32_t *)take(ns * 4), *seg_sorted = (int32_t *)take(ns * 4);
    uint32_t *seg_key = (uint32_t *)take(ns * 4), *seg_key_out = (uint32_t *)take(ns * 4);
    int64_t *entries = (int64_t *)take((size_t)(n_tiles > 0 ? n_tiles : 1) * 8);
    void *tmp = ws + at;
    size_t tmp_bytes = workspace_bytes - at;
    if (n_seg > 0) {
        const unsigned sb = (unsigned)((n_seg + 255) / 256 < 4096 ? (n_seg + 255) / 256 : 4096);
        hipLaunchKernelGGL(plan_segments_kernel, dim3(sb), dim3(256), 0, s, p, d_rowptr, d_long_rows, d_long_seg_ptr, n_long, n_seg, seg_beg, seg_len,
                           seg_rank, seg_key, seg_id);
        ELIMREC_LAUNCH_CHECK("plan_segments");
        size_t b = tmp_bytes;
        hipError_t e = rocprim::radix_sort_pairs(tmp, b, (const uint32_t *)seg_key, seg_key_out, (const int32_t *)seg_id, seg_sorted, (size_t)n_seg, 0, 32,
                                                 s, false);
        if (e != hipSuccess) return check_hip(e, "radix_sort_pairs(segments)");
    }
    if (n_tiles > 0) {
        const int64_t ng = n_tiles * G;
        const unsigned tb = (unsigned)((ng + 255) / 256 < 8192 ? (ng + 255) / 256 : 8192);
        hipLaunchKernelGGL(plan_tiles_kernel, dim3(tb), dim3(256), 0, s, p, c, d_rowptr, d_order, (const int64_t *)seg_beg, (const int32_t *)seg_len,
                           (const int32_t *)seg_rank, (const int32_t *)seg_sorted, d_gb, d_tile_len, d_gs, d_tile_dst, d_tile_long);
        ELIMREC_LAUNCH_CHECK("plan_tiles");
        hipLaunchKernelGGL(plan_steps_kernel, dim3((unsigned)((n_tiles + 255) / 256 < 4096 ? (n_tiles + 255) / 256 : 4096)), dim3(256), 0, s, n_tiles, G,
                           (const int32_t *)d_tile_len, entries);
        ELIMREC_LAUNCH_CHECK("plan_steps");
        size_t b = tmp_bytes;
        hipError_t e = rocprim::exclusive_scan(tmp, b, (const int64_t *)entries, d_tile_off, (int64_t)0, (size_t)n_tiles, rocprim::plus<int64_t>(), s, false);
        if (e != hipSuccess) return check_hip(e, "exclusive_scan(tile offsets)");
    }
    hipLaunchKernelGGL(plan_totals_kernel, dim3(1), dim3(256), 0, s, n_tiles, c.n_t4 + c.n_t1 + c.n_tseg, (const int64_t *)d_tile_off, (const int64_t *)entries,
                       d_tile_off + n_tiles, d_totals);
    ELIMREC_LAUNCH_CHECK("plan_totals");
    return 0;
}

// Stage 3: d_tile_col / d_tile_val [total + 128], zero-filled by the caller, receive every non-zero at tile_off[tile] + step * G + group.
extern "C" int elimrec_plan_scatter(int64_t n_tiles, int G, const int64_t *d_tile_off, const int64_t *d_gb, const int32_t *d_tile_len,
                                    const int32_t *d_gs, const int32_t *d_col, const float *d_val, int32_t *d_tile_col, float *d_tile_val,
                                    void *stream) {
    ELIMREC_REQUIRE(d_tile_off && d_gb && d_tile_len && d_gs && d_col && d_val && d_tile_col && d_tile_val && G >= 1, "plan_scatter: null pointer");
    if (n_tiles <= 0) return 0;
    const int64_t ng = n_tiles * G;
    hipLaunchKernelGGL(plan_scatter_kernel, dim3((unsigned)((ng + 255) / 256 < 16384 ? (ng + 255) / 256 : 16384)), dim3(256), 0, (hipStream_t)stream, ng, G,
                       d_tile_off, d_gb, d_tile_len, d_gs, d_col, d_val, d_tile_col, d_tile_val);
    ELIMREC_LAUNCH_CHECK("plan_scatter");
    return 0;
}
