// Cosine-BPR head, its analytic gradient, the deterministic row scatter-add that replaces
// IndexBackward, and the row-sparse input gradient of the head Linears.
// (models/EliMRec.py:129-142,277-297 forward; main.py:99-100 autograd.)
#include "common.h"
#include <chrono>
#include <cstring>
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <rocprim/rocprim.hpp>

namespace elimrec {

constexpr int kMaxBlocks = 8;
struct BlockWeights { float w[kMaxBlocks]; };

__device__ __forceinline__ float4 ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
__device__ __forceinline__ void st4(float *p, float4 v) { *reinterpret_cast<float4 *>(p) = v; }
__device__ __forceinline__ float dot4(float4 a, float4 b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }

// One wave per triplet. A d-wide block needs d/4 lanes (float4 each), so the wave splits into 64/LB groups of
// LB = pow2 >= d/4 lanes and works on that many head blocks at once (d = 64: four blocks side by side instead of
// one after the other with 48 idle lanes). Reductions are xor butterflies inside a group -- the same additions
// as a full-wave butterfly whose other lanes hold zeros -- and the loss terms are added in block order.
template <bool PUBLISH>
__device__ __forceinline__ void bpr_head_body(const float *__restrict__ Y, int64_t ldy, int64_t U,
                                              const int64_t *__restrict__ users,
                                              const int64_t *__restrict__ pos,
                                              const int64_t *__restrict__ neg, int B, int d, int n_blocks,
                                              const BlockWeights &bw, float inv_b, float *__restrict__ loss_rows,
                                              float *__restrict__ grad_rows, int32_t *__restrict__ keys,
                                              const int32_t *__restrict__ slot_rows, int LB) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (b >= B) return;
    // rows of Y: node ids, or (compact table) the row of each of the triplet's three slots
    const int64_t ru = slot_rows ? (int64_t)slot_rows[3 * b] : users[b];
    const int64_t rp = slot_rows ? (int64_t)slot_rows[3 * b + 1] : U + pos[b];
    const int64_t rn = slot_rows ? (int64_t)slot_rows[3 * b + 2] : U + neg[b];
    const float *ya = Y + ru * ldy, *yp = Y + rp * ldy, *yn = Y + rn * ldy;
    const int ldg = n_blocks * d;
    float *ga = grad_rows ? grad_rows + (int64_t)(3 * b + 0) * ldg : nullptr;
    float *gp = grad_rows ? grad_rows + (int64_t)(3 * b + 1) * ldg : nullptr;
    float *gn = grad_rows ? grad_rows + (int64_t)(3 * b + 2) * ldg : nullptr;
    const float eps = 1e-12f;
    const int PB = 64 / LB, grp = lane / LB, sl = lane - grp * LB;
    float loss = 0.f;
    for (int k0 = 0; k0 < n_blocks; k0 += PB) {
        const int k = k0 + grp;
        const bool have = k < n_blocks;
        float wk = 0.f;
#pragma unroll
        for (int q = 0; q < kMaxBlocks; ++q) wk = (q == k) ? bw.w[q] : wk;
        const int off = k * d;
        float term = 0.f;
        if (have && wk == 0.f) {
            if (grad_rows)
                for (int v = sl * 4; v < d; v += 4 * LB) {
                    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
                    st4(ga + off + v, z); st4(gp + off + v, z); st4(gn + off + v, z);
                }
        }
        const bool on = have && wk != 0.f;
        float saa = 0.f, spp = 0.f, snn = 0.f, sap = 0.f, san = 0.f;
        if (on)
            for (int v = sl * 4; v < d; v += 4 * LB) {
                const float4 a = ld4(ya + off + v), p = ld4(yp + off + v), n = ld4(yn + off + v);
                saa += dot4(a, a); spp += dot4(p, p); snn += dot4(n, n); sap += dot4(a, p); san += dot4(a, n);
            }
        for (int o = LB >> 1; o > 0; o >>= 1) {
            saa += __shfl_xor(saa, o, 64); spp += __shfl_xor(spp, o, 64); snn += __shfl_xor(snn, o, 64);
            sap += __shfl_xor(sap, o, 64); san += __shfl_xor(san, o, 64);
        }
        if (on) {
            const float na = sqrtf(saa), np_ = sqrtf(spp), nn = sqrtf(snn);
            const float da = fmaxf(na, eps), dp = fmaxf(np_, eps), dn = fmaxf(nn, eps);   // F.normalize denominators
            const float cp = sap / (da * dp), cn = san / (da * dn);
            const float x = cn - cp;
            const float sp = (x > 20.f) ? x : log1pf(expf(x));                           // F.softplus
            term = wk * sp * inv_b;
            if (grad_rows) {
                const float sig = (x > 20.f) ? 1.f : 1.f / (1.f + expf(-x));
                const float g = wk * sig * inv_b;
                // d/da of  <a/da, n/dn - p/dp>: (v - ahat*<ahat,v>)/da when |a| > eps, v/eps otherwise.
                const bool fa = na > eps, fp = np_ > eps, fn = nn > eps;
                for (int v = sl * 4; v < d; v += 4 * LB) {
                    const float4 a = ld4(ya + off + v), p = ld4(yp + off + v), n = ld4(yn + off + v);
                    float4 ra, rp4, rn4;
#define ELIMREC_BPR_COMP(c)                                                              \
    {                                                                                    \
        const float ah = a.c / da, ph = p.c / dp, nh = n.c / dn;                         \
        const float vv = nh - ph;                                                        \
        ra.c = g * (fa ? (vv - ah * x) / da : vv / eps);                                 \
        rp4.c = -g * (fp ? (ah - ph * cp) / dp : ah / eps);                              \
        rn4.c = g * (fn ? (ah - nh * cn) / dn : ah / eps);                               \
    }
                    ELIMREC_BPR_COMP(x) ELIMREC_BPR_COMP(y) ELIMREC_BPR_COMP(z) ELIMREC_BPR_COMP(w)
#undef ELIMREC_BPR_COMP
                    st4(ga + off + v, ra); st4(gp + off + v, rp4); st4(gn + off + v, rn4);
                }
            }
        }
        for (int q = 0; q < PB; ++q) {                   // block order: the loss is the same sum as one block at a time
            const float t = __shfl(term, q * LB, 64);
            if (k0 + q < n_blocks) loss += t;
        }
    }
    if (lane == 0) {
        if (PUBLISH) {      // write-through: another workgroup of this launch adds the rows up (bpr_head_sum_kernel)
            __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)loss_rows, 0, B * 4, 0x00020000);
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(loss), rs, (unsigned)b * 4u, 0, 16 /* sc1 */);
        } else {
            loss_rows[b] = loss;
        }
        if (keys) { keys[3 * b] = (int32_t)ru; keys[3 * b + 1] = (int32_t)rp; keys[3 * b + 2] = (int32_t)rn; }
    }
}

__global__ __launch_bounds__(256) void bpr_head_kernel(const float *__restrict__ Y, int64_t ldy, int64_t U,
                                                       const int64_t *__restrict__ users,
                                                       const int64_t *__restrict__ pos,
                                                       const int64_t *__restrict__ neg, int B, int d, int n_blocks,
                                                       BlockWeights bw, float inv_b, float *__restrict__ loss_rows,
                                                       float *__restrict__ grad_rows, int32_t *__restrict__ keys,
                                                       const int32_t *__restrict__ slot_rows, int LB) {
    bpr_head_body<false>(Y, ldy, U, users, pos, neg, B, d, n_blocks, bw, inv_b, loss_rows, grad_rows, keys, slot_rows, LB);
}

// bpr_head over the compact rows + the batch loss in the same launch: every workgroup publishes its loss rows
// write-through, drains them and draws a ticket; the one that draws the last ticket does one agent-scope acquire and
// adds all B rows in sum_kernel's order -- thread t of 1024 virtual threads adds x[t], x[t+1024], ..., then the binary
// tree -- so the loss has the bits of elimrec_sum over the same rows (cdna_hip_programming.md G16, counter form).
__global__ __launch_bounds__(256) void bpr_head_sum_kernel(const float *__restrict__ Y, int64_t ldy, int B, int d, int n_blocks,
                                                           BlockWeights bw, float inv_b, float *__restrict__ loss_rows,
                                                           float *__restrict__ grad_rows, const int32_t *__restrict__ slot_rows,
                                                           int LB, float *__restrict__ loss_out, int32_t *__restrict__ ticket,
                                                           unsigned long long *pub_slots, int n_pub, uint32_t *__restrict__ pub_counter) {
    __shared__ float s[1024];
    __shared__ int is_last;
    bpr_head_body<true>(Y, ldy, 0, nullptr, nullptr, nullptr, B, d, n_blocks, bw, inv_b, loss_rows, grad_rows, nullptr, slot_rows, LB);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const int tk = __hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        is_last = tk == (int)gridDim.x - 1;
    }
    __syncthreads();
    if (!is_last) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    for (int vt = threadIdx.x; vt < 1024; vt += 256) {
        float acc = 0.f;
        for (int i = vt; i < B; i += 1024) acc += loss_rows[i];
        s[vt] = acc;
    }
    __syncthreads();
    for (int w = 512; w > 0; w >>= 1) {
        for (int idx = threadIdx.x; idx < w; idx += 256) s[idx] += s[idx + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        loss_out[0] = s[0];
        __hip_atomic_store(ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (pub_slots) {
            // the loss to the host, 120 us into the step instead of behind its last launch: (sequence number, value) as ONE 8-byte
            // system-scope store into coherent host memory. The sequence number counts the publishing launches of this stream
            // (one thread per launch touches the counter; launches of a stream are ordered) -- the host counts the same way.
            const uint32_t q = pub_counter[0] + 1u;
            pub_counter[0] = q;
            __hip_atomic_store(pub_slots + (q % (uint32_t)n_pub), ((unsigned long long)q << 32) | (unsigned long long)__float_as_uint(s[0]),
                               __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// Fixed-order sum: thread t adds x[t], x[t+1024], ...; then a binary tree over the 1024 partials.
__global__ __launch_bounds__(1024) void sum_kernel(const float *__restrict__ x, int64_t n, float *__restrict__ out) {
    __shared__ float s[1024];
    float acc = 0.f;
    for (int64_t i = threadIdx.x; i < n; i += 1024) acc += x[i];
    s[threadIdx.x] = acc;
    __syncthreads();
    for (int w = 512; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) s[threadIdx.x] += s[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = s[0];
}

// ------------------------------------------------------------------ segment reduce
__global__ void iota_kernel(int32_t *v, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) v[i] = (int32_t)i;
}

__global__ void heads_kernel(const int32_t *__restrict__ ks, int64_t n, int32_t *__restrict__ flag) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) flag[i] = (i == 0 || ks[i] != ks[i - 1]) ? 1 : 0;
}

__global__ void finalize_segments_kernel(const int32_t *__restrict__ ks, const int32_t *__restrict__ flag,
                                         const int32_t *__restrict__ segid, int64_t n, int32_t split_key,
                                         int32_t *__restrict__ active_rows, int32_t *__restrict__ seg_start,
                                         int32_t *__restrict__ seg_info) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int32_t s = segid[i] - 1;
    if (flag[i]) { active_rows[s] = ks[i]; seg_start[s] = (int32_t)i; }
    if (i == n - 1) { seg_info[0] = s + 1; seg_start[s + 1] = (int32_t)n; }
    if (i == 0 && ks[0] >= split_key) seg_info[1] = 0;
    if (ks[i] < split_key && (i == n - 1 || ks[i + 1] >= split_key)) seg_info[1] = s + 1;
}

// publishes the (begin,end) slot ranges linear_bwd_w consumes (seg_info layout: see the header)
__global__ void publish_ranges_kernel(int32_t *__restrict__ seg_info) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const int na = seg_info[0], nl = seg_info[1];
        seg_info[2] = 0; seg_info[3] = nl; seg_info[4] = nl; seg_info[5] = na; seg_info[6] = 0; seg_info[7] = na;
    }
}

// ------------------------------------------------------------------ segment plan in ONE workgroup
// Keys are node ids below `key_space`, so "sort and unique" is a bitmap: set the bit of every key (LDS atomics,
// order irrelevant), prefix-sum the word popcounts, and the rank of a key -- its segment -- is
// prefix[word] + popc(bits below it). Segments come out ascending by node id exactly as the radix-sort path
// produces them; member lists are filled with atomics and then put in ascending slot order (insertion sort for
// the usual 1-3 members, a cooperative rank sort for the rare long list), so the plan is a pure function of the
// key list. One launch of 1024 threads replaces iota + radix sort (6 launches) + heads + scan (2) + finalize +
// publish.
constexpr int PLAN_T = 1024;
constexpr int PLAN_LONG = 8;       // member lists longer than this are rank-sorted by a wave (or the workgroup)
constexpr int PLAN_HUGE = 512;

// exclusive prefix of one value per thread over the 1024-thread workgroup; *total = sum of all values
__device__ __forceinline__ int plan_block_scan(int v, int *wave_sums, int *total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(inc, o, 64);
        if (lane >= o) inc += t;
    }
    __syncthreads();                                  // wave_sums may still be read from a previous scan
    if (lane == 63) wave_sums[wave] = inc;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < PLAN_T / 64; ++w) {
        const int t = wave_sums[w];
        if (w < wave) base += t;
        tot += t;
    }
    *total = tot;
    return base + inc - v;
}

// Up to PLAN_LDS_N slots: counts, offsets and member lists live in LDS and the keys in registers, so the ~60
// dependent round trips of the phases below cost LDS latency; the results are copied out at the end. (The same
// phases on global arrays cost an L2 round trip each -- measured 4x slower than the radix-sort path at 49 k slots,
// which therefore keeps the larger key lists.)
// optional batch front end of the planner: the keys are the node ids of B triplets, computed (and range-checked) here
struct PlanBatch {
    const int64_t *users, *pos, *neg;
    int64_t U, I;
    int32_t *keys_out, *err;
    int32_t pad_key;
    int pad;
};

constexpr int PLAN_LDS_N = 8192;
constexpr int PLAN_KPT = PLAN_LDS_N / PLAN_T;

__global__ __launch_bounds__(PLAN_T) void segment_plan_kernel(const int32_t *__restrict__ keys, int n, int key_space,
                                                              int split_key, int32_t *__restrict__ active_rows,
                                                              int32_t *__restrict__ seg_info,
                                                              int32_t *__restrict__ slot_seg,
                                                              int32_t *__restrict__ g_seg_start,
                                                              int32_t *__restrict__ g_members,
                                                              uint32_t *__restrict__ key_bitmap, PlanBatch pb) {
    extern __shared__ uint32_t plan_lds[];
    const int nw = (key_space + 31) >> 5;
    uint32_t *bm = plan_lds, *pre = plan_lds + nw;
    int *wave_sums = (int *)(pre + nw);               // [16]
    int *long_list = wave_sums + 16;                  // [1 + 1023]: count, then segment ids of long member lists
    int32_t *cnt = (int32_t *)(long_list + 1024);      // [PLAN_LDS_N]; doubles as the rank-sort scratch at the end
    int32_t *seg_start = cnt + PLAN_LDS_N;            // [PLAN_LDS_N + 1]
    int32_t *members = seg_start + PLAN_LDS_N + 1;    // [PLAN_LDS_N]
    int32_t *tmp = cnt;
    auto ld = [&](const int32_t *p) -> int32_t { return *p; };
    const int tid = threadIdx.x;
    int kreg[PLAN_KPT];
#pragma unroll
    for (int i = 0; i < PLAN_KPT; ++i) {
        const int j = tid + i * PLAN_T;
        if (j >= n) { kreg[i] = -1; continue; }
        if (!pb.users) { kreg[i] = keys[j]; continue; }
        // the node id of triplet slot j, range-checked as elimrec_triplet_rows_checked does (slot 3b + {0,1,2})
        const int b = j / 3, which = j - 3 * b;
        int64_t idx = which == 0 ? pb.users[b] : (which == 1 ? pb.pos[b] : pb.neg[b]);
        const int64_t lim = which == 0 ? pb.U : pb.I;
        if (idx < 0 || idx >= lim) {
            if (pb.err) atomicOr(pb.err, 1 << which);
            idx = 0;
        }
        kreg[i] = (int)(which == 0 ? idx : pb.U + idx);
        if (pb.keys_out) pb.keys_out[j] = kreg[i];
    }
    constexpr int iters = PLAN_KPT;
    auto key_of = [&](int i, int j) -> int { return kreg[i]; };
    for (int w = tid; w < nw; w += PLAN_T) bm[w] = 0u;
    if (tid == 0) long_list[0] = 0;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < iters; ++i) {
        const int j = tid + i * PLAN_T;
        if (j < n) {
            const uint32_t k = (uint32_t)key_of(i, j);
            atomicOr(&bm[k >> 5], 1u << (k & 31));
        }
    }
    __syncthreads();
    // ranks: exclusive prefix of the word popcounts
    const int per = (nw + PLAN_T - 1) / PLAN_T;
    const int w0 = min(tid * per, nw), w1 = min(w0 + per, nw);
    int local = 0;
    for (int w = w0; w < w1; ++w) local += __popc(bm[w]);
    int n_act;
    int run = plan_block_scan(local, wave_sums, &n_act);
    for (int w = w0; w < w1; ++w) {
        pre[w] = (uint32_t)run;
        uint32_t bits = bm[w];
        if (key_bitmap) key_bitmap[w] = bits;
        while (bits) {
            const int b = __ffs(bits) - 1;
            active_rows[run] = w * 32 + b;
            cnt[run] = 0;
            ++run;
            bits &= bits - 1;
        }
    }
    __syncthreads();
    int sreg[PLAN_KPT];
#pragma unroll
    for (int i = 0; i < iters; ++i) {
        const int j = tid + i * PLAN_T;
        if (j < n) {
            const uint32_t k = (uint32_t)key_of(i, j);
            const int seg = (int)pre[k >> 5] + __popc(bm[k >> 5] & ((1u << (k & 31)) - 1u));
            slot_seg[j] = seg;
            sreg[i] = seg;
            atomicAdd(&cnt[seg], 1);
        }
    }
    __syncthreads();
    // member-list offsets: exclusive prefix of the counts; cnt becomes the fill cursor
    const int per2 = (n_act + PLAN_T - 1) / PLAN_T;
    const int s0 = min(tid * per2, n_act), s1 = min(s0 + per2, n_act);
    local = 0;
    for (int sgm = s0; sgm < s1; ++sgm) local += ld(cnt + sgm);
    int total;
    run = plan_block_scan(local, wave_sums, &total);
    for (int sgm = s0; sgm < s1; ++sgm) {
        const int c = ld(cnt + sgm);
        seg_start[sgm] = run;
        cnt[sgm] = 0;
        if (c > PLAN_LONG) {
            const int q = atomicAdd(&long_list[0], 1);
            if (q < 1023) long_list[1 + q] = sgm;
        }
        run += c;
    }
    if (tid == 0) seg_start[n_act] = n;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < iters; ++i) {
        const int j = tid + i * PLAN_T;
        if (j < n) {
            const int seg = sreg[i];
            const int pos = ld(seg_start + seg) + atomicAdd(&cnt[seg], 1);
            members[pos] = j;
        }
    }
    __syncthreads();
    const int n_long = long_list[0];
    const bool all_serial = n_long > 1023;             // more long lists than the table holds: sort them serially too
    for (int sgm = tid; sgm < n_act; sgm += PLAN_T) {
        const int b = ld(seg_start + sgm), e = ld(seg_start + sgm + 1);
        if (e - b > PLAN_LONG && !all_serial) continue;
        for (int i = b + 1; i < e; ++i) {             // insertion sort, ascending slot
            const int v = ld(members + i);
            int q = i - 1;
            while (q >= b) {
                const int u = ld(members + q);
                if (u <= v) break;
                members[q + 1] = u;
                --q;
            }
            members[q + 1] = v;
        }
    }
    if (!all_serial) {
        // rank sort (slots are distinct, so the ranks are a permutation): one wave per list, lists of more than
        // PLAN_HUGE members by the whole workgroup. tmp[b, e) is private to the list.
        const int lane = tid & 63, wave = tid >> 6;
        for (int li = wave; li < n_long; li += PLAN_T / 64) {
            const int sgm = long_list[1 + li];
            const int b = ld(seg_start + sgm), e = ld(seg_start + sgm + 1);
            if (e - b > PLAN_HUGE) continue;
            for (int i = b + lane; i < e; i += 64) {
                const int v = ld(members + i);
                int r = 0;
                for (int q = b; q < e; ++q) r += ld(members + q) < v;
                tmp[b + r] = v;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            for (int i = b + lane; i < e; i += 64) members[i] = ld(tmp + i);
        }
        for (int li = 0; li < n_long; ++li) {
            const int sgm = long_list[1 + li];
            const int b = ld(seg_start + sgm), e = ld(seg_start + sgm + 1);
            if (e - b <= PLAN_HUGE) continue;         // uniform over the workgroup
            __syncthreads();
            for (int i = b + tid; i < e; i += PLAN_T) {
                const int v = ld(members + i);
                int r = 0;
                for (int q = b; q < e; ++q) r += ld(members + q) < v;
                tmp[b + r] = v;
            }
            __syncthreads();
            for (int i = b + tid; i < e; i += PLAN_T) members[i] = ld(tmp + i);
        }
    }
    __syncthreads();                                  // publish the member lists for segment_apply
    for (int i = tid; i < n; i += PLAN_T) g_members[i] = members[i];
    for (int i = tid; i <= n_act; i += PLAN_T) g_seg_start[i] = seg_start[i];
    if (tid == 0) {
        int n_lo = n_act;
        if (split_key < key_space) {
            const uint32_t k = (uint32_t)(split_key < 0 ? 0 : split_key);
            n_lo = (int)pre[k >> 5] + __popc(bm[k >> 5] & ((1u << (k & 31)) - 1u));
        }
        seg_info[0] = n_act; seg_info[1] = n_lo;
        seg_info[2] = 0; seg_info[3] = n_lo; seg_info[4] = n_lo; seg_info[5] = n_act; seg_info[6] = 0; seg_info[7] = n_act;
    }
    if (pb.pad)                                       // unused tail of the active-row list: distinct negative keys
        for (int r = n_act + tid; r < n; r += PLAN_T) active_rows[r] = pb.pad_key + r;
}


// ------------------------------------------------------------------ the same bitmap plan on the whole device
// Key lists beyond the one-workgroup planner's LDS (more than PLAN_LDS_N slots: batches above 2730 triplets) took the
// radix-sort path: iota + 6 sort launches + heads + 2 scan launches + finalize + publish + slot map + bitmap + padding, some 22
// launches of ~5 us each on the step's second stream -- 160 us, twice what the main stream has to do before it needs the plan
// (a cliff of +0.1 ms per step between B = 2048 and B = 4096). The bitmap algorithm needs no sort: the same phases as
// segment_plan_kernel, each as one launch over the slots (or one workgroup for the two prefix sums), global atomics instead
// of LDS atomics, and the member lists put in ascending slot order afterwards (a thread per short list, a wave per long one):
// 7 launches, the same plan bit for bit.
__global__ __launch_bounds__(256) void plan_bits_kernel(const int32_t *__restrict__ keys, int64_t n, uint32_t *__restrict__ bm,
                                                         int32_t *__restrict__ cnt, int32_t *__restrict__ cursor, PlanBatch pb) {
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j > n) return;
    cnt[j] = 0;                                                   // counters [0, n] (one past the last possible segment) ...
    if (j == n) return;
    cursor[j] = 0;                                                // ... and the fill cursors
    int k;
    if (!pb.users) k = keys[j];
    else {
        const int64_t b = j / 3;
        const int which = (int)(j - 3 * b);
        int64_t idx = which == 0 ? pb.users[b] : (which == 1 ? pb.pos[b] : pb.neg[b]);
        const int64_t lim = which == 0 ? pb.U : pb.I;
        if (idx < 0 || idx >= lim) {
            if (pb.err) atomicOr(pb.err, 1 << which);
            idx = 0;
        }
        k = (int)(which == 0 ? idx : pb.U + idx);
        if (pb.keys_out) pb.keys_out[j] = k;
    }
    atomicOr(&bm[(uint32_t)k >> 5], 1u << (k & 31));
}

// ranks: pre[] = exclusive prefix of the bitmap words' popcounts, a device-wide scan (rocPRIM over a popcount iterator: a
// one-workgroup walk over the words and their rows took 51 us at the Tiktok shape). Then, one launch: a thread per bitmap word
// writes the word's rows into the ascending list of active rows (and the two threads that know them, seg_info), a thread per
// slot looks its segment up and counts it, the unused tail of the row list gets its padding keys.
struct PopcOp { __device__ int32_t operator()(uint32_t w) const { return (int32_t)__popc(w); } };

__global__ __launch_bounds__(256) void plan_slots_kernel(const int32_t *__restrict__ keys, int64_t n, const uint32_t *__restrict__ bm,
                                                          int nw, const int32_t *__restrict__ pre, int split_key, int key_space,
                                                          int32_t *__restrict__ active_rows, int32_t *__restrict__ seg_info,
                                                          int32_t *__restrict__ long_list, int32_t *__restrict__ slot_seg,
                                                          int32_t *__restrict__ cnt, int pad, int32_t pad_key) {
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int n_act = pre[nw - 1] + __popc(bm[nw - 1]);
    if (j < nw) {
        int run = pre[j];
        uint32_t bits = bm[j];
        while (bits) {
            const int b = __ffs(bits) - 1;
            active_rows[run++] = (int)j * 32 + b;
            bits &= bits - 1;
        }
        if (j == 0) {
            int n_lo = n_act;
            if (split_key < key_space) {                            // rank of the first key >= split_key
                const uint32_t k = (uint32_t)(split_key < 0 ? 0 : split_key);
                n_lo = pre[k >> 5] + __popc(bm[k >> 5] & ((1u << (k & 31)) - 1u));
            }
            seg_info[0] = n_act; seg_info[1] = n_lo;
            seg_info[2] = 0; seg_info[3] = n_lo; seg_info[4] = n_lo; seg_info[5] = n_act; seg_info[6] = 0; seg_info[7] = n_act;
            long_list[0] = 0;
        }
    }
    if (j >= n) return;
    if (pad && j >= n_act) active_rows[j] = pad_key + (int32_t)j;
    const uint32_t k = (uint32_t)keys[j];
    const int seg = pre[k >> 5] + __popc(bm[k >> 5] & ((1u << (k & 31)) - 1u));
    slot_seg[j] = seg;
    atomicAdd(&cnt[seg], 1);
}

// (member-list offsets: a device-wide exclusive scan of the counters [0, n], rocPRIM -- a one-workgroup scan walks 60 000
// counters in global memory one L2 round trip at a time)
__global__ __launch_bounds__(256) void plan_fill_kernel(const int32_t *__restrict__ slot_seg, int64_t n, const int32_t *__restrict__ seg_start,
                                                         int32_t *__restrict__ cursor, int32_t *__restrict__ members) {
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    const int seg = slot_seg[j];
    members[seg_start[seg] + atomicAdd(&cursor[seg], 1)] = (int32_t)j;
}

// ascending slot order inside every member list (the atomics above filled them in arrival order): a thread per short list
// (insertion sort), then a wave per long list (rank sort through tmp: slots are distinct, the ranks are a permutation)
__global__ __launch_bounds__(256) void plan_sort_short_kernel(const int32_t *__restrict__ seg_info, const int32_t *__restrict__ seg_start,
                                                               int32_t *__restrict__ members, int32_t *__restrict__ long_list) {
    const int sgm = (int)(blockIdx.x * 256 + threadIdx.x);
    if (sgm >= seg_info[0]) return;
    const int b = seg_start[sgm], e = seg_start[sgm + 1];
    if (e - b > PLAN_LONG) {                                      // a long list: into the table plan_sort_long_kernel walks
        long_list[1 + atomicAdd(&long_list[0], 1)] = sgm;
        return;
    }
    int v[PLAN_LONG];
#pragma unroll
    for (int i = 0; i < PLAN_LONG; ++i) v[i] = b + i < e ? members[b + i] : INT32_MAX;
#pragma unroll
    for (int i = 1; i < PLAN_LONG; ++i) {                     // a fixed 8-element insertion network on registers
#pragma unroll
        for (int q = i; q > 0; --q) {
            const int lo = min(v[q - 1], v[q]), hi = max(v[q - 1], v[q]);
            v[q - 1] = lo; v[q] = hi;
        }
    }
#pragma unroll
    for (int i = 0; i < PLAN_LONG; ++i)
        if (b + i < e) members[b + i] = v[i];
}

// a workgroup per long list: up to 64 members by the first wave (every lane ranks its member against the others with
// cross-lane reads), up to PLAN_SORT_LDS by a bitonic sort in LDS, longer ones (batches beyond ~150 k triplets) by a rank sort
// through global scratch
constexpr int PLAN_SORT_LDS = 8192;
__global__ __launch_bounds__(256) void plan_sort_long_kernel(const int32_t *__restrict__ long_list, const int32_t *__restrict__ seg_start,
                                                              int32_t *__restrict__ members, int32_t *__restrict__ tmp) {
    __shared__ int32_t buf[PLAN_SORT_LDS];
    const int tid = threadIdx.x, lane = tid & 63;
    const int n_long = long_list[0];
    for (int li = (int)blockIdx.x; li < n_long; li += (int)gridDim.x) {
        const int sgm = long_list[1 + li];
        const int b = seg_start[sgm], e = seg_start[sgm + 1], m = e - b;
        if (m <= 64) {
            if (tid < 64) {
                const int v = lane < m ? members[b + lane] : INT32_MAX;
                int r = 0;
                for (int q = 0; q < m; ++q) r += __shfl(v, q, 64) < v;
                if (lane < m) members[b + r] = v;
            }
            continue;
        }
        if (m <= PLAN_SORT_LDS) {
            int p2 = 128;
            while (p2 < m) p2 <<= 1;
            __syncthreads();                                      // buf of the previous list
            for (int i = tid; i < p2; i += 256) buf[i] = i < m ? members[b + i] : INT32_MAX;
            __syncthreads();
            for (int k = 2; k <= p2; k <<= 1) {
                for (int jj = k >> 1; jj > 0; jj >>= 1) {
                    for (int i = tid; i < p2; i += 256) {
                        const int x = i ^ jj;
                        if (x > i) {
                            const int a0 = buf[i], a1 = buf[x];
                            const bool up = (i & k) == 0;
                            if ((a0 > a1) == up) { buf[i] = a1; buf[x] = a0; }
                        }
                    }
                    __syncthreads();
                }
            }
            for (int i = tid; i < m; i += 256) members[b + i] = buf[i];
            continue;
        }
        for (int i = b + tid; i < e; i += 256) {                 // (rare) rank sort through tmp[b, e), copied back by the next launch
            const int v = members[i];
            int r = 0;
            for (int q = b; q < e; ++q) r += members[q] < v;
            tmp[b + r] = v;
        }
    }
}
__global__ __launch_bounds__(256) void plan_copy_long_kernel(const int32_t *__restrict__ long_list, const int32_t *__restrict__ seg_start,
                                                              int32_t *__restrict__ members, const int32_t *__restrict__ tmp) {
    const int n_long = long_list[0];
    for (int li = (int)blockIdx.x; li < n_long; li += (int)gridDim.x) {
        const int sgm = long_list[1 + li];
        const int b = seg_start[sgm], e = seg_start[sgm + 1];
        if (e - b <= PLAN_SORT_LDS) continue;
        for (int i = b + (int)threadIdx.x; i < e; i += 256) members[i] = tmp[i];
    }
}

__global__ void pad_keys_kernel(int32_t *__restrict__ keys, const int32_t *__restrict__ count, int64_t n, int32_t pad_key) {
    const int64_t r = (int64_t)*count + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r < n) keys[r] = pad_key + (int32_t)r;
}

__global__ void key_bitmap_kernel(const int32_t *__restrict__ keys, int64_t n, uint32_t *__restrict__ bitmap) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) atomicOr(&bitmap[keys[i] >> 5], 1u << (keys[i] & 31));
}

// segment of every slot from the sorted path's arrays (slot_seg[src[i]] = segment holding position i)
__global__ void slot_segments_kernel(const int32_t *__restrict__ src, const int32_t *__restrict__ segid, int64_t n,
                                     int32_t *__restrict__ slot_seg) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) slot_seg[src[i]] = segid[i] - 1;
}

__global__ __launch_bounds__(256) void segment_sum_kernel(const float *__restrict__ rows,
                                                          const int32_t *__restrict__ src,
                                                          const int32_t *__restrict__ seg_start,
                                                          const int32_t *__restrict__ seg_info, int64_t n, int ld4_,
                                                          const float *__restrict__ scale, float *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t s = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (s >= n || s >= seg_info[0]) return;
    const int beg = seg_start[s], end = seg_start[s + 1];
    const float sc = scale ? scale[0] : 1.f;
    const float4 *r4 = reinterpret_cast<const float4 *>(rows);
    float4 *o4 = reinterpret_cast<float4 *>(out);
    for (int c = lane; c < ld4_; c += 64) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int i = beg; i < end; ++i) {
            const float4 v = r4[(int64_t)src[i] * ld4_ + c];
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
        o4[s * ld4_ + c] = make_float4(acc.x * sc, acc.y * sc, acc.z * sc, acc.w * sc);
    }
}

// ------------------------------------------------------------------ head backward wrt input
constexpr int kMaxHeads = 4;
struct HeadPtrs { const float *w[kMaxHeads]; int mblock[kMaxHeads]; };

// A workgroup owns HB_ROWS active rows: their dY rows are staged in LDS once, every thread owns one
// output column (of each 256-column chunk) for all rows, so a weight element is loaded once per
// workgroup column and reused HB_ROWS times; dY values are LDS broadcasts.
constexpr int HB_ROWS = 16;

__global__ __launch_bounds__(256) void head_bwd_input_kernel(const float *__restrict__ dY, int64_t lddy,
                                                             const int32_t *__restrict__ active_rows,
                                                             const int32_t *__restrict__ seg_info, int64_t n_max,
                                                             int64_t U, int d, int C, int S, HeadPtrs hp,
                                                             const float *__restrict__ W_user,
                                                             const float *__restrict__ W_item, float gscale,
                                                             float *__restrict__ G0, int64_t ldg, int scatter_cols,
                                                             float *__restrict__ compact) {
    extern __shared__ float dys[];                       // [HB_ROWS][Cy]
    __shared__ int64_t node[HB_ROWS];
    const int Cy = (1 + S) * d;
    const int tid = threadIdx.x;
    const int64_t s0 = (int64_t)blockIdx.x * HB_ROWS;
    int64_t n_act = seg_info[0];
    if (n_act > n_max) n_act = n_max;
    if (s0 >= n_act) return;
    const int rows = (int)((n_act - s0) < HB_ROWS ? (n_act - s0) : HB_ROWS);
    for (int e = tid * 4; e < HB_ROWS * Cy; e += 1024) {
        const int r = e / Cy, c = e - r * Cy;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r < rows) v = ld4(dY + (s0 + r) * lddy + c);
        st4(dys + r * Cy + c, v);
    }
    if (tid < HB_ROWS) node[tid] = (tid < rows) ? (int64_t)active_rows[s0 + tid] : -1;
    __syncthreads();
    const bool any_user = node[0] < U;                   // rows are sorted by node id: users first
    const bool any_item = node[rows - 1] >= U;
    for (int c = tid; c < C; c += 256) {
        float acc[HB_ROWS];
#pragma unroll
        for (int r = 0; r < HB_ROWS; ++r) acc[r] = 0.f;
        if (any_user != any_item) {                      // the common case: one weight matrix for the whole tile
            const float *Wf = any_user ? W_user : W_item;
            // weight rows are prefetched 8 deep (L2-resident, ~0.5 us away); dY comes as 16-B LDS broadcasts
            for (int k = 0; k < d; k += 8) {
                float w[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) w[q] = (k + q < d) ? Wf[(int64_t)(k + q) * C + c] : 0.f;
#pragma unroll
                for (int r = 0; r < HB_ROWS; ++r) {
                    const float4 y0 = ld4(dys + r * Cy + k);
                    acc[r] = fmaf(y0.x, w[0], acc[r]); acc[r] = fmaf(y0.y, w[1], acc[r]);
                    acc[r] = fmaf(y0.z, w[2], acc[r]); acc[r] = fmaf(y0.w, w[3], acc[r]);
                    if (k + 4 < d) {
                        const float4 y1 = ld4(dys + r * Cy + k + 4);
                        acc[r] = fmaf(y1.x, w[4], acc[r]); acc[r] = fmaf(y1.y, w[5], acc[r]);
                        acc[r] = fmaf(y1.z, w[6], acc[r]); acc[r] = fmaf(y1.w, w[7], acc[r]);
                    }
                }
            }
        } else {                                         // the one tile that straddles the user/item boundary
            for (int k = 0; k < d; ++k) {
                const float wu = W_user[(int64_t)k * C + c], wi = W_item[(int64_t)k * C + c];
#pragma unroll
                for (int r = 0; r < HB_ROWS; ++r) acc[r] = fmaf(dys[r * Cy + k], (node[r] < U) ? wu : wi, acc[r]);
            }
        }
        const int mb = c / d;                            // table block of this column
        for (int h = 0; h < S; ++h) {
            if (hp.mblock[h] != mb) continue;
            const float *Wh = hp.w[h] + (c - mb * d);
            const float *dyh = dys + (1 + h) * d;
            for (int k = 0; k < d; k += 8) {
                float w[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) w[q] = (k + q < d) ? Wh[(int64_t)(k + q) * d] : 0.f;
#pragma unroll
                for (int r = 0; r < HB_ROWS; ++r) {
                    const float4 y0 = ld4(dyh + r * Cy + k);
                    acc[r] = fmaf(y0.x, w[0], acc[r]); acc[r] = fmaf(y0.y, w[1], acc[r]);
                    acc[r] = fmaf(y0.z, w[2], acc[r]); acc[r] = fmaf(y0.w, w[3], acc[r]);
                    if (k + 4 < d) {
                        const float4 y1 = ld4(dyh + r * Cy + k + 4);
                        acc[r] = fmaf(y1.x, w[4], acc[r]); acc[r] = fmaf(y1.y, w[5], acc[r]);
                        acc[r] = fmaf(y1.z, w[6], acc[r]); acc[r] = fmaf(y1.w, w[7], acc[r]);
                    }
                }
            }
        }
#pragma unroll
        for (int r = 0; r < HB_ROWS; ++r)
            if (r < rows) {
                const float v = acc[r] * gscale;
                if (G0 && c < scatter_cols) G0[node[r] * ldg + c] = v;
                if (compact) compact[(s0 + r) * C + c] = v;
            }
    }
}

// MFMA form of the same contraction (used when d % 32 == 0): a workgroup owns 32 active rows, stages
// their dY rows in LDS (row stride Cy+1: the 32 lanes of a half-wave read 32 different rows at one k),
// and each wave produces 32x32 output tiles with v_mfma_f32_32x32x2_f32: K runs over the d fused
// columns (B operand = rows of W_user / W_item, read from L2 eight k-steps ahead) and then over the d
// columns of the single-modal head that feeds this table block. A tile that straddles the user/item
// boundary (rows are sorted by node id) accumulates both weight matrices and selects per row.
typedef float v16f_ __attribute__((ext_vector_type(16)));
constexpr int HM_ROWS = 32;

__device__ __forceinline__ v16f_ hm_accumulate(v16f_ acc, const float *__restrict__ a_lds, int a_stride,
                                               const float *__restrict__ Wp, int64_t ldw, int K, int li, int lk) {
    // acc[row i][col j] += sum_k a_lds[i*a_stride + k] * Wp[k*ldw + j]
    // The loop is L2-latency bound, not MFMA bound: 16 B-operand loads (a 32-deep K chunk) are issued
    // back to back, then their 16 MFMAs; eight waves per workgroup interleave these phases.
    constexpr int PF = 16;
    for (int k0 = 0; k0 < K; k0 += 2 * PF) {
        float b[PF];
#pragma unroll
        for (int q = 0; q < PF; ++q) {
            const int k = k0 + 2 * q + lk;
            b[q] = (k < K) ? Wp[(int64_t)k * ldw + li] : 0.f;
        }
#pragma unroll
        for (int q = 0; q < PF; ++q) {
            const int k = k0 + 2 * q + lk;
            const float a = (k < K) ? a_lds[li * a_stride + k] : 0.f;
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b[q], acc, 0, 0, 0);
        }
    }
    return acc;
}

// optional fused segment reduce in front of head_bwd_input_mfma_kernel
struct SegSrc {
    const float *rows;            // [n x Cy] gradient rows, or nullptr: dY is read as given
    const int32_t *members;       // member slots grouped by segment (elimrec_segment_plan)
    const int32_t *seg_start;
    const float *scale;           // nullable device fp32[1]
    float *reduced;               // [n x lddy] receives dY
};

__global__ __launch_bounds__(512) void head_bwd_input_mfma_kernel(const float *__restrict__ dY, int64_t lddy,
                                                                  const int32_t *__restrict__ active_rows,
                                                                  const int32_t *__restrict__ seg_info, int64_t n_max,
                                                                  int64_t U, int d, int C, int S, HeadPtrs hp,
                                                                  const float *__restrict__ W_user,
                                                                  const float *__restrict__ W_item, float gscale,
                                                                  float *__restrict__ G0, int64_t ldg, int scatter_cols,
                                                                  float *__restrict__ compact, SegSrc seg) {
    extern __shared__ float dys[];                       // [HM_ROWS][Cy + 1]
    __shared__ int64_t node[HM_ROWS];
    const int Cy = (1 + S) * d, ldy = Cy + 1;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t s0 = (int64_t)blockIdx.x * HM_ROWS;
    int64_t n_act = seg_info[0];
    if (n_act > n_max) n_act = n_max;
    if (s0 >= n_act) return;
    const int rows = (int)((n_act - s0) < HM_ROWS ? (n_act - s0) : HM_ROWS);
    const float seg_scale = (seg.rows && seg.scale) ? seg.scale[0] : 1.f;
    // Staging in batches of HM_STAGE elements per thread so that the dependent loads of the fused segment reduce
    // (segment bounds -> member slot -> gradient row) are issued for all of a thread's elements at once: three
    // memory round trips per batch, not three per element. Rows have one member nearly always; further members are
    // added in ascending slot order by the (rare) loop, so the sums are the ones segment_sum_kernel forms.
    constexpr int HM_STAGE = 4;
    for (int e0 = tid * 4; e0 < HM_ROWS * Cy; e0 += 2048 * HM_STAGE) {
        int r[HM_STAGE], c[HM_STAGE], beg[HM_STAGE], end[HM_STAGE], mem[HM_STAGE];
        bool in[HM_STAGE];
        float4 v[HM_STAGE];
#pragma unroll
        for (int q = 0; q < HM_STAGE; ++q) {
            const int e = e0 + 2048 * q;
            r[q] = e / Cy; c[q] = e - r[q] * Cy;
            in[q] = e < HM_ROWS * Cy && r[q] < rows;
            beg[q] = 0; end[q] = 0;
            if (in[q] && seg.rows) { beg[q] = seg.seg_start[s0 + r[q]]; end[q] = seg.seg_start[s0 + r[q] + 1]; }
        }
        if (seg.rows) {
#pragma unroll
            for (int q = 0; q < HM_STAGE; ++q) mem[q] = (in[q] && beg[q] < end[q]) ? seg.members[beg[q]] : 0;
        }
#pragma unroll
        for (int q = 0; q < HM_STAGE; ++q) {
            v[q] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (in[q]) {
                if (!seg.rows) v[q] = ld4(dY + (s0 + r[q]) * lddy + c[q]);
                else if (beg[q] < end[q]) {
                    const float4 x = ld4(seg.rows + (int64_t)mem[q] * Cy + c[q]);
                    v[q].x += x.x; v[q].y += x.y; v[q].z += x.z; v[q].w += x.w;    // 0 + x, as the serial loop does
                }
            }
        }
#pragma unroll
        for (int q = 0; q < HM_STAGE; ++q) {
            const int e = e0 + 2048 * q;
            if (e >= HM_ROWS * Cy) continue;
            if (in[q] && seg.rows) {
                for (int i = beg[q] + 1; i < end[q]; ++i) {
                    const float4 x = ld4(seg.rows + (int64_t)seg.members[i] * Cy + c[q]);
                    v[q].x += x.x; v[q].y += x.y; v[q].z += x.z; v[q].w += x.w;
                }
                v[q] = make_float4(v[q].x * seg_scale, v[q].y * seg_scale, v[q].z * seg_scale, v[q].w * seg_scale);
                st4(seg.reduced + (s0 + r[q]) * lddy + c[q], v[q]);
            }
            float *dst = dys + r[q] * ldy + c[q];
            dst[0] = v[q].x; dst[1] = v[q].y; dst[2] = v[q].z; dst[3] = v[q].w;
        }
    }
    if (tid < HM_ROWS) node[tid] = (tid < rows) ? (int64_t)active_rows[s0 + tid] : -1;
    __syncthreads();
    const bool any_user = node[0] < U;
    const bool any_item = node[rows - 1] >= U;
    const bool mixed = any_user && any_item;
    const int li = lane & 31, lk = lane >> 5;
    const int n_tiles = C / 32;
    for (int t = wave; t < n_tiles; t += 8) {
        const int c0 = t * 32;
        const int mb = c0 / d;
        v16f_ acc = {0}, acc2 = {0};
        acc = hm_accumulate(acc, dys, ldy, (any_user ? W_user : W_item) + c0, C, d, li, lk);
        if (mixed) acc2 = hm_accumulate(acc2, dys, ldy, W_item + c0, C, d, li, lk);
        for (int h = 0; h < S; ++h) {
            if (hp.mblock[h] != mb) continue;
            const float *Wh = hp.w[h] + (c0 - mb * d);
            acc = hm_accumulate(acc, dys + (1 + h) * d, ldy, Wh, d, d, li, lk);
            if (mixed) acc2 = hm_accumulate(acc2, dys + (1 + h) * d, ldy, Wh, d, d, li, lk);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (row < rows) {
                const int64_t nd = node[row];
                const float v = ((mixed && nd >= U) ? acc2[r] : acc[r]) * gscale;
                if (G0 && c0 + li < scatter_cols) G0[nd * ldg + c0 + li] = v;
                if (compact) compact[(s0 + row) * C + c0 + li] = v;
            }
        }
    }
}

// The same kernel on 16-row tiles (taken when the caller hands over the packed operands, see the launch site): twice the workgroups
// at half the threads, a quarter of the LDS, v_mfma_f32_16x16x4_f32, four 16-column output tiles per wave.
typedef float v4f_ __attribute__((ext_vector_type(4)));
constexpr int HM16 = 16;

__device__ __forceinline__ v4f_ hm16_accumulate(v4f_ acc, const float *__restrict__ a_row, const float *__restrict__ Wp,
                                                int64_t ldw, int K, int kq) {
    // acc[row i][col j] += sum_k a_row[k] * Wp[k*ldw]   (a_row = this lane's LDS row, Wp = this lane's column)
    constexpr int PF = 16;
    for (int k0 = 0; k0 < K; k0 += 4 * PF) {
        float b[PF];
#pragma unroll
        for (int q = 0; q < PF; ++q) {
            const int k = k0 + 4 * q + kq;
            b[q] = (k < K) ? Wp[(int64_t)k * ldw] : 0.f;
        }
#pragma unroll
        for (int q = 0; q < PF; ++q) {
            const int k = k0 + 4 * q + kq;
            const float a = (k < K) ? a_row[k] : 0.f;
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b[q], acc, 0, 0, 0);
        }
    }
    return acc;
}

// the same with the B operand from the packed copy (head.hip, common.h head_pack_layout): bp = start of the column tile +
// lane, one coalesced 256-B load per MFMA; K = 64
__device__ __forceinline__ v4f_ hm16_accumulate_packed(v4f_ acc, const float *__restrict__ a_row, const float *__restrict__ bp, int kq) {
    float b[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) b[q] = bp[q * 64];
#pragma unroll
    for (int q = 0; q < 16; ++q) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a_row[4 * q + kq], b[q], acc, 0, 0, 0);
    return acc;
}

// slab-major adjoint source tables (slab.hip layout: [d / w][N][w]) the 16-row head backward can fill directly when one
// rank owns every column and every active row is listed once: what elimrec_slab_merge_rows(M >= 1) would write
// ... or, with peers (split != NULL): the [H | G] rows cut into the `world` column slices a column-sharded job sends them,
// split[q][slot] = [H[slot][q*dl : (q+1)*dl] | G[slot][q*dl : (q+1)*dl]] (layout [world x n_max x 2*dl]: what
// elimrec_source_rows_split makes of the compact rows)
struct SlabSources { float *A, *B; int64_t N; int w, w_shift; float *split; int64_t n_max; int dl; };

struct HeadPackPtrs { const float *f[2]; const float *s[kMaxHeads]; };      // null f[0]: weights read unpacked

// SEG: the segment reduce of the slot rows fused in front (the dY rows are formed here); !SEG: dY rows given (a reduce launch of its
// own ran before: large batches, where the launch is paced by how many tiles a CU holds and the staging registers cost a third
// of them)
// PACKED: the operands from the packed copy, recdim 64 (the caller checked): the instantiation carries no code for weights read
// unpacked
// (Measured in round 5, B = 32768 = 65 k active rows, the same bits each time: 32- and 64-row tiles of this kernel -- every operand
// load feeding 2 / 4 MFMAs on independent accumulators -- are SLOWER at every batch size (B = 2048 0.289 -> 0.302 / 0.326 ms per
// step, B = 32768 0.758 -> 0.765 / 0.779); the segment sums as a launch of their own (a wave per segment, sixteen members in
// flight) + this kernel with SEG = false: 73.6 + 73.0 us against 133 fused, step 0.743 -> 0.764 ms -- the sums' launch is the
// chain of the batch's most popular item (604 members), which the fused form hides under the other tiles.)
template <bool SEG, bool PACKED>
__global__ __launch_bounds__(256) void head_bwd_input16_kernel(const float *__restrict__ dY, int64_t lddy,
                                                               const int32_t *__restrict__ active_rows,
                                                               const int32_t *__restrict__ seg_info, int64_t n_max,
                                                               int64_t U, int d, int C, int S, HeadPtrs hp,
                                                               const float *__restrict__ W_user,
                                                               const float *__restrict__ W_item, float gscale,
                                                               float *__restrict__ G0, int64_t ldg, int scatter_cols,
                                                               float *__restrict__ compact, SegSrc seg, HeadPackPtrs pk,
                                                               SlabSources src) {
    extern __shared__ float dys[];                       // [16][Cy + 4]
    __shared__ int64_t node[HM16];
    const int Cy = (1 + S) * d, ldy = Cy + 4;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t s0 = (int64_t)blockIdx.x * HM16;
    int64_t n_act = seg_info[0];
    if (n_act > n_max) n_act = n_max;
    if (s0 >= n_act) return;
    const int rows = (int)((n_act - s0) < HM16 ? (n_act - s0) : HM16);
    const bool segd = SEG && seg.rows != nullptr;
    const float seg_scale = (segd && seg.scale) ? seg.scale[0] : 1.f;
    // staging as in the 32-row kernel: the dependent loads of the fused segment reduce in batches of 4 per thread
    constexpr int ST = 4;
    for (int e0 = tid * 4; e0 < HM16 * Cy; e0 += 1024 * ST) {
        int r[ST], c[ST], beg[ST], end[ST], mem[ST];
        bool in[ST];
        float4 v[ST];
#pragma unroll
        for (int q = 0; q < ST; ++q) {
            const int e = e0 + 1024 * q;
            r[q] = e / Cy; c[q] = e - r[q] * Cy;
            in[q] = e < HM16 * Cy && r[q] < rows;
            beg[q] = 0; end[q] = 0;
            if (in[q] && segd) { beg[q] = seg.seg_start[s0 + r[q]]; end[q] = seg.seg_start[s0 + r[q] + 1]; }
        }
        if (segd) {
#pragma unroll
            for (int q = 0; q < ST; ++q) mem[q] = (in[q] && beg[q] < end[q]) ? seg.members[beg[q]] : 0;
        }
#pragma unroll
        for (int q = 0; q < ST; ++q) {
            v[q] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (in[q]) {
                if (!segd) v[q] = ld4(dY + (s0 + r[q]) * lddy + c[q]);
                else if (beg[q] < end[q]) {
                    const float4 x = ld4(seg.rows + (int64_t)mem[q] * Cy + c[q]);
                    v[q].x += x.x; v[q].y += x.y; v[q].z += x.z; v[q].w += x.w;    // 0 + x, as the serial loop does
                }
            }
        }
#pragma unroll
        for (int q = 0; q < ST; ++q) {
            const int e = e0 + 1024 * q;
            if (e >= HM16 * Cy) continue;
            if (in[q] && segd) {
                // a popular item is listed by dozens of slots (48 of 6144 at the Tiktok shape), and walking them one dependent load
                // pair at a time was the launch's longest chain (26 us; 19.5 with sixteen, then four, members' rows in flight --
                // added in member order, as before)
                int i = beg[q] + 1;
                for (; i + 16 <= end[q]; i += 16) {
                    int mm[16];
                    float4 x[16];
#pragma unroll
                    for (int u = 0; u < 16; ++u) mm[u] = seg.members[i + u];
#pragma unroll
                    for (int u = 0; u < 16; ++u) x[u] = ld4(seg.rows + (int64_t)mm[u] * Cy + c[q]);
#pragma unroll
                    for (int u = 0; u < 16; ++u) { v[q].x += x[u].x; v[q].y += x[u].y; v[q].z += x[u].z; v[q].w += x[u].w; }
                }
                for (; i + 4 <= end[q]; i += 4) {
                    int mm[4];
                    float4 x[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) mm[u] = seg.members[i + u];
#pragma unroll
                    for (int u = 0; u < 4; ++u) x[u] = ld4(seg.rows + (int64_t)mm[u] * Cy + c[q]);
#pragma unroll
                    for (int u = 0; u < 4; ++u) { v[q].x += x[u].x; v[q].y += x[u].y; v[q].z += x[u].z; v[q].w += x[u].w; }
                }
                for (; i < end[q]; ++i) {
                    const float4 x = ld4(seg.rows + (int64_t)seg.members[i] * Cy + c[q]);
                    v[q].x += x.x; v[q].y += x.y; v[q].z += x.z; v[q].w += x.w;
                }
                v[q] = make_float4(v[q].x * seg_scale, v[q].y * seg_scale, v[q].z * seg_scale, v[q].w * seg_scale);
                st4(seg.reduced + (s0 + r[q]) * lddy + c[q], v[q]);
            }
            st4(dys + r[q] * ldy + c[q], v[q]);
        }
    }
    if (tid < HM16) node[tid] = (tid < rows) ? (int64_t)active_rows[s0 + tid] : -1;
    __syncthreads();
    const bool any_user = node[0] < U;
    const bool any_item = node[rows - 1] >= U;
    const bool mixed = any_user && any_item;
    const int li = lane & 15, kq = lane >> 4;
    const int n_tiles = C / 16;
    const float *a_row = dys + li * ldy;
    // adjoint sources (one rank, d = 64): H = the sum of a row's M column blocks in block order, G = block 0. A wave's
    // tiles t = wave, wave + 4, ... are then the same 16 columns of consecutive blocks: H accumulates in registers
    float hs[4] = {0.f, 0.f, 0.f, 0.f};
    const int last_block = C / d - 1;
    // the store part of a finished output tile (c0 = its first column, mb = its table block)
    auto emit = [&](int c0, int mb, const v4f_ &acc, const v4f_ &acc2) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 4 * kq + r;
            if (row < rows) {
                const int64_t nd = node[row];
                const float v = ((mixed && nd >= U) ? acc2[r] : acc[r]) * gscale;
                if (G0 && c0 + li < scatter_cols) G0[nd * ldg + c0 + li] = v;
                if (compact) compact[(s0 + row) * C + c0 + li] = v;
                if (src.A || src.split) {
                    const int cb = c0 - mb * d + li;                 // column within the block
                    const float h = mb == 0 ? v : hs[r] + v;
                    hs[r] = h;
                    if (src.split) {
                        const int q = cb / src.dl, j = cb - q * src.dl;
                        float *rp = src.split + ((int64_t)q * src.n_max + (s0 + row)) * 2 * src.dl;
                        if (mb == last_block) rp[j] = h;
                        if (mb == 0) rp[src.dl + j] = v;
                    } else {
                        const int64_t at = ((int64_t)(cb >> src.w_shift) * src.N + nd) * src.w + (cb & (src.w - 1));
                        if (mb == 0) (nd < U ? src.B : src.A)[at] = v;
                        if (mb == last_block) (nd < U ? src.A : src.B)[at] = h;
                    }
                }
            }
        }
    };
    if (PACKED && !mixed && C <= 256) {
        // Packed operands, one weight matrix for the whole tile (every tile but the one that straddles the user / item boundary):
        // wave w owns column tile w of every table block k (t = w + 4k), i.e. up to 8 rounds of 16 MFMAs -- the fusion operand of
        // block k, then the single-modal head that feeds block k -- each behind one 16-register operand load from L2. The loads of
        // round r + 1 are issued before the MFMAs of round r (the serial form exposed a round trip per round: the launch's chain
        // at 2 workgroups per CU). Same MFMAs in the same order on the same accumulators: the same bits.
        const int M = C / 64;
        const float *pf = pk.f[any_user ? 0 : 1] + (int64_t)wave * 16 * 64 + lane;      // + 4 kb tiles per block
        const float *ph[4];
        const float *ah[4];
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            ph[kb] = nullptr; ah[kb] = a_row;
            for (int h = 0; h < S; ++h)
                if (hp.mblock[h] == kb) { ph[kb] = pk.s[h] + (int64_t)wave * 16 * 64 + lane; ah[kb] = a_row + (1 + h) * d; }
        }
        float bc[16], bn[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) bc[q] = pf[q * 64];
        const v4f_ zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            if (kb < M) {
                // round 1: the fusion operand of block kb (bc holds it); meanwhile this block's head operand, or the next block's
                const float *nxt = ph[kb] ? ph[kb] : (kb + 1 < M ? pf + (int64_t)4 * (kb + 1) * 16 * 64 : nullptr);
                if (nxt) {
#pragma unroll
                    for (int q = 0; q < 16; ++q) bn[q] = nxt[q * 64];
                }
                v4f_ acc = zero;
#pragma unroll
                for (int q = 0; q < 16; ++q) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a_row[4 * q + kq], bc[q], acc, 0, 0, 0);
#pragma unroll
                for (int q = 0; q < 16; ++q) bc[q] = bn[q];
                if (ph[kb]) {
                    // round 2: the single-modal head that feeds block kb; meanwhile the next block's fusion operand
                    if (kb + 1 < M) {
                        const float *n2 = pf + (int64_t)4 * (kb + 1) * 16 * 64;
#pragma unroll
                        for (int q = 0; q < 16; ++q) bn[q] = n2[q * 64];
                    }
                    const float *ar = ah[kb];
#pragma unroll
                    for (int q = 0; q < 16; ++q) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ar[4 * q + kq], bc[q], acc, 0, 0, 0);
#pragma unroll
                    for (int q = 0; q < 16; ++q) bc[q] = bn[q];
                }
                emit((wave + 4 * kb) * 16, kb, acc, zero);
            }
        }
        return;
    }
    for (int t = wave; t < n_tiles; t += 4) {
        const int c0 = t * 16;
        const int mb = c0 / d;
        v4f_ acc = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};
        if (PACKED) {
            const int64_t ft = (int64_t)t * 16 * 64 + lane;                       // column tile t of the [C x 64] fusion operand
            acc = hm16_accumulate_packed(acc, a_row, pk.f[any_user ? 0 : 1] + ft, kq);
            if (mixed) acc2 = hm16_accumulate_packed(acc2, a_row, pk.f[1] + ft, kq);
            for (int h = 0; h < S; ++h) {
                if (hp.mblock[h] != mb) continue;
                const float *bp = pk.s[h] + (int64_t)((c0 - mb * d) / 16) * 16 * 64 + lane;
                acc = hm16_accumulate_packed(acc, a_row + (1 + h) * d, bp, kq);
                if (mixed) acc2 = hm16_accumulate_packed(acc2, a_row + (1 + h) * d, bp, kq);
            }
        } else if (!PACKED) {
            acc = hm16_accumulate(acc, a_row, (any_user ? W_user : W_item) + c0 + li, C, d, kq);
            if (mixed) acc2 = hm16_accumulate(acc2, a_row, W_item + c0 + li, C, d, kq);
            for (int h = 0; h < S; ++h) {
                if (hp.mblock[h] != mb) continue;
                const float *Wh = hp.w[h] + (c0 - mb * d) + li;
                acc = hm16_accumulate(acc, a_row + (1 + h) * d, Wh, d, d, kq);
                if (mixed) acc2 = hm16_accumulate(acc2, a_row + (1 + h) * d, Wh, d, d, kq);
            }
        }
        emit(c0, mb, acc, acc2);
    }
}

struct SegLayout {
    size_t keys_sorted, vals_in, vals_sorted, flag, segid, seg_start, long_list, sort_tmp, scan_tmp, total;
    size_t sort_bytes, scan_bytes;
};

static int seg_layout(int64_t n, SegLayout &L) {
    size_t sort_bytes = 0, scan_bytes = 0;
    hipError_t e = rocprim::radix_sort_pairs<rocprim::default_config, const int32_t *, int32_t *, const int32_t *, int32_t *>(
        nullptr, sort_bytes, nullptr, nullptr, nullptr, nullptr, (size_t)n, 0, 32, 0, false);
    if (e != hipSuccess) return check_hip(e, "radix_sort_pairs(size)");
    e = rocprim::inclusive_scan<rocprim::default_config, const int32_t *, int32_t *, rocprim::plus<int32_t>>(
        nullptr, scan_bytes, nullptr, nullptr, (size_t)n, rocprim::plus<int32_t>(), 0, false);
    if (e != hipSuccess) return check_hip(e, "inclusive_scan(size)");
    size_t scan2 = 0;
    e = rocprim::exclusive_scan<rocprim::default_config, const int32_t *, int32_t *, int32_t, rocprim::plus<int32_t>>(
        nullptr, scan2, nullptr, nullptr, (int32_t)0, (size_t)n + 1, rocprim::plus<int32_t>(), 0, false);
    if (e != hipSuccess) return check_hip(e, "exclusive_scan(size)");
    if (scan2 > scan_bytes) scan_bytes = scan2;          // (also covers the scan of <= n bitmap-word popcounts)
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
    const size_t nb = (size_t)n * sizeof(int32_t);
    L.keys_sorted = take(nb); L.vals_in = take(nb); L.vals_sorted = take(nb); L.flag = take(nb + sizeof(int32_t)); L.segid = take(nb);
    L.seg_start = take(nb + sizeof(int32_t));
    L.long_list = take(nb + sizeof(int32_t));
    L.sort_tmp = take(sort_bytes ? sort_bytes : 4); L.scan_tmp = take(scan_bytes ? scan_bytes : 4);
    L.sort_bytes = sort_bytes; L.scan_bytes = scan_bytes; L.total = off;
    return 0;
}


// One 16-lane group per (triplet slot): node id out, and optionally the leading `cols` columns of that row.
__global__ __launch_bounds__(256) void triplet_rows_kernel(const int64_t *__restrict__ users, const int64_t *__restrict__ pos,
                                                           const int64_t *__restrict__ neg, int64_t B, int64_t U,
                                                           int32_t *__restrict__ rows, const float *__restrict__ src,
                                                           int64_t lds, int c4, float *__restrict__ dst, int64_t ldd,
                                                           int64_t I, int32_t *__restrict__ err) {
    const int64_t slot = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
    const int sub = threadIdx.x & 15;
    if (slot >= 3 * B) return;
    const int64_t b = slot / 3;
    const int j = (int)(slot - 3 * b);
    int64_t idx = j == 0 ? users[b] : (j == 1 ? pos[b] : neg[b]);
    const int64_t lim = j == 0 ? U : I;
    if (I >= 0 && (idx < 0 || idx >= lim)) {     // the reference raises IndexError here; flag it and stay in bounds
        if (err && sub == 0) atomicOr(err, 1 << j);
        idx = 0;
    }
    const int64_t node = j == 0 ? idx : U + idx;
    if (sub == 0) rows[slot] = (int32_t)node;
    if (src)
        for (int c = sub; c < c4; c += 16)
            *reinterpret_cast<float4 *>(dst + slot * ldd + 4 * c) = *reinterpret_cast<const float4 *>(src + node * lds + 4 * c);
}

__global__ __launch_bounds__(256) void gather_rows_kernel(const float *__restrict__ src, int64_t lds,
                                                          const int32_t *__restrict__ rows, const int32_t *__restrict__ count,
                                                          int64_t n, int c4, float *__restrict__ dst, int64_t ldd) {
    const int64_t r = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
    const int sub = threadIdx.x & 15;
    const int64_t lim = count ? min((int64_t)*count, n) : n;
    if (r >= lim) return;
    const int64_t node = rows[r];
    for (int c = sub; c < c4; c += 16)
        *reinterpret_cast<float4 *>(dst + r * ldd + 4 * c) = *reinterpret_cast<const float4 *>(src + node * lds + 4 * c);
}

}  // namespace elimrec

extern "C" int elimrec_triplet_rows_checked(const int64_t *d_users, const int64_t *d_pos, const int64_t *d_neg, int64_t B,
                                            int64_t U, int64_t I, int32_t *d_rows, int32_t *d_err, void *stream) {
    ELIMREC_REQUIRE(d_users && d_pos && d_neg && d_rows && d_err, "triplet_rows_checked: null pointer");
    ELIMREC_REQUIRE(U >= 0 && I >= 0, "triplet_rows_checked: bad table sizes");
    if (B <= 0) return 0;
    hipLaunchKernelGGL(elimrec::triplet_rows_kernel, dim3((unsigned)((3 * B + 15) / 16)), dim3(256), 0, (hipStream_t)stream, d_users,
                       d_pos, d_neg, B, U, d_rows, (const float *)nullptr, (int64_t)0, 0, (float *)nullptr, (int64_t)0, I, d_err);
    ELIMREC_LAUNCH_CHECK("triplet_rows_checked");
    return 0;
}

extern "C" int elimrec_triplet_rows(const int64_t *d_users, const int64_t *d_pos, const int64_t *d_neg, int64_t B, int64_t U,
                                    int32_t *d_rows, const float *d_src, int64_t lds, int cols, float *d_dst, int64_t ldd,
                                    void *stream) {
    ELIMREC_REQUIRE(d_users && d_pos && d_neg && d_rows, "triplet_rows: null pointer");
    ELIMREC_REQUIRE(!d_src || (d_dst && cols > 0 && cols % 4 == 0 && lds % 4 == 0 && ldd % 4 == 0),
                    "triplet_rows: cols, lds, ldd must be multiples of 4");
    if (B <= 0) return 0;
    hipLaunchKernelGGL(elimrec::triplet_rows_kernel, dim3((unsigned)((3 * B + 15) / 16)), dim3(256), 0, (hipStream_t)stream, d_users,
                       d_pos, d_neg, B, U, d_rows, d_src, lds, cols / 4, d_dst, ldd, (int64_t)-1, (int32_t *)nullptr);
    ELIMREC_LAUNCH_CHECK("triplet_rows");
    return 0;
}

extern "C" int elimrec_gather_rows(const float *d_src, int64_t lds, const int32_t *d_rows, const int32_t *d_count, int64_t n,
                                   int cols, float *d_dst, int64_t ldd, void *stream) {
    ELIMREC_REQUIRE(d_src && d_rows && d_dst, "gather_rows: null pointer");
    ELIMREC_REQUIRE(cols > 0 && cols % 4 == 0 && lds % 4 == 0 && ldd % 4 == 0, "gather_rows: cols, lds, ldd must be multiples of 4");
    if (n <= 0) return 0;
    hipLaunchKernelGGL(elimrec::gather_rows_kernel, dim3((unsigned)((n + 15) / 16)), dim3(256), 0, (hipStream_t)stream, d_src, lds,
                       d_rows, d_count, n, cols / 4, d_dst, ldd);
    ELIMREC_LAUNCH_CHECK("gather_rows");
    return 0;
}


using namespace elimrec;

static int bpr_group_lanes(int d) {      // lanes per head block: pow2 >= d/4, at most the wave
    int lb = 1;
    while (lb < d / 4 && lb < 64) lb <<= 1;
    return lb;
}

extern "C" int elimrec_bpr_head(const float *d_Y, int64_t ldy, int64_t U, int64_t I, const int64_t *d_users,
                                const int64_t *d_pos, const int64_t *d_neg, int B, int d, int n_blocks,
                                const float *block_weights, float *d_loss_rows, float *d_grad_rows, int32_t *d_keys,
                                void *stream) {
    ELIMREC_REQUIRE(d_Y && d_users && d_pos && d_neg && d_loss_rows && block_weights, "bpr_head: null pointer");
    ELIMREC_REQUIRE(d > 0 && d % 4 == 0, "bpr_head: recdim must be a positive multiple of 4");
    ELIMREC_REQUIRE(n_blocks >= 1 && n_blocks <= kMaxBlocks, "bpr_head: 1..%d head blocks supported", kMaxBlocks);
    ELIMREC_REQUIRE(ldy % 4 == 0 && ldy >= (int64_t)n_blocks * d, "bpr_head: bad ldy");
    ELIMREC_REQUIRE(U + I < (int64_t)INT32_MAX, "bpr_head: node ids must fit int32");
    ELIMREC_REQUIRE(!d_grad_rows || d_keys, "bpr_head: keys required with grad_rows");
    if (B <= 0) return 0;
    BlockWeights bw;
    for (int k = 0; k < kMaxBlocks; ++k) bw.w[k] = k < n_blocks ? block_weights[k] : 0.f;
    hipLaunchKernelGGL(bpr_head_kernel, dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)stream, d_Y, ldy, U, d_users,
                       d_pos, d_neg, B, d, n_blocks, bw, 1.0f / (float)B, d_loss_rows, d_grad_rows, d_keys,
                       (const int32_t *)nullptr, bpr_group_lanes(d));
    ELIMREC_LAUNCH_CHECK("bpr_head");
    return 0;
}

extern "C" int elimrec_bpr_head_rows(const float *d_Y, int64_t ldy, const int32_t *d_slot_rows, int B, int d,
                                     int n_blocks, const float *block_weights, float *d_loss_rows, float *d_grad_rows,
                                     void *stream) {
    ELIMREC_REQUIRE(d_Y && d_slot_rows && d_loss_rows && block_weights, "bpr_head_rows: null pointer");
    ELIMREC_REQUIRE(d > 0 && d % 4 == 0, "bpr_head_rows: recdim must be a positive multiple of 4");
    ELIMREC_REQUIRE(n_blocks >= 1 && n_blocks <= kMaxBlocks, "bpr_head_rows: 1..%d head blocks supported", kMaxBlocks);
    ELIMREC_REQUIRE(ldy % 4 == 0 && ldy >= (int64_t)n_blocks * d, "bpr_head_rows: bad ldy");
    if (B <= 0) return 0;
    BlockWeights bw;
    for (int k = 0; k < kMaxBlocks; ++k) bw.w[k] = k < n_blocks ? block_weights[k] : 0.f;
    hipLaunchKernelGGL(bpr_head_kernel, dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)stream, d_Y, ldy, (int64_t)0,
                       (const int64_t *)nullptr, (const int64_t *)nullptr, (const int64_t *)nullptr, B, d, n_blocks, bw,
                       1.0f / (float)B, d_loss_rows, d_grad_rows, (int32_t *)nullptr, d_slot_rows, bpr_group_lanes(d));
    ELIMREC_LAUNCH_CHECK("bpr_head_rows");
    return 0;
}

extern "C" int elimrec_bpr_head_rows_sum(const float *d_Y, int64_t ldy, const int32_t *d_slot_rows, int B, int d,
                                         int n_blocks, const float *block_weights, float *d_loss_rows, float *d_grad_rows,
                                         float *d_loss, int32_t *d_ticket, void *stream) {
    ELIMREC_REQUIRE(d_Y && d_slot_rows && d_loss_rows && block_weights && d_loss && d_ticket, "bpr_head_rows_sum: null pointer");
    ELIMREC_REQUIRE(d > 0 && d % 4 == 0, "bpr_head_rows_sum: recdim must be a positive multiple of 4");
    ELIMREC_REQUIRE(n_blocks >= 1 && n_blocks <= kMaxBlocks, "bpr_head_rows_sum: 1..%d head blocks supported", kMaxBlocks);
    ELIMREC_REQUIRE(ldy % 4 == 0 && ldy >= (int64_t)n_blocks * d, "bpr_head_rows_sum: bad ldy");
    if (B <= 0) return check_hip(hipMemsetAsync(d_loss, 0, sizeof(float), (hipStream_t)stream), "memset(loss)");
    BlockWeights bw;
    for (int k = 0; k < kMaxBlocks; ++k) bw.w[k] = k < n_blocks ? block_weights[k] : 0.f;
    hipLaunchKernelGGL(bpr_head_sum_kernel, dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)stream, d_Y, ldy, B, d, n_blocks, bw,
                       1.0f / (float)B, d_loss_rows, d_grad_rows, d_slot_rows, bpr_group_lanes(d), d_loss, d_ticket,
                       (unsigned long long *)nullptr, 0, (uint32_t *)nullptr);
    ELIMREC_LAUNCH_CHECK("bpr_head_rows_sum");
    return 0;
}

// ... and the loss PUBLISHED to the host from that launch (main.py:102 of the reference reads `loss.cpu().item()` after every
// step: a read of the device tensor waits for the whole step -- adjoint hops, Adam -- and the host cannot enqueue step t + 1
// under step t). pub: elimrec_loss_pub_create's block; the host waits on the slot with elimrec_loss_pub_wait, not on the stream.
struct LossPub { unsigned long long *h_slots; unsigned long long *d_slots; uint32_t *d_counter; int n; uint32_t issued; };

extern "C" int elimrec_bpr_head_rows_sum_pub(const float *d_Y, int64_t ldy, const int32_t *d_slot_rows, int B, int d,
                                             int n_blocks, const float *block_weights, float *d_loss_rows, float *d_grad_rows,
                                             float *d_loss, int32_t *d_ticket, void *pub, void *stream) {
    ELIMREC_REQUIRE(d_Y && d_slot_rows && d_loss_rows && block_weights && d_loss && d_ticket && pub, "bpr_head_rows_sum_pub: null pointer");
    ELIMREC_REQUIRE(d > 0 && d % 4 == 0, "bpr_head_rows_sum_pub: recdim must be a positive multiple of 4");
    ELIMREC_REQUIRE(n_blocks >= 1 && n_blocks <= kMaxBlocks, "bpr_head_rows_sum_pub: 1..%d head blocks supported", kMaxBlocks);
    ELIMREC_REQUIRE(ldy % 4 == 0 && ldy >= (int64_t)n_blocks * d && B > 0, "bpr_head_rows_sum_pub: bad ldy / empty batch");
    LossPub *lp = (LossPub *)pub;
    BlockWeights bw;
    for (int k = 0; k < kMaxBlocks; ++k) bw.w[k] = k < n_blocks ? block_weights[k] : 0.f;
    hipLaunchKernelGGL(bpr_head_sum_kernel, dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)stream, d_Y, ldy, B, d, n_blocks, bw,
                       1.0f / (float)B, d_loss_rows, d_grad_rows, d_slot_rows, bpr_group_lanes(d), d_loss, d_ticket,
                       lp->d_slots, lp->n, lp->d_counter);
    ELIMREC_LAUNCH_CHECK("bpr_head_rows_sum_pub");
    lp->issued += 1;                       // (the sequence number the launch just enqueued will publish)
    return 0;
}

extern "C" int elimrec_loss_pub_create(int n_slots, void **out_pub) {
    ELIMREC_REQUIRE(out_pub && n_slots >= 2, "loss_pub_create: n_slots >= 2");
    LossPub *lp = new LossPub();
    lp->n = n_slots; lp->issued = 0;
    int rc = check_hip(hipHostMalloc((void **)&lp->h_slots, (size_t)n_slots * 8, hipHostMallocMapped | hipHostMallocCoherent), "hipHostMalloc(loss pub)");
    if (rc) { delete lp; return rc; }
    memset(lp->h_slots, 0, (size_t)n_slots * 8);
    rc = check_hip(hipHostGetDevicePointer((void **)&lp->d_slots, lp->h_slots, 0), "hipHostGetDevicePointer(loss pub)");
    if (!rc) rc = check_hip(hipMalloc((void **)&lp->d_counter, 256), "hipMalloc(loss pub counter)");
    if (!rc) rc = check_hip(hipMemset(lp->d_counter, 0, 256), "hipMemset(loss pub counter)");
    if (rc) { (void)hipHostFree(lp->h_slots); delete lp; return rc; }
    *out_pub = lp;
    return 0;
}

extern "C" int elimrec_loss_pub_destroy(void *pub) {
    if (!pub) return 0;
    LossPub *lp = (LossPub *)pub;
    (void)hipFree(lp->d_counter);
    (void)hipHostFree(lp->h_slots);
    delete lp;
    return 0;
}

// the sequence number of the most recently ENQUEUED publishing launch (what a caller notes right after issuing a step)
extern "C" uint32_t elimrec_loss_pub_issued(void *pub) { return pub ? ((LossPub *)pub)->issued : 0u; }

// Waits (spinning on the coherent host slot, no stream synchronisation) until launch `seq` has published; *value = its loss.
// Returns 0; ELIMREC_E_UNSUPPORTED when the slot already holds a LATER launch's value (the ring wrapped: read the device tensor
// instead); ELIMREC_E_WORKSPACE on time-out.
extern "C" int elimrec_loss_pub_wait(void *pub, uint32_t seq, double timeout_s, float *value) {
    ELIMREC_REQUIRE(pub && value && seq > 0, "loss_pub_wait: bad arguments");
    LossPub *lp = (LossPub *)pub;
    const volatile unsigned long long *slot = lp->h_slots + (seq % (uint32_t)lp->n);
    const auto t0 = std::chrono::steady_clock::now();
    for (uint64_t spin = 0;; ++spin) {
        const unsigned long long w = __atomic_load_n(slot, __ATOMIC_ACQUIRE);
        const uint32_t got = (uint32_t)(w >> 32);
        if (got == seq) { const uint32_t b = (uint32_t)w; memcpy(value, &b, 4); return 0; }
        if ((int32_t)(got - seq) > 0) { set_error("loss_pub_wait: slot of launch %u already holds launch %u", seq, got); return ELIMREC_E_UNSUPPORTED; }
        if ((spin & 1023) == 1023 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s) {
            set_error("loss_pub_wait: launch %u did not publish within %.1f s", seq, timeout_s);
            return ELIMREC_E_WORKSPACE;
        }
        __builtin_ia32_pause();
    }
}

extern "C" int elimrec_sum(const float *d_x, int64_t n, float *d_out, void *stream) {
    ELIMREC_REQUIRE(d_x && d_out && n >= 0, "sum: bad arguments");
    hipLaunchKernelGGL(sum_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, d_x, n, d_out);
    ELIMREC_LAUNCH_CHECK("sum");
    return 0;
}

// plan workspace = the sorted path's layout (it also holds the fast path's arrays: vals_sorted = member lists,
// seg_start, flag = counts / cursors, segid = rank-sort scratch)
extern "C" size_t elimrec_segment_plan_workspace(int64_t n) {
    SegLayout L;
    if (n <= 0 || seg_layout(n, L)) return 0;
    return L.total;
}
extern "C" size_t elimrec_segment_reduce_workspace(int64_t n) { return elimrec_segment_plan_workspace(n); }

static int plan_fast_max_keys() { return 1 << 30; }          // (the one-workgroup planner for every batch size)

static int segment_plan_impl(const int32_t *d_keys, int64_t n, int32_t split_key, int64_t key_space,
                             int32_t *d_active_rows, int32_t *d_seg_info, int32_t *d_slot_seg,
                             uint32_t *d_key_bitmap, void *d_workspace, size_t workspace_bytes, void *stream,
                             const PlanBatch &pb);

extern "C" int elimrec_segment_plan(const int32_t *d_keys, int64_t n, int32_t split_key, int64_t key_space,
                                    int32_t *d_active_rows, int32_t *d_seg_info, int32_t *d_slot_seg,
                                    uint32_t *d_key_bitmap, void *d_workspace, size_t workspace_bytes, void *stream) {
    PlanBatch pb = {};
    return segment_plan_impl(d_keys, n, split_key, key_space, d_active_rows, d_seg_info, d_slot_seg, d_key_bitmap, d_workspace,
                             workspace_bytes, stream, pb);
}

extern "C" int elimrec_batch_plan(const int64_t *d_users, const int64_t *d_pos, const int64_t *d_neg, int64_t B, int64_t U,
                                  int64_t I, int32_t *d_keys, int32_t *d_active_rows, int32_t *d_seg_info,
                                  int32_t *d_slot_seg, uint32_t *d_key_bitmap, int32_t pad_key, int32_t *d_err,
                                  void *d_workspace, size_t workspace_bytes, void *stream) {
    ELIMREC_REQUIRE(d_users && d_pos && d_neg && d_keys && d_err, "batch_plan: null pointer");
    ELIMREC_REQUIRE(B > 0 && U >= 0 && I >= 0 && U + I < INT32_MAX, "batch_plan: bad sizes");
    PlanBatch pb = {d_users, d_pos, d_neg, U, I, d_keys, d_err, pad_key, 1};
    return segment_plan_impl(d_keys, 3 * B, (int32_t)U, U + I, d_active_rows, d_seg_info, d_slot_seg, d_key_bitmap, d_workspace,
                             workspace_bytes, stream, pb);
}

static int segment_plan_impl(const int32_t *d_keys, int64_t n, int32_t split_key, int64_t key_space,
                             int32_t *d_active_rows, int32_t *d_seg_info, int32_t *d_slot_seg,
                             uint32_t *d_key_bitmap, void *d_workspace, size_t workspace_bytes, void *stream,
                             const PlanBatch &pb) {
    ELIMREC_REQUIRE(d_keys && d_active_rows && d_seg_info && d_slot_seg && d_workspace, "segment_plan: null pointer");
    ELIMREC_REQUIRE(n > 0 && n < INT32_MAX, "segment_plan: bad n");
    ELIMREC_REQUIRE(!d_key_bitmap || key_space > 0, "segment_plan: the key bitmap needs key_space");
    SegLayout L;
    int rc = seg_layout(n, L);
    if (rc) return rc;
    if (workspace_bytes < L.total) {
        set_error("segment_plan: workspace too small (%zu < %zu)", workspace_bytes, L.total);
        return ELIMREC_E_WORKSPACE;
    }
    char *ws = (char *)d_workspace;
    int32_t *ks = (int32_t *)(ws + L.keys_sorted), *vin = (int32_t *)(ws + L.vals_in);
    int32_t *vs = (int32_t *)(ws + L.vals_sorted), *flag = (int32_t *)(ws + L.flag);
    int32_t *segid = (int32_t *)(ws + L.segid), *seg_start = (int32_t *)(ws + L.seg_start);
    hipStream_t s = (hipStream_t)stream;
    const size_t plan_lds = ((size_t)2 * ((key_space + 31) / 32) + 16 + 1024) * sizeof(uint32_t) +
                            (size_t)(3 * PLAN_LDS_N + 1) * sizeof(int32_t);
    if (key_space > 0 && key_space <= plan_fast_max_keys() && n <= PLAN_LDS_N && plan_lds <= 160 * 1024) {
        static size_t lds_set = 0;
        if (plan_lds > 64 * 1024 && plan_lds > lds_set) {
            hipError_t ea = hipFuncSetAttribute((const void *)segment_plan_kernel,
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)plan_lds);
            if (ea != hipSuccess) return check_hip(ea, "segment_plan: LDS size");
            lds_set = plan_lds;
        }
        hipLaunchKernelGGL(segment_plan_kernel, dim3(1), dim3(PLAN_T), plan_lds, s, d_keys, (int)n, (int)key_space,
                           (int)split_key, d_active_rows, d_seg_info, d_slot_seg, seg_start, vs, d_key_bitmap, pb);
        ELIMREC_LAUNCH_CHECK("segment_plan");
        return 0;
    }
    const int64_t nw_words = (key_space + 31) / 32;
    static int sorted_only = -1;
    if (sorted_only < 0) { const char *e = getenv("ELIMREC_PLAN_SORTED"); sorted_only = (e && e[0] == '1') ? 1 : 0; }
    if (key_space > 0 && nw_words <= n && !sorted_only) {
        // the bitmap plan, device-wide (see plan_bits_kernel): bm = the caller's key bitmap or the (unused) vals_in array, pre in
        // keys_sorted, counters in flag, the rank sort's scratch in segid, member lists in vals_sorted -- where segment_apply and
        // the head backward look for them
        uint32_t *bm = d_key_bitmap ? d_key_bitmap : (uint32_t *)vin;
        int32_t *long_list = (int32_t *)(ws + L.long_list);
        const unsigned nb = (unsigned)((n + 255) / 256);
        hipError_t e = hipMemsetAsync(bm, 0, (size_t)nw_words * sizeof(uint32_t), s);
        if (e != hipSuccess) return check_hip(e, "memset(plan bitmap)");
        hipLaunchKernelGGL(plan_bits_kernel, dim3((unsigned)((n + 256) / 256)), dim3(256), 0, s, d_keys, n, bm, flag, segid, pb);
        ELIMREC_LAUNCH_CHECK("plan_bits");
        size_t cb1 = L.scan_bytes;
        e = rocprim::exclusive_scan(ws + L.scan_tmp, cb1, rocprim::make_transform_iterator((const uint32_t *)bm, PopcOp()), ks, (int32_t)0,
                                    (size_t)nw_words, rocprim::plus<int32_t>(), s, false);
        if (e != hipSuccess) return check_hip(e, "exclusive_scan(plan ranks)");
        hipLaunchKernelGGL(plan_slots_kernel, dim3((unsigned)((std::max<int64_t>(n, nw_words) + 255) / 256)), dim3(256), 0, s,
                           pb.users ? (const int32_t *)pb.keys_out : d_keys, n, (const uint32_t *)bm, (int)nw_words, (const int32_t *)ks,
                           (int)split_key, (int)key_space, d_active_rows, d_seg_info, long_list, d_slot_seg, flag, pb.pad, pb.pad_key);
        ELIMREC_LAUNCH_CHECK("plan_slots");
        size_t cb2 = L.scan_bytes;
        e = rocprim::exclusive_scan(ws + L.scan_tmp, cb2, (const int32_t *)flag, seg_start, (int32_t)0, (size_t)n + 1,
                                    rocprim::plus<int32_t>(), s, false);          // seg_start[n_act .. n] = n
        if (e != hipSuccess) return check_hip(e, "exclusive_scan(plan)");
        hipLaunchKernelGGL(plan_fill_kernel, dim3(nb), dim3(256), 0, s, (const int32_t *)d_slot_seg, n, (const int32_t *)seg_start, segid, vs);
        ELIMREC_LAUNCH_CHECK("plan_fill");
        hipLaunchKernelGGL(plan_sort_short_kernel, dim3(nb), dim3(256), 0, s, (const int32_t *)d_seg_info, (const int32_t *)seg_start, vs, long_list);
        ELIMREC_LAUNCH_CHECK("plan_sort_short");
        hipLaunchKernelGGL(plan_sort_long_kernel, dim3(2048), dim3(256), 0, s, (const int32_t *)long_list, (const int32_t *)seg_start, vs, ks);
        ELIMREC_LAUNCH_CHECK("plan_sort_long");
        if (n > PLAN_SORT_LDS) {
            hipLaunchKernelGGL(plan_copy_long_kernel, dim3(256), dim3(256), 0, s, (const int32_t *)long_list, (const int32_t *)seg_start, vs,
                               (const int32_t *)ks);
            ELIMREC_LAUNCH_CHECK("plan_copy_long");
        }
        return 0;
    }
    if (pb.users) {      // key lists beyond the one-workgroup planner: the keys by their own (range-checked) launch first
        int rc2 = elimrec_triplet_rows_checked(pb.users, pb.pos, pb.neg, n / 3, pb.U, pb.I, pb.keys_out, pb.err, stream);
        if (rc2) return rc2;
    }
    const unsigned nb = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(iota_kernel, dim3(nb), dim3(256), 0, s, vin, n);
    ELIMREC_LAUNCH_CHECK("iota");
    size_t sb = L.sort_bytes;
    hipError_t e = rocprim::radix_sort_pairs(ws + L.sort_tmp, sb, d_keys, ks, (const int32_t *)vin, vs, (size_t)n, 0,
                                             32, s, false);
    if (e != hipSuccess) return check_hip(e, "radix_sort_pairs");
    hipLaunchKernelGGL(heads_kernel, dim3(nb), dim3(256), 0, s, ks, n, flag);
    ELIMREC_LAUNCH_CHECK("heads");
    size_t cb = L.scan_bytes;
    e = rocprim::inclusive_scan(ws + L.scan_tmp, cb, (const int32_t *)flag, segid, (size_t)n,
                                rocprim::plus<int32_t>(), s, false);
    if (e != hipSuccess) return check_hip(e, "inclusive_scan");
    hipLaunchKernelGGL(finalize_segments_kernel, dim3(nb), dim3(256), 0, s, ks, flag, segid, n, split_key,
                       d_active_rows, seg_start, d_seg_info);
    ELIMREC_LAUNCH_CHECK("finalize_segments");
    hipLaunchKernelGGL(publish_ranges_kernel, dim3(1), dim3(64), 0, s, d_seg_info);
    ELIMREC_LAUNCH_CHECK("publish_ranges");
    hipLaunchKernelGGL(slot_segments_kernel, dim3(nb), dim3(256), 0, s, vs, segid, n, d_slot_seg);
    ELIMREC_LAUNCH_CHECK("slot_segments");
    if (d_key_bitmap) {
        e = hipMemsetAsync(d_key_bitmap, 0, (size_t)((key_space + 31) / 32) * sizeof(uint32_t), s);
        if (e != hipSuccess) return check_hip(e, "memset(key bitmap)");
        hipLaunchKernelGGL(key_bitmap_kernel, dim3(nb), dim3(256), 0, s, d_keys, n, d_key_bitmap);
        ELIMREC_LAUNCH_CHECK("key_bitmap");
    }
    if (pb.pad) {
        hipLaunchKernelGGL(pad_keys_kernel, dim3(nb), dim3(256), 0, s, d_active_rows, (const int32_t *)d_seg_info, n, pb.pad_key);
        ELIMREC_LAUNCH_CHECK("pad_keys");
    }
    return 0;
}

extern "C" int elimrec_segment_apply(const float *d_rows, int64_t n, int ld, const int32_t *d_seg_info,
                                     const float *d_scale, float *d_reduced, const void *d_workspace,
                                     size_t workspace_bytes, void *stream) {
    ELIMREC_REQUIRE(d_rows && d_seg_info && d_reduced && d_workspace, "segment_apply: null pointer");
    ELIMREC_REQUIRE(n > 0 && n < INT32_MAX && ld > 0 && ld % 4 == 0, "segment_apply: bad n/ld");
    SegLayout L;
    int rc = seg_layout(n, L);
    if (rc) return rc;
    if (workspace_bytes < L.total) {
        set_error("segment_apply: workspace too small (%zu < %zu)", workspace_bytes, L.total);
        return ELIMREC_E_WORKSPACE;
    }
    const char *ws = (const char *)d_workspace;
    const int32_t *vs = (const int32_t *)(ws + L.vals_sorted), *seg_start = (const int32_t *)(ws + L.seg_start);
    hipLaunchKernelGGL(segment_sum_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, d_rows, vs,
                       seg_start, d_seg_info, n, ld / 4, d_scale, d_reduced);
    ELIMREC_LAUNCH_CHECK("segment_sum");
    return 0;
}

extern "C" int elimrec_segment_reduce_rows(const float *d_rows, const int32_t *d_keys, int64_t n, int ld,
                                           int32_t split_key, int32_t *d_active_rows, float *d_reduced,
                                           const float *d_scale, int32_t *d_seg_info, void *d_workspace,
                                           size_t workspace_bytes, void *stream) {
    ELIMREC_REQUIRE(d_rows && d_keys && d_active_rows && d_reduced && d_seg_info && d_workspace,
                    "segment_reduce_rows: null pointer");
    ELIMREC_REQUIRE(n > 0 && n < INT32_MAX && ld > 0 && ld % 4 == 0, "segment_reduce_rows: bad n/ld");
    SegLayout L;
    int rc = seg_layout(n, L);
    if (rc) return rc;
    // key space unknown here: the sorted plan; its slot->segment map goes to the (then unused) vals_in array
    rc = elimrec_segment_plan(d_keys, n, split_key, 0, d_active_rows, d_seg_info,
                              (int32_t *)((char *)d_workspace + L.vals_in), nullptr, d_workspace, workspace_bytes, stream);
    if (rc) return rc;
    return elimrec_segment_apply(d_rows, n, ld, d_seg_info, d_scale, d_reduced, d_workspace, workspace_bytes, stream);
}

extern "C" int elimrec_head_bwd_input(const float *d_dY, int64_t lddy, const int32_t *d_active_rows,
                                      const int32_t *d_seg_info, int64_t n_max, int64_t U, int d, int C, int S,
                                      const int *head_mblock, const float *d_W_user, const float *d_W_item,
                                      const float *const *d_W_heads, float gscale, float *d_G0, int64_t ldg,
                                      int scatter_cols, float *d_compact, void *stream) {
    ELIMREC_REQUIRE(d_dY && d_active_rows && d_seg_info && d_W_user && d_W_item && (d_G0 || d_compact),
                    "head_bwd_input: null pointer");
    ELIMREC_REQUIRE(!d_G0 || (ldg >= scatter_cols && scatter_cols >= 0 && scatter_cols <= C), "head_bwd_input: bad ldg/scatter_cols");
    ELIMREC_REQUIRE(S >= 0 && S <= kMaxHeads, "head_bwd_input: at most %d heads", kMaxHeads);
    ELIMREC_REQUIRE(d > 0 && d % 4 == 0 && C % d == 0, "head_bwd_input: bad d/C");
    HeadPtrs hp;
    for (int h = 0; h < kMaxHeads; ++h) {
        hp.w[h] = h < S ? d_W_heads[h] : nullptr;
        hp.mblock[h] = h < S ? head_mblock[h] : -1;
    }
    if (n_max <= 0) return 0;
    if (d % 32 == 0 && (size_t)HM_ROWS * ((1 + S) * d + 1) * sizeof(float) <= 96 * 1024) {
        const size_t lds_m = (size_t)HM_ROWS * ((1 + S) * d + 1) * sizeof(float);
        ELIMREC_REQUIRE(lddy % 4 == 0, "head_bwd_input: lddy must be a multiple of 4");
        hipLaunchKernelGGL(head_bwd_input_mfma_kernel, dim3((unsigned)((n_max + HM_ROWS - 1) / HM_ROWS)), dim3(512),
                           lds_m, (hipStream_t)stream, d_dY, lddy, d_active_rows, d_seg_info, n_max, U, d, C, S, hp,
                           d_W_user, d_W_item, gscale, d_G0, ldg, scatter_cols, d_compact, SegSrc{});
        ELIMREC_LAUNCH_CHECK("head_bwd_input_mfma");
        return 0;
    }
    const size_t lds = (size_t)HB_ROWS * (1 + S) * d * sizeof(float);
    ELIMREC_REQUIRE(lds <= 128 * 1024, "head_bwd_input: (1+S)*d too large for the LDS tile");
    ELIMREC_REQUIRE(lddy % 4 == 0, "head_bwd_input: lddy must be a multiple of 4");
    hipLaunchKernelGGL(head_bwd_input_kernel, dim3((unsigned)((n_max + HB_ROWS - 1) / HB_ROWS)), dim3(256), lds,
                       (hipStream_t)stream, d_dY, lddy, d_active_rows, d_seg_info, n_max, U, d, C, S, hp, d_W_user,
                       d_W_item, gscale, d_G0, ldg, scatter_cols, d_compact);
    ELIMREC_LAUNCH_CHECK("head_bwd_input");
    return 0;
}

// segment_apply + head_bwd_input in one launch (the MFMA form stages the dY rows in LDS anyway: it sums them
// from the member gradient rows instead of reading them back). Falls back to the two launches when the MFMA
// form does not apply.
static int segment_apply_head_bwd_impl(const float *d_rows, int64_t n, int ld, const int32_t *d_active_rows,
                                       const int32_t *d_seg_info, const float *d_scale, float *d_reduced,
                                       const void *d_plan_workspace, size_t plan_workspace_bytes, int64_t U, int d,
                                       int C, int S, const int *head_mblock, const float *d_W_user,
                                       const float *d_W_item, const float *const *d_W_heads, float *d_compact,
                                       const float *d_pack_bwd, const SlabSources *src, void *stream);

extern "C" int elimrec_segment_apply_head_bwd(const float *d_rows, int64_t n, int ld, const int32_t *d_active_rows,
                                              const int32_t *d_seg_info, const float *d_scale, float *d_reduced,
                                              const void *d_plan_workspace, size_t plan_workspace_bytes, int64_t U, int d,
                                              int C, int S, const int *head_mblock, const float *d_W_user,
                                              const float *d_W_item, const float *const *d_W_heads, float *d_compact,
                                              void *stream) {
    return segment_apply_head_bwd_impl(d_rows, n, ld, d_active_rows, d_seg_info, d_scale, d_reduced, d_plan_workspace,
                                       plan_workspace_bytes, U, d, C, S, head_mblock, d_W_user, d_W_item, d_W_heads, d_compact,
                                       nullptr, nullptr, stream);
}

extern "C" int elimrec_segment_apply_head_bwd_packed(const float *d_rows, int64_t n, int ld, const int32_t *d_active_rows,
                                                     const int32_t *d_seg_info, const float *d_scale, float *d_reduced,
                                                     const void *d_plan_workspace, size_t plan_workspace_bytes, int64_t U,
                                                     int d, int C, int S, const int *head_mblock, const float *d_W_user,
                                                     const float *d_W_item, const float *const *d_W_heads,
                                                     float *d_compact, const float *d_pack_bwd, void *stream) {
    ELIMREC_REQUIRE(d_pack_bwd, "segment_apply_head_bwd_packed: null pack pointer");
    return segment_apply_head_bwd_impl(d_rows, n, ld, d_active_rows, d_seg_info, d_scale, d_reduced, d_plan_workspace,
                                       plan_workspace_bytes, U, d, C, S, head_mblock, d_W_user, d_W_item, d_W_heads, d_compact,
                                       d_pack_bwd, nullptr, stream);
}

extern "C" int elimrec_segment_apply_head_bwd_sources(const float *d_rows, int64_t n, int ld, const int32_t *d_active_rows,
                                                      const int32_t *d_seg_info, const float *d_scale, float *d_reduced,
                                                      const void *d_plan_workspace, size_t plan_workspace_bytes, int64_t U,
                                                      int d, int C, int S, const int *head_mblock, const float *d_W_user,
                                                      const float *d_W_item, const float *const *d_W_heads,
                                                      float *d_compact, const float *d_pack_bwd, int64_t N, int ns, int w,
                                                      float *d_SrcA, float *d_SrcB, void *stream) {
    ELIMREC_REQUIRE(d_pack_bwd && d_SrcA && d_SrcB, "segment_apply_head_bwd_sources: null pointer");
    ELIMREC_REQUIRE(d == 64 && w >= 4 && (w & (w - 1)) == 0 && ns * w == d && N >= U,
                    "segment_apply_head_bwd_sources: recdim 64 in [ns x N x w] slabs (d=%d, ns=%d, w=%d)", d, ns, w);
    SlabSources src = {d_SrcA, d_SrcB, N, w, 0, nullptr, 0, 0};
    while ((1 << src.w_shift) < w) ++src.w_shift;
    return segment_apply_head_bwd_impl(d_rows, n, ld, d_active_rows, d_seg_info, d_scale, d_reduced, d_plan_workspace,
                                       plan_workspace_bytes, U, d, C, S, head_mblock, d_W_user, d_W_item, d_W_heads, d_compact,
                                       d_pack_bwd, &src, stream);
}

extern "C" int elimrec_segment_apply_head_bwd_split(const float *d_rows, int64_t n, int ld, const int32_t *d_active_rows,
                                                    const int32_t *d_seg_info, const float *d_scale, float *d_reduced,
                                                    const void *d_plan_workspace, size_t plan_workspace_bytes, int64_t U,
                                                    int d, int C, int S, const int *head_mblock, const float *d_W_user,
                                                    const float *d_W_item, const float *const *d_W_heads,
                                                    float *d_compact, const float *d_pack_bwd, int64_t n_max, int world,
                                                    float *d_out, void *stream) {
    ELIMREC_REQUIRE(d_pack_bwd && d_out, "segment_apply_head_bwd_split: null pointer");
    ELIMREC_REQUIRE(d == 64 && world >= 1 && d % world == 0 && (d / world) % 4 == 0 && n_max >= n,
                    "segment_apply_head_bwd_split: recdim 64 in %d column slices, n_max >= n", world);
    SlabSources src = {nullptr, nullptr, 0, 0, 0, d_out, n_max, d / world};
    return segment_apply_head_bwd_impl(d_rows, n, ld, d_active_rows, d_seg_info, d_scale, d_reduced, d_plan_workspace,
                                       plan_workspace_bytes, U, d, C, S, head_mblock, d_W_user, d_W_item, d_W_heads, d_compact,
                                       d_pack_bwd, &src, stream);
}

static int segment_apply_head_bwd_impl(const float *d_rows, int64_t n, int ld, const int32_t *d_active_rows,
                                       const int32_t *d_seg_info, const float *d_scale, float *d_reduced,
                                       const void *d_plan_workspace, size_t plan_workspace_bytes, int64_t U, int d,
                                       int C, int S, const int *head_mblock, const float *d_W_user,
                                       const float *d_W_item, const float *const *d_W_heads, float *d_compact,
                                       const float *d_pack_bwd, const SlabSources *src, void *stream) {
    ELIMREC_REQUIRE(d_rows && d_active_rows && d_seg_info && d_reduced && d_plan_workspace && d_W_user && d_W_item &&
                    d_compact, "segment_apply_head_bwd: null pointer");
    ELIMREC_REQUIRE(n > 0 && n < INT32_MAX && ld > 0 && ld % 4 == 0, "segment_apply_head_bwd: bad n/ld");
    ELIMREC_REQUIRE(S >= 0 && S <= kMaxHeads && d > 0 && d % 4 == 0 && C % d == 0 && ld == (1 + S) * d,
                    "segment_apply_head_bwd: bad d/C/S/ld");
    const size_t lds_m = (size_t)HM_ROWS * ((1 + S) * d + 1) * sizeof(float);
    if (!(d % 32 == 0 && lds_m <= 96 * 1024)) {
        int rc = elimrec_segment_apply(d_rows, n, ld, d_seg_info, d_scale, d_reduced, d_plan_workspace,
                                       plan_workspace_bytes, stream);
        if (rc) return rc;
        return elimrec_head_bwd_input(d_reduced, ld, d_active_rows, d_seg_info, n, U, d, C, S, head_mblock, d_W_user,
                                      d_W_item, d_W_heads, 1.0f, nullptr, 0, 0, d_compact, stream);
    }
    SegLayout L;
    int rc = seg_layout(n, L);
    if (rc) return rc;
    if (plan_workspace_bytes < L.total) {
        set_error("segment_apply_head_bwd: plan workspace too small");
        return ELIMREC_E_WORKSPACE;
    }
    const char *ws = (const char *)d_plan_workspace;
    SegSrc seg = {d_rows, (const int32_t *)(ws + L.vals_sorted), (const int32_t *)(ws + L.seg_start), d_scale, d_reduced};
    HeadPtrs hp;
    for (int h = 0; h < kMaxHeads; ++h) {
        hp.w[h] = h < S ? d_W_heads[h] : nullptr;
        hp.mblock[h] = h < S ? head_mblock[h] : -1;
    }
    // 16-row tiles: with the weights read unpacked from L2 the smaller MFMA doubles the operand loads (32.4 us against
    // 28.5 for the 32-row kernel at the Tiktok shape) -- used when the caller hands over the packed operands
    // (recdim 64)
    const bool packed = d_pack_bwd && d == 64;
    if (packed && C % 16 == 0 && (1 + S) * d % 4 == 0) {
        const size_t lds16 = (size_t)HM16 * ((1 + S) * d + 4) * sizeof(float);
        HeadPackPtrs pk = {};
        if (packed) {
            pk.f[0] = d_pack_bwd; pk.f[1] = d_pack_bwd + (int64_t)C * 64;
            for (int h = 0; h < S && h < kMaxHeads; ++h) pk.s[h] = d_pack_bwd + (int64_t)2 * C * 64 + (int64_t)h * 64 * 64;
        }
        SlabSources ss = src ? *src : SlabSources{};
        if (packed)
            hipLaunchKernelGGL((head_bwd_input16_kernel<true, true>), dim3((unsigned)((n + HM16 - 1) / HM16)), dim3(256), lds16,
                               (hipStream_t)stream, (const float *)d_reduced, (int64_t)ld, d_active_rows, d_seg_info, n, U, d, C, S, hp,
                               d_W_user, d_W_item, 1.0f, (float *)nullptr, (int64_t)0, 0, d_compact, seg, pk, ss);
        else
            hipLaunchKernelGGL((head_bwd_input16_kernel<true, false>), dim3((unsigned)((n + HM16 - 1) / HM16)), dim3(256), lds16,
                               (hipStream_t)stream, (const float *)d_reduced, (int64_t)ld, d_active_rows, d_seg_info, n, U, d, C, S, hp,
                               d_W_user, d_W_item, 1.0f, (float *)nullptr, (int64_t)0, 0, d_compact, seg, pk, ss);
        ELIMREC_LAUNCH_CHECK("segment_apply_head_bwd16");
        return 0;
    }
    ELIMREC_REQUIRE(!src, "segment_apply_head_bwd_sources: needs the packed operands of the 16-row head (recdim 64)");
    hipLaunchKernelGGL(head_bwd_input_mfma_kernel, dim3((unsigned)((n + HM_ROWS - 1) / HM_ROWS)), dim3(512), lds_m,
                       (hipStream_t)stream, (const float *)d_reduced, (int64_t)ld, d_active_rows, d_seg_info, n, U, d, C, S, hp,
                       d_W_user, d_W_item, 1.0f, (float *)nullptr, (int64_t)0, 0, d_compact, seg);
    ELIMREC_LAUNCH_CHECK("segment_apply_head_bwd");
    return 0;
}

