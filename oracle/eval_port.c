/* CPU oracle (C restatement) for the evaluator + sampler helpers of the EliMRec hot path.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT CODE: only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library. The product path never links or dlopens it.
 *
 * Pinned by tests/test_oracle_golden.py against tests/golden/metrics.npz and sampler.npz
 * (captured from the reference's own Cython/C++ build) and against oracle/_ref/libref_eval.so
 * (the reference's evaluate.h / arg_topk.h compiled where they lie) when that is present.
 *
 * Paths cited are relative to /root/reference.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ---- top-K: std::partial_sort_copy over indices with comp(a,b) = score[a] > score[b] -------
 * evaluator/backend/cpp/include/evaluate.h:23-34 and util/cython/include/arg_topk.h:15-25.
 * Tie order matters, so libstdc++'s algorithm is restated step by
 * step: __partial_sort_copy / __make_heap / __adjust_heap / __push_heap / __sort_heap
 * (bits/stl_algo.h, bits/stl_heap.h of GCC 11). */
static inline int comp(const float *s, int a, int b) { return s[a] > s[b]; }

static void push_heap_(const float *s, int *first, long hole, long top, int value) {
    long parent = (hole - 1) / 2;
    while (hole > top && comp(s, first[parent], value)) {
        first[hole] = first[parent];
        hole = parent;
        parent = (hole - 1) / 2;
    }
    first[hole] = value;
}

static void adjust_heap_(const float *s, int *first, long hole, long len, int value) {
    const long top = hole;
    long child = hole;
    while (child < (len - 1) / 2) {
        child = 2 * (child + 1);
        if (comp(s, first[child], first[child - 1])) child--;
        first[hole] = first[child];
        hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
        child = 2 * (child + 1);
        first[hole] = first[child - 1];
        hole = child - 1;
    }
    push_heap_(s, first, hole, top, value);
}

static void make_heap_(const float *s, int *first, long len) {
    if (len < 2) return;
    long parent = (len - 2) / 2;
    for (;;) {
        int value = first[parent];
        adjust_heap_(s, first, parent, len, value);
        if (parent == 0) return;
        parent--;
    }
}

void oracle_arg_topk_1d(const float *scores, int n, int top_k, int *result) {
    long kept = 0;
    int i = 0;
    if (top_k <= 0) return;
    while (i < n && kept < top_k) result[kept++] = i++;
    make_heap_(scores, result, kept);
    for (; i < n; ++i)
        if (comp(scores, i, result[0])) adjust_heap_(scores, result, 0, kept, i);
    /* __sort_heap */
    long last = kept;
    while (last > 1) {
        --last;
        int value = result[last];
        result[last] = result[0];
        adjust_heap_(scores, result, 0, last, value);
    }
}

/* ---- metrics: evaluator/backend/cpp/include/metric.h:17-106 -------------------------------
 * truth is a set of distinct item ids, given as a sorted-or-not int array without duplicates
 * removed here (callers pass unique ids; size is the number passed, as unordered_set::size()
 * of a duplicate-free list). */
static int in_truth(const int *truth, int nt, int x) {
    for (int i = 0; i < nt; ++i) if (truth[i] == x) return 1;
    return 0;
}

static void m_precision(const int *rank, int k, const int *t, int nt, float *out) { /* metric.h:17-28 */
    int hits = 0;
    for (int i = 0; i < k; ++i) {
        if (in_truth(t, nt, rank[i])) hits += 1;
        out[i] = (float)(1.0 * hits / (i + 1));
    }
}
static void m_recall(const int *rank, int k, const int *t, int nt, float *out) {    /* metric.h:31-43 */
    int hits = 0;
    for (int i = 0; i < k; ++i) {
        if (in_truth(t, nt, rank[i])) hits += 1;
        out[i] = (float)(1.0 * hits / (size_t)nt);
    }
}
static void m_ap(const int *rank, int k, const int *t, int nt, float *out) {        /* metric.h:46-62 */
    int hits = 0;
    float pre = 0, sum_pre = 0;
    for (int i = 0; i < k; ++i) {
        if (in_truth(t, nt, rank[i])) {
            hits += 1;
            pre = (float)(1.0 * hits / (i + 1));
            sum_pre += pre;
        }
        out[i] = (hits == 0) ? 0.0f : sum_pre / hits;
    }
}
static void m_ndcg(const int *rank, int k, const int *t, int nt, float *out) {      /* metric.h:66-83 */
    float idcg = 0, dcg = 0;
    for (unsigned int i = 0; i < (unsigned int)k; ++i) {
        if (in_truth(t, nt, rank[i])) dcg = (float)(dcg + 1.0 / log2((double)(i + 2)));
        if (i < (size_t)nt) idcg = (float)(idcg + 1.0 / log2((double)(i + 2)));
        out[i] = dcg / idcg;
    }
}
static void m_mrr(const int *rank, int k, const int *t, int nt, float *out) {       /* metric.h:86-106 */
    for (int i = 0; i < k; ++i) {
        if (in_truth(t, nt, rank[i])) {
            float rr = (float)(1.0 / (i + 1));
            for (int j = i; j < k; ++j) out[j] = rr;
            break;
        } else {
            out[i] = 0.0f;
        }
    }
}

/* evaluate.h:23-64: one row of scores per user -> metrics laid out [user][metric][k]. */
void oracle_evaluate_matrix(const float *scores, int n_users, int n_items,
                            const int64_t *truth_ptr, const int *truth_items,
                            const int *metric_ids, int n_metrics, int top_k,
                            float *results, int *topk_out /* nullable [n_users*top_k] */) {
    int *rank = (int *)malloc(sizeof(int) * (size_t)top_k);
    for (int u = 0; u < n_users; ++u) {
        oracle_arg_topk_1d(scores + (size_t)u * n_items, n_items, top_k, rank);
        if (topk_out) memcpy(topk_out + (size_t)u * top_k, rank, sizeof(int) * (size_t)top_k);
        const int *t = truth_items + truth_ptr[u];
        int nt = (int)(truth_ptr[u + 1] - truth_ptr[u]);
        for (int m = 0; m < n_metrics; ++m) {
            float *out = results + ((size_t)u * n_metrics + m) * top_k;
            switch (metric_ids[m]) {                                                /* metric.h:108-114 */
                case 1: m_precision(rank, top_k, t, nt, out); break;
                case 2: m_recall(rank, top_k, t, nt, out); break;
                case 3: m_ap(rank, top_k, t, nt, out); break;
                case 4: m_ndcg(rank, top_k, t, nt, out); break;
                case 5: m_mrr(rank, top_k, t, nt, out); break;
                default: break;
            }
        }
    }
    free(rank);
}

/* Metrics from an already ranked top-K list (used to check the device top-K + metric kernels). */
void oracle_metrics_from_rank(const int *rank, int n_users, int top_k,
                              const int64_t *truth_ptr, const int *truth_items,
                              const int *metric_ids, int n_metrics, float *results) {
    for (int u = 0; u < n_users; ++u) {
        const int *r = rank + (size_t)u * top_k;
        const int *t = truth_items + truth_ptr[u];
        int nt = (int)(truth_ptr[u + 1] - truth_ptr[u]);
        for (int m = 0; m < n_metrics; ++m) {
            float *out = results + ((size_t)u * n_metrics + m) * top_k;
            switch (metric_ids[m]) {
                case 1: m_precision(r, top_k, t, nt, out); break;
                case 2: m_recall(r, top_k, t, nt, out); break;
                case 3: m_ap(r, top_k, t, nt, out); break;
                case 4: m_ndcg(r, top_k, t, nt, out); break;
                case 5: m_mrr(r, top_k, t, nt, out); break;
                default: break;
            }
        }
    }
}

/* ---- sampler RNG: util/cython/random_choice.pyx:12-62 ---------------------------------------
 * llrand(): five 15-bit libc rand() draws shifted into an unsigned 64-bit word (high bits fall
 * off); randint_choice(): a = llrand() % high, rejected while a is in the exclusion set.
 * libc rand() is never seeded by the reference (srand imported, never called). */
uint64_t oracle_llrand(void) {
    uint64_t r = 0;
    for (int i = 0; i < 5; ++i) r = (r << 15) | (uint64_t)(rand() & 0x7FFF);
    return r;
}

void oracle_srand(unsigned seed) { srand(seed); }

/* exclusion: sorted ascending int array (membership only matters). replace=True only. */
void oracle_randint_choice(int high, int size, const int *exclusion, int n_excl, int *out) {
    int i = 0;
    while (size - i) {
        int a = (int)(oracle_llrand() % (uint64_t)high);
        int lo = 0, hi = n_excl, found = 0;
        while (lo < hi) {
            int mid = (lo + hi) / 2;
            if (exclusion[mid] == a) { found = 1; break; }
            if (exclusion[mid] < a) lo = mid + 1; else hi = mid;
        }
        if (!found) out[i++] = a;
    }
}
