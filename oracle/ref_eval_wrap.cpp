// Builds oracle/_ref/libref_eval.so: the REFERENCE's own top-K + metric code, compiled from
// the headers where they lie under /root/reference (never copied into this repo):
//   evaluator/backend/cpp/include/evaluate.h   (cpp_evaluate_matrix, eval_one_user)
//   evaluator/backend/cpp/include/metric.h     (precision/recall/ap/ndcg/mrr)
//   util/cython/include/arg_topk.h             (arg_top_k_2d)
//   util/cython/include/thread_pool.h
// This file only adds a C ABI in front of them (the reference reaches them through Cython:
// evaluator/backend/cpp/cpp_evaluator.pyx:28-42, util/cython/arg_topk.pyx:16-35).
// TEST INFRASTRUCTURE: used to validate oracle/eval_port.c and the device evaluator.
#include "evaluate.h"
#include "arg_topk.h"
#include <cstdint>

extern "C" {

void ref_evaluate_matrix(float *scores, int n_users, int n_items, const int64_t *truth_ptr,
                         const int *truth_items, const int *metric_ids, int n_metrics, int top_k,
                         int thread_num, float *results) {
    std::vector<std::unordered_set<int>> test_items(n_users);
    for (int u = 0; u < n_users; ++u)
        for (int64_t j = truth_ptr[u]; j < truth_ptr[u + 1]; ++j) test_items[u].insert(truth_items[j]);
    std::vector<int> metric(metric_ids, metric_ids + n_metrics);
    cpp_evaluate_matrix(scores, n_items, test_items, metric, top_k, thread_num, results);
}

void ref_arg_topk_2d(float *scores, int n_items, int n_rows, int top_k, int thread_num, int *out) {
    arg_top_k_2d(scores, n_items, n_rows, top_k, thread_num, out);
}

}  // extern "C"
