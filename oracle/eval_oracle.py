"""CPU oracle for the evaluator loop and the pairwise sampler -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this. It wraps
oracle/_build/liboracle_port.so (our C restatement, oracle/eval_port.c) and, when present,
oracle/_ref/libref_eval.so (the reference's own headers compiled in place), and restates the
Python loops around them. Paths cited are relative to /root/reference.
"""
import collections
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
METRIC_IDS = {"Precision": 1, "Recall": 2, "MAP": 3, "NDCG": 4, "MRR": 5}  # cpp/uni_evaluator.py:14


def build(force=False):
    """Compile the checker libraries (make -C oracle)."""
    port = os.path.join(_HERE, "_build", "liboracle_port.so")
    if force or not os.path.exists(port) or os.path.getmtime(port) < os.path.getmtime(os.path.join(_HERE, "eval_port.c")):
        subprocess.check_call(["make", "-C", _HERE, "_build/liboracle_port.so"], stdout=subprocess.DEVNULL)
    if os.path.isdir("/root/reference/evaluator/backend/cpp/include"):
        ref = os.path.join(_HERE, "_ref", "libref_eval.so")
        if force or not os.path.exists(ref):
            subprocess.check_call(["make", "-C", _HERE, "ref"], stdout=subprocess.DEVNULL)


_port = None
_ref = None


def port_lib():
    global _port
    if _port is None:
        build()
        _port = ctypes.CDLL(os.path.join(_HERE, "_build", "liboracle_port.so"))
        _port.oracle_llrand.restype = ctypes.c_uint64
    return _port


def ref_lib():
    """The reference's own C++ (or None if oracle/_ref was not built)."""
    global _ref
    if _ref is None:
        p = os.path.join(_HERE, "_ref", "libref_eval.so")
        if not os.path.exists(p):
            return None
        _ref = ctypes.CDLL(p)
    return _ref


def _ptr(a, t):
    return a.ctypes.data_as(ctypes.POINTER(t))


def truth_to_csr(truth_lists):
    ptr = np.cumsum([0] + [len(t) for t in truth_lists]).astype(np.int64)
    items = np.asarray([i for t in truth_lists for i in t], dtype=np.int32)
    return ptr, items


def evaluate_matrix(scores, truth_ptr, truth_items, metric_ids, top_k, use_ref=False, thread_num=2):
    """cpp_evaluator.pyx:28-42 -> evaluate.h:45-64. Returns (results [U, n_metrics*K], topk [U,K] or None)."""
    scores = np.ascontiguousarray(scores, dtype=np.float32)
    n_users, n_items = scores.shape
    truth_ptr = np.ascontiguousarray(truth_ptr, dtype=np.int64)
    truth_items = np.ascontiguousarray(truth_items, dtype=np.int32)
    mids = np.ascontiguousarray(metric_ids, dtype=np.int32)
    res = np.zeros((n_users, len(mids) * top_k), dtype=np.float32)
    if use_ref:
        lib = ref_lib()
        lib.ref_evaluate_matrix(_ptr(scores, ctypes.c_float), n_users, n_items, _ptr(truth_ptr, ctypes.c_int64),
                                _ptr(truth_items, ctypes.c_int), _ptr(mids, ctypes.c_int), len(mids), top_k,
                                thread_num, _ptr(res, ctypes.c_float))
        topk = np.zeros((n_users, top_k), dtype=np.int32)
        lib.ref_arg_topk_2d(_ptr(scores, ctypes.c_float), n_items, n_users, top_k, thread_num, _ptr(topk, ctypes.c_int))
        return res, topk
    topk = np.zeros((n_users, top_k), dtype=np.int32)
    port_lib().oracle_evaluate_matrix(_ptr(scores, ctypes.c_float), n_users, n_items, _ptr(truth_ptr, ctypes.c_int64),
                                      _ptr(truth_items, ctypes.c_int), _ptr(mids, ctypes.c_int), len(mids), top_k,
                                      _ptr(res, ctypes.c_float), _ptr(topk, ctypes.c_int))
    return res, topk


def metrics_from_rank(rank, truth_ptr, truth_items, metric_ids, top_k):
    rank = np.ascontiguousarray(rank, dtype=np.int32)
    n_users = rank.shape[0]
    truth_ptr = np.ascontiguousarray(truth_ptr, dtype=np.int64)
    truth_items = np.ascontiguousarray(truth_items, dtype=np.int32)
    mids = np.ascontiguousarray(metric_ids, dtype=np.int32)
    res = np.zeros((n_users, len(mids) * top_k), dtype=np.float32)
    port_lib().oracle_metrics_from_rank(_ptr(rank, ctypes.c_int), n_users, top_k, _ptr(truth_ptr, ctypes.c_int64),
                                        _ptr(truth_items, ctypes.c_int), _ptr(mids, ctypes.c_int), len(mids),
                                        _ptr(res, ctypes.c_float))
    return res


def uni_evaluate(predict_fn, user_train_dict, user_test_dict, metrics=("Precision", "Recall", "NDCG"),
                 top_k=(10,), batch_size=128, use_ref=False):
    """evaluator/backend/cpp/uni_evaluator.py:104-203 (user_neg_test=None branch)."""
    metric_ids = [METRIC_IDS[m] for m in metrics]
    max_top = top_k if isinstance(top_k, int) else max(top_k)
    top_show = np.arange(max_top) + 1 if isinstance(top_k, int) else np.sort(top_k)
    test_users = list(user_test_dict.keys())
    batch_result = []
    for s in range(0, len(test_users), batch_size):
        batch_users = test_users[s:s + batch_size]
        test_items = [user_test_dict[u] for u in batch_users]
        score = np.array(predict_fn(batch_users), dtype=np.float32)
        for idx, user in enumerate(batch_users):                                  # :149-154
            score[idx][user_train_dict.get(user, [])] = -np.inf
        tp, ti = truth_to_csr([sorted(set(t)) for t in test_items])
        res, _ = evaluate_matrix(score, tp, ti, metric_ids, max_top, use_ref=use_ref)
        batch_result.append(res)
    all_res = np.concatenate(batch_result, axis=0)
    final = np.mean(all_res, axis=0).reshape(len(metric_ids), max_top)[:, top_show - 1].reshape(-1)
    buf = "\t".join([("%.8f" % x).ljust(12) for x in final])
    return final, buf


# --------------------------------------------------------------------------- sampler
def randint_choice(high, size=1, exclusion=None):
    """util/cython/random_choice.pyx:20-62 (replace=True, p=None). libc rand() stream."""
    if size <= 0:
        raise ValueError("'size' must be a positive integer.")
    if exclusion is not None and high <= len(exclusion):
        raise ValueError("The number of 'exclusion' is greater than 'high'.")
    excl = np.ascontiguousarray(sorted(set(int(e) for e in exclusion)) if exclusion is not None else [], dtype=np.int32)
    out = np.zeros(size, dtype=np.int32)
    port_lib().oracle_randint_choice(int(high), int(size), _ptr(excl, ctypes.c_int), len(excl), _ptr(out, ctypes.c_int))
    return int(out[0]) if size == 1 else out.tolist()


def pairwise_sampling_v2(user_pos_dict, num_samples, num_item):
    """data/sampler.py:93-126."""
    user_arr = np.array(list(user_pos_dict.keys()), dtype=np.int32)
    user_idx = randint_choice(len(user_arr), size=num_samples)
    users_list = user_arr[user_idx]
    user_pos_len = collections.OrderedDict()
    for u in users_list:
        user_pos_len[u] = user_pos_len.get(u, 0) + 1
    pos_s, neg_s = {}, {}
    for user, n in user_pos_len.items():
        pos_items = np.asarray(user_pos_dict[user])
        idx = randint_choice(len(pos_items), size=n)
        idx = idx if isinstance(idx, list) else [idx]
        pos_s[user] = list(pos_items[idx])
        neg = randint_choice(num_item, size=n, exclusion=user_pos_dict[user])
        neg_s[user] = neg if isinstance(neg, list) else [neg]
    pos_list = [pos_s[u].pop() for u in users_list]
    neg_list = [neg_s[u].pop() for u in users_list]
    return users_list, pos_list, neg_list


def pairwise_sampler_v2_epoch(user_pos_dict, num_items, batch_size, shuffle=True):
    """data/sampler.py:336-344 + util/data_iterator.py:44-58,145-152 (np.random.permutation shuffle)."""
    num_trainings = sum(len(v) for v in user_pos_dict.values())
    u, p, n = pairwise_sampling_v2(user_pos_dict, num_trainings, num_items)
    order = np.random.permutation(len(u)).tolist() if shuffle else list(range(len(u)))
    for s in range(0, len(order), batch_size):
        idx = order[s:s + batch_size]
        yield [int(u[i]) for i in idx], [int(p[i]) for i in idx], [int(n[i]) for i in idx]
