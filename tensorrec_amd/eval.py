"""
Evaluation metrics with the reference's names and results (tensorrec/eval.py:7-192), computed from the ranks of the
POSITIVE test interactions only -- the reference multiplies the dense [n_users, n_items] rank matrix by the dense
interaction mask (``sparse.A``, removed in current SciPy); here the same numbers come from one gather per positive pair,
so the functions also accept a ``PairRanks`` (ranks of the positive pairs only, e.g. from the item-shardable
rank-count kernel) instead of the [U, I] matrix.

    precision@k = hits / k        recall@k = hits / n_positives(user)        f1 = 2 mean(p) mean(r) / (mean(p) + mean(r))
    ndcg@k      = sum_{hits} (2^rel - 1) / log2(rank + 1)  /  sum_{j < min(n_pos, 10)} (2^rel_(j) - 1) / log2(j + 2)
"""
import numpy as np
import scipy.sparse as sp


class PairRanks(object):
    """Ranks of the positive test interactions, in CSR (row-major) order of ``test_interactions > 0``."""

    def __init__(self, rows, ranks, values, n_users):
        self.rows, self.ranks, self.values, self.n_users = rows, ranks, values, n_users


def _pairs(predicted_ranks, test_interactions):
    if isinstance(predicted_ranks, PairRanks):
        return predicted_ranks
    m = sp.csr_matrix(test_interactions)
    m.sort_indices()
    coo = m.tocoo()
    pos = coo.data > 0
    rows, cols, vals = coo.row[pos], coo.col[pos], coo.data[pos]
    ranks = np.asarray(predicted_ranks)[rows, cols]
    return PairRanks(rows, ranks, vals, m.shape[0])


def _per_user(rows, weights, n_users):
    return np.bincount(rows, weights=weights, minlength=n_users)


def _filter(x, pr, preserve_rows):
    if preserve_rows:
        return x
    return x[_per_user(pr.rows, None, pr.n_users) > 0]


def precision_at_k(predicted_ranks, test_interactions, k=10, preserve_rows=False):
    """(eval.py:7-30)"""
    pr = _pairs(predicted_ranks, test_interactions)
    hits = _per_user(pr.rows, (pr.ranks < k + 1).astype(float), pr.n_users)
    return _filter(hits / k, pr, preserve_rows)


def recall_at_k(predicted_ranks, test_interactions, k=10, preserve_rows=False):
    """(eval.py:33-59); users without positives give nan when preserve_rows is True, as 0/0 does in the reference"""
    pr = _pairs(predicted_ranks, test_interactions)
    hits = _per_user(pr.rows, (pr.ranks < k + 1).astype(float), pr.n_users)
    retrieved = _per_user(pr.rows, None, pr.n_users).astype(float)
    with np.errstate(invalid="ignore", divide="ignore"):
        return _filter(hits / retrieved, pr, preserve_rows)


def _idcg(hits, k=10):
    """(eval.py:76-79)"""
    hits = np.asarray(hits, dtype=float)
    sorted_hits = hits[np.argsort(-hits)][:min(len(hits), k)]
    return np.sum((2 ** sorted_hits - 1) / np.log2(np.arange(len(sorted_hits)) + 2))


def ndcg_at_k(predicted_ranks, test_interactions, k=10, preserve_rows=False):
    """(eval.py:91-117).  The reference evaluates the ideal DCG through ``np.apply_along_axis(_idcg, 1, relevance.A)``
    without forwarding ``k`` (eval.py:109), so the normaliser is always IDCG@10; reproduced."""
    pr = _pairs(predicted_ranks, test_interactions)
    hit = pr.ranks < k + 1
    gain = np.where(hit, (2.0 ** pr.values - 1.0) / np.log2(np.where(hit, pr.ranks, 1) + 1.0), 0.0)
    dcg = _per_user(pr.rows, gain, pr.n_users)
    idcg = np.zeros(pr.n_users)
    order = np.argsort(pr.rows, kind="stable")
    bounds = np.searchsorted(pr.rows[order], np.arange(pr.n_users + 1))
    for u in range(pr.n_users):
        if bounds[u + 1] > bounds[u]:
            idcg[u] = _idcg(pr.values[order[bounds[u]:bounds[u + 1]]])
    with np.errstate(invalid="ignore", divide="ignore"):
        return _filter(dcg / idcg, pr, preserve_rows)


def f1_score_at_k(predicted_ranks, test_interactions, k=10, preserve_rows=False):
    """(eval.py:120-148)"""
    mean_p = np.mean(precision_at_k(predicted_ranks, test_interactions, k=k, preserve_rows=preserve_rows))
    mean_r = np.mean(recall_at_k(predicted_ranks, test_interactions, k=k, preserve_rows=preserve_rows))
    return (2.0 * mean_p * mean_r) / (mean_p + mean_r)


def fit_and_eval(model, user_features, item_features, train_interactions, test_interactions, fit_kwargs, recall_k=30,
                 precision_k=5, ndcg_k=30):
    """(eval.py:151-167)"""
    model.fit(user_features=user_features, item_features=item_features, interactions=train_interactions, **fit_kwargs)
    predicted_ranks = model.predict_rank(user_features=user_features, item_features=item_features)
    out = []
    for inter in (test_interactions, train_interactions):
        out += [np.mean(recall_at_k(predicted_ranks, inter, k=recall_k)),
                np.mean(precision_at_k(predicted_ranks, inter, k=precision_k)),
                np.mean(ndcg_at_k(predicted_ranks, inter, k=ndcg_k))]
    return tuple(out)


def eval_random_ranks_on_dataset(interactions, recall_k=30, precision_k=5, ndcg_k=30):
    """(eval.py:182-192)"""
    n_users, n_items = interactions.shape
    random_guesses = np.array([np.random.choice(a=n_items, size=n_items, replace=False) + 1 for _ in range(n_users)])
    return (np.mean(recall_at_k(random_guesses, interactions, k=recall_k)),
            np.mean(precision_at_k(random_guesses, interactions, k=precision_k)),
            np.mean(ndcg_at_k(random_guesses, interactions, k=ndcg_k)))
