"""
oracle/eval_dense.py -- TEST INFRASTRUCTURE.  Literal restatement of tensorrec/eval.py:7-117 in its dense form
(rank matrix * dense interaction mask), with ``.A`` replaced by ``.toarray()`` (removed in current SciPy).  Pinned by
the reference's known answers in test/test_eval.py:84-148 (tests/test_eval_metrics.py)."""
import numpy as np
import scipy.sparse as sp


def precision_at_k(predicted_ranks, test_interactions, k=10, preserve_rows=False):
    positive = sp.csr_matrix(test_interactions) > 0
    ranks = sp.csr_matrix(np.asarray(predicted_ranks) * positive.toarray())
    ranks.data = np.less(ranks.data, (k + 1), ranks.data)
    precision = np.squeeze(np.array(ranks.sum(axis=1))).astype(float) / k
    if not preserve_rows:
        precision = precision[positive.getnnz(axis=1) > 0]
    return precision


def recall_at_k(predicted_ranks, test_interactions, k=10, preserve_rows=False):
    positive = sp.csr_matrix(test_interactions) > 0
    ranks = sp.csr_matrix(np.asarray(predicted_ranks) * positive.toarray())
    ranks.data = np.less(ranks.data, (k + 1), ranks.data)
    retrieved = np.squeeze(positive.getnnz(axis=1))
    hit = np.squeeze(np.array(ranks.sum(axis=1)))
    if not preserve_rows:
        hit = hit[positive.getnnz(axis=1) > 0]
        retrieved = retrieved[positive.getnnz(axis=1) > 0]
    return hit.astype(float) / retrieved.astype(float)


def setup_ndcg(predicted_ranks, test_interactions, k=10):
    ti = sp.csr_matrix(test_interactions)
    pos = ti > 0
    ror = sp.csr_matrix(np.asarray(predicted_ranks) * pos.toarray())
    relevance = sp.csr_matrix(ti.toarray() * pos.toarray())
    k_mask = np.less(ror.data, k + 1)
    ror_at_k = np.maximum(np.multiply(ror.data, k_mask), 1)
    return relevance, k_mask, ror, ror_at_k


def idcg(hits, k=10):
    sorted_hits = hits[np.argsort(-hits)][:min(len(hits), k)]
    return np.sum((2 ** sorted_hits - 1) / np.log2(np.arange(len(sorted_hits)) + 2))


def dcg(relevance, k_mask, ror_at_k, ror):
    numer = (2 ** np.multiply(relevance.data, k_mask)) - 1
    denom = np.log2(ror_at_k + 1)
    ror = ror.copy().astype(float)
    ror.data = numer / denom
    return np.asarray(ror.sum(axis=1)).flatten()


def ndcg_at_k(predicted_ranks, test_interactions, k=10, preserve_rows=False):
    relevance, k_mask, ror, ror_at_k = setup_ndcg(predicted_ranks, test_interactions, k)
    d = dcg(relevance, k_mask, ror_at_k, ror)
    i = np.apply_along_axis(idcg, 1, relevance.toarray())
    with np.errstate(invalid="ignore", divide="ignore"):
        ndcg = d / i
    if not preserve_rows:
        ndcg = ndcg[(sp.csr_matrix(test_interactions) > 0).getnnz(axis=1) > 0]
    return ndcg
