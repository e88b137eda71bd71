"""tensorrec_amd/synth.py (CPU): the planted-cluster Zipf interaction generator behind bench.py's trained_weights_mode."""
import numpy as np

from tensorrec_amd.synth import planted_cluster_interactions


def test_planted_cluster_interactions_shape_and_structure():
    n_u, n_i, n_c, per = 4000, 6000, 16, 12
    train, held, ucl, icl = planted_cluster_interactions(n_u, n_i, n_c, per, seed=3, holdout=0.1, device="cpu")
    assert train.shape == held.shape == (n_u, n_i) and ucl.shape == (n_u,) and icl.shape == (n_i,)
    assert train.has_sorted_indices and np.all(train.data == 1.0) and np.all(held.data == 1.0)
    both = train + held
    assert both.data.max() == 1.0                                    # a pair is either trained on or held out, never both
    per_user = np.diff(both.indptr)
    assert per_user.max() <= per and per_user.mean() > 0.75 * per    # duplicates of a (user, item) draw collapse (small catalogue: many)
    assert 0.05 < held.nnz / float(both.nnz) < 0.15
    coo = both.tocoo()
    own = (ucl[coo.row] == icl[coo.col]).mean()
    assert 0.7 < own < 0.9                                           # ~0.8 of the draws come from the user's own cluster
    counts = np.bincount(coo.col, minlength=n_i)
    assert counts.max() > 30 * max(1.0, np.median(counts))           # Zipf popularity: a heavy head


def test_planted_cluster_interactions_deterministic_per_seed():
    a = planted_cluster_interactions(500, 700, 8, 5, seed=1, device="cpu")
    b = planted_cluster_interactions(500, 700, 8, 5, seed=1, device="cpu")
    c = planted_cluster_interactions(500, 700, 8, 5, seed=2, device="cpu")
    assert (a[0] != b[0]).nnz == 0 and a[1] is None and np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3])
    assert (a[0] != c[0]).nnz > 0
