"""Synthetic interaction data with structure (benchmarks and diagnostics; the counterpart of the reference's
``util.generate_dummy_data``, tensorrec/util.py:61-85, for shapes where a uniform random matrix teaches a model nothing).

``planted_cluster_interactions``: users and items belong to planted taste clusters, items have a Zipf popularity; a
user draws most of its interactions from its own cluster (popularity-weighted) and the rest from the whole catalogue.
A model fitted on it has what trained recommenders have and random weights do not: clustered directions, row norms and
biases that follow popularity."""
import numpy as np
import scipy.sparse as sp
import torch


def planted_cluster_interactions(n_users, n_items, n_clusters=256, per_user=20, own_cluster_share=0.8, zipf_exponent=0.9,
                                 seed=0, holdout=0.0, device=None):
    """Returns (train CSR [n_users, n_items] of ones, held-out CSR or None, user_cluster [n_users], item_cluster [n_items]).
    One inverse-CDF lookup per interaction, done with torch on ``device`` (default: the GPU when there is one -- 20M
    lookups in a 1M-entry table take ~15 s of cache misses on one host core); the draws depend on the device's generator."""
    if device is None:
        device = torch.device("cuda") if torch.cuda.is_available() else torch.device("cpu")
    device = torch.device(device)
    g = torch.Generator(device=device)
    g.manual_seed(int(seed))
    user_cluster = torch.randint(0, n_clusters, (n_users,), device=device, generator=g)
    item_cluster = torch.randint(0, n_clusters, (n_items,), device=device, generator=g)
    popularity = 1.0 / torch.arange(1, n_items + 1, device=device, dtype=torch.float64) ** zipf_exponent
    popularity = popularity[torch.randperm(n_items, device=device, generator=g)]   # popular items spread over the id range
    # items grouped by cluster; cumulative popularity inside the concatenation
    order = torch.argsort(item_cluster, stable=True)
    cdf = torch.cumsum(popularity[order], 0)
    sorted_cl = item_cluster[order]
    cl = torch.arange(n_clusters, device=device)
    starts = torch.searchsorted(sorted_cl, cl, right=False)
    ends = torch.searchsorted(sorted_cl, cl, right=True)
    zero = torch.zeros((), dtype=torch.float64, device=device)
    lo = torch.where(starts > 0, cdf[(starts - 1).clamp(min=0)], zero)
    hi = torch.where(ends > 0, cdf[(ends - 1).clamp(min=0)], zero)
    c = user_cluster.repeat_interleave(per_user)
    n = n_users * per_user
    own = torch.rand((n,), device=device, generator=g) < own_cluster_share
    r = torch.rand((n,), device=device, generator=g, dtype=torch.float64)
    own &= hi[c] > lo[c]                                                # a cluster without items: draw globally
    target = torch.where(own, lo[c] + r * (hi[c] - lo[c]), r * cdf[-1])
    pos = torch.searchsorted(cdf, target, right=True).clamp(max=n_items - 1)
    items = order[pos].to(torch.int32).reshape(n_users, per_user)
    items, _ = torch.sort(items, dim=1)                                 # CSR built directly: rows are already grouped
    keep = torch.ones((n_users, per_user), dtype=torch.bool, device=device)
    keep[:, 1:] = items[:, 1:] != items[:, :-1]                         # duplicates of a (user, item) pair collapse to one
    if holdout > 0.0:
        held_mask = keep & (torch.rand((n_users, per_user), device=device, generator=g) < holdout)
    else:
        held_mask = torch.zeros_like(keep)

    def csr(mask):
        indptr = np.zeros(n_users + 1, dtype=np.int64)
        np.cumsum(mask.sum(dim=1).cpu().numpy(), out=indptr[1:])
        cols = items[mask].cpu().numpy()
        return sp.csr_matrix((np.ones(cols.shape[0], np.float32), cols, indptr), shape=(n_users, n_items))

    train = csr(keep & ~held_mask)
    held = csr(held_mask) if holdout > 0.0 else None
    return train, held, user_cluster.cpu().numpy().astype(np.int32), item_cluster.cpu().numpy().astype(np.int32)
