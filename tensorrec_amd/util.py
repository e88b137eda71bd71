"""
Host-side helpers with the reference's names and semantics (tensorrec/util.py:12-117).
"""
import math
import random

import numpy as np
import scipy.sparse as sp


def sample_items(n_items, n_users, n_sampled_items, replace, rng=None):
    """The reference's host sampler (util.py:12-21): one ``choice`` per user, (user, item) pairs emitted user-major
    as an int64 [n_users * n_sampled_items, 2] array.  Kept for HostSampler / replay parity runs; the default
    training path samples on the device instead (csrc/sampler.hip)."""
    rng = np.random if rng is None else rng
    items_per_user = [rng.choice(a=n_items, size=n_sampled_items, replace=replace) for _ in range(n_users)]
    users = np.repeat(np.arange(n_users, dtype=np.int64), n_sampled_items)
    items = np.asarray(items_per_user, dtype=np.int64).reshape(-1)
    return np.stack([users, items], axis=1)


def calculate_batched_alpha(num_batches, alpha):
    """(util.py:24-31)"""
    if num_batches < 1:
        raise ValueError('num_batches must be >=1, num_batches={}'.format(num_batches))
    elif num_batches > 1:
        batched_alpha = alpha / (math.e * math.log(num_batches))
    else:
        batched_alpha = alpha
    return batched_alpha


def generate_dummy_data(num_users=15000, num_items=30000, interaction_density=.00045, num_user_features=200,
                        num_item_features=200, n_features_per_user=20, n_features_per_item=20, pos_int_ratio=.5,
                        return_datasets=False, random_state=None):
    """Random interactions and features with the reference's shapes and densities (util.py:61-85).
    ``random_state`` is an extension (the reference is unseeded); ``return_datasets`` needs tf.data and is refused."""
    if pos_int_ratio <= 0.0:
        raise Exception("pos_int_ratio must be > 0")
    if return_datasets:
        raise ValueError("return_datasets=True builds tf.data.Datasets, which do not exist in this engine")
    rs = np.random.RandomState(random_state) if random_state is not None else None
    interactions = sp.rand(num_users, num_items, density=interaction_density * pos_int_ratio, random_state=rs)
    if pos_int_ratio < 1.0:
        interactions += -1 * sp.rand(num_users, num_items, density=interaction_density * (1 - pos_int_ratio),
                                     random_state=rs)
    user_features = sp.rand(num_users, num_user_features, density=float(n_features_per_user) / num_user_features,
                            random_state=rs)
    item_features = sp.rand(num_items, num_item_features, density=float(n_features_per_item) / num_item_features,
                            random_state=rs)
    return interactions, user_features, item_features


def generate_dummy_data_with_indicator(num_users=15000, num_items=30000, interaction_density=.00045, pos_int_ratio=.5,
                                       seed=None):
    """Indicator features plus random tag columns (util.py:88-117).  ``seed`` is an extension."""
    rnd = random.Random(seed) if seed is not None else random
    n_user_features = int(num_users * 1.2)
    n_user_tags = num_users * 3
    n_item_features = int(num_items * 1.2)
    n_item_tags = num_items * 3
    n_interactions = (num_users * num_items) * interaction_density

    user_features = sp.lil_matrix((num_users, n_user_features))
    for i in range(num_users):
        user_features[i, i] = 1
    for i in range(n_user_tags):
        user_features[rnd.randrange(num_users), rnd.randrange(num_users, n_user_features)] = 1

    item_features = sp.lil_matrix((num_items, n_item_features))
    for i in range(num_items):
        item_features[i, i] = 1
    for i in range(n_item_tags):
        item_features[rnd.randrange(num_items), rnd.randrange(num_items, n_item_features)] = 1

    interactions = sp.lil_matrix((num_users, num_items))
    for i in range(int(n_interactions * pos_int_ratio)):
        interactions[rnd.randrange(num_users), rnd.randrange(num_items)] = 1
    for i in range(int(n_interactions * (1 - pos_int_ratio))):
        interactions[rnd.randrange(num_users), rnd.randrange(num_items)] = -1

    return interactions, user_features, item_features
