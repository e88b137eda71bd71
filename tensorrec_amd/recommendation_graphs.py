"""
Free functions of tensorrec/recommendation_graphs.py:4-137 on device tensors.  The fused engine in tensorrec.py uses
kernel epilogues for the bias / rank steps; these functions are the composable forms with identical semantics
(tests replay the reference's known-answer vectors through them on the GPU).
"""
import torch

from . import ops
from .framework import Variable, zeros


def project_biases(tf_features, n_features, name='feature_biases'):
    """Projects per-feature biases to per-actor biases (recommendation_graphs.py:4-19)."""
    tf_feature_biases = Variable(lambda: zeros([n_features, 1]), name=name)
    tf_projected_biases = ops.sparse_matvec(tf_features, tf_feature_biases)
    return tf_feature_biases, tf_projected_biases


def split_sparse_tensor_indices(tf_sparse_tensor, n_dimensions=2):
    """(recommendation_graphs.py:22-30)"""
    return (tf_sparse_tensor.x_user, tf_sparse_tensor.x_item)[:n_dimensions]


def bias_prediction_dense(tf_prediction, tf_projected_user_biases, tf_projected_item_biases):
    """(recommendation_graphs.py:33-41)"""
    return tf_prediction + tf_projected_user_biases[:, None] + tf_projected_item_biases[None, :]


def bias_prediction_serial(tf_prediction_serial, tf_projected_user_biases, tf_projected_item_biases, tf_x_user,
                           tf_x_item):
    """(recommendation_graphs.py:44-57)"""
    return tf_prediction_serial + tf_projected_user_biases[tf_x_user] + tf_projected_item_biases[tf_x_item]


def densify_sampled_item_predictions(tf_sample_predictions_serial, tf_n_sampled_items, tf_n_users):
    """(recommendation_graphs.py:60-70)"""
    return tf_sample_predictions_serial.reshape(int(tf_n_users), int(tf_n_sampled_items))


def rank_predictions(tf_prediction):
    """Ranks per user, 1 = best, ties to the lower index (recommendation_graphs.py:73-82) -- K4 counting kernel."""
    return ops.rank_rows(tf_prediction.detach())


def collapse_mixture_of_tastes(tastes_predictions, tastes_attentions):
    """Max over the tastes, or the softmax(attention)-weighted sum (recommendation_graphs.py:85-109) -- K9, one pass."""
    tastes_predictions = list(tastes_predictions)
    if tastes_attentions is None and len(tastes_predictions) == 1:
        return tastes_predictions[0]                  # reduce_max over one taste
    return ops.collapse_tastes(tastes_predictions, tastes_attentions)


def relative_cosine(tf_tensor_1, tf_tensor_2):
    """(recommendation_graphs.py:112-121)"""
    from .prediction_graphs import CosineSimilarityPredictionGraph
    return CosineSimilarityPredictionGraph().connect_dense_prediction_graph(tf_tensor_1, tf_tensor_2)


def predict_similar_items(prediction_graph_factory, tf_item_representation, tf_similar_items_ids):
    """(recommendation_graphs.py:124-137)"""
    ids = torch.as_tensor(tf_similar_items_ids, dtype=torch.int64, device=tf_item_representation.device)
    gathered_items = tf_item_representation[ids].contiguous()
    return prediction_graph_factory.connect_dense_prediction_graph(
        tf_user_representation=gathered_items,
        tf_item_representation=tf_item_representation
    )
