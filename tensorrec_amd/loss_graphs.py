"""
Loss graphs -- same classes, flags and kwargs contract as tensorrec/loss_graphs.py:5-227.

``connect_loss_graph`` receives the kwargs selected by the class flags (tensorrec/tensorrec.py:463-482) and returns a
scalar OR a vector; the trainer differentiates the SUM of ``loss + alpha * reg`` exactly as TF does for a non-scalar
``tf_loss``.  Subclasses take what they need and swallow the rest with ``**kwargs``
(test/test_readme.py:83-95).
"""
import abc

import torch

from . import ops


class AbstractLossGraph(object):
    __metaclass__ = abc.ABCMeta

    # If True, dense prediction results will be passed to the loss function
    is_dense = False

    # If True, randomly sampled predictions will be passed to the loss function
    is_sample_based = False
    # If True, and if is_sample_based is True, predictions will be sampled with replacement
    is_sampled_with_replacement = False

    @abc.abstractmethod
    def connect_loss_graph(self, tf_prediction_serial, tf_interactions_serial, tf_interactions, tf_n_users, tf_n_items,
                           tf_prediction, tf_rankings, tf_sample_predictions, tf_n_sampled_items):
        """
        Always passed: tf_prediction_serial [n_interactions], tf_interactions_serial [n_interactions],
        tf_interactions (sparse.Interactions: .indices, .values, .dense_shape), tf_n_users, tf_n_items.
        If is_dense: tf_prediction [n_users, n_items], tf_rankings [n_users, n_items].
        If is_sample_based: tf_sample_predictions [n_users, n_sampled_items], tf_n_sampled_items.
        :return: the loss value (scalar or vector tensor).
        """
        pass


class RMSELossGraph(AbstractLossGraph):
    """Root mean square error over the interactions (loss_graphs.py:53-59)."""

    def connect_loss_graph(self, tf_prediction_serial, tf_interactions_serial, **kwargs):
        return ops.rmse_loss(tf_prediction_serial, tf_interactions_serial)


class RMSEDenseLossGraph(AbstractLossGraph):
    """RMSE against the dense interaction matrix, non-interacted pairs counting as 0 (loss_graphs.py:62-72)."""
    is_dense = True

    def connect_loss_graph(self, tf_interactions, tf_prediction, **kwargs):
        error = -1.0 * tf_prediction
        flat = error.reshape(-1)
        lin = tf_interactions.x_user * tf_prediction.shape[1] + tf_interactions.x_item
        flat = flat.index_add(0, lin, tf_interactions.values)          # tf.sparse_add(interactions, -prediction)
        return torch.sqrt(torch.mean(flat * flat))


def _separation(pos, neg):
    """1 - Normal(neg_mean - pos_mean, sqrt(neg_var + pos_var)).cdf(0)  (loss_graphs.py:90-96); tf.nn.moments is
    the population variance."""
    pos_mean, neg_mean = pos.mean(), neg.mean()
    pos_var = ((pos - pos_mean) ** 2).mean()
    neg_var = ((neg - neg_mean) ** 2).mean()
    loc = neg_mean - pos_mean
    scale = torch.sqrt(neg_var + pos_var)
    cdf0 = 0.5 * (1.0 + torch.erf((0.0 - loc) / (scale * 1.4142135623730951)))
    return 1.0 - cdf0


class SeparationLossGraph(AbstractLossGraph):
    """Overlap of the normal fits of positive and non-positive interaction predictions (loss_graphs.py:75-97)."""

    def connect_loss_graph(self, tf_prediction_serial, tf_interactions_serial, **kwargs):
        positive = tf_prediction_serial[tf_interactions_serial > 0.0]
        negative = tf_prediction_serial[tf_interactions_serial <= 0.0]
        return _separation(positive, negative)


class SeparationDenseLossGraph(AbstractLossGraph):
    """Separation loss over the dense matrix, non-interacted pairs counting as negatives (loss_graphs.py:100-134)."""
    is_dense = True

    def connect_loss_graph(self, tf_prediction, tf_interactions, **kwargs):
        dense = torch.zeros(tf_prediction.shape, dtype=torch.float32, device=tf_prediction.device)
        dense.index_put_((tf_interactions.x_user, tf_interactions.x_item), tf_interactions.values, accumulate=True)
        inter_serial = dense.reshape(-1)
        pred_serial = tf_prediction.reshape(-1)
        return _separation(pred_serial[inter_serial > 0.0], pred_serial[inter_serial <= 0.0])


class WMRBLossGraph(AbstractLossGraph):
    """
    Approximation of http://ceur-ws.org/Vol-1905/recsys2017_poster3.pdf  (loss_graphs.py:137-180).
    Interactions can be any positive values, but magnitude is ignored. Negative interactions are ignored.
    Returns the [n_positive_interactions] vector, as the reference does.
    """
    is_sample_based = True
    balanced = False

    def connect_loss_graph(self, tf_prediction_serial, tf_interactions, tf_sample_predictions, tf_n_items,
                           tf_n_sampled_items, **kwargs):
        return self.weighted_margin_rank_batch(tf_prediction_serial=tf_prediction_serial,
                                               tf_interactions=tf_interactions,
                                               tf_sample_predictions=tf_sample_predictions,
                                               tf_n_items=tf_n_items,
                                               tf_n_sampled_items=tf_n_sampled_items)

    def weighted_margin_rank_batch(self, tf_prediction_serial, tf_interactions, tf_sample_predictions, tf_n_items,
                                   tf_n_sampled_items):
        # one fused kernel per direction (K6); n_items / n_sampled_items come from the tensors' shapes
        return ops.wmrb_loss(tf_prediction_serial, tf_sample_predictions, tf_interactions, balanced=self.balanced)


class BalancedWMRBLossGraph(WMRBLossGraph):
    """WMRB weighted by value / sum(positive values of the item)  (loss_graphs.py:183-227)."""
    balanced = True
