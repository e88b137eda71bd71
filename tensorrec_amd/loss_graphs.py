"""
Loss graphs -- same classes, flags and kwargs contract as tensorrec/loss_graphs.py:5-227.

``connect_loss_graph`` receives the kwargs selected by the class flags (tensorrec/tensorrec.py:463-482) and returns a
scalar OR a vector; the trainer differentiates the SUM of ``loss + alpha * reg`` exactly as TF does for a non-scalar
``tf_loss``.  Subclasses take what they need and swallow the rest with ``**kwargs``
(test/test_readme.py:83-95).
"""
import abc

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
    """RMSE against the dense interaction matrix, non-interacted pairs counting as 0 (loss_graphs.py:62-72): one streaming
    reduction over the [n_users, n_items] predictions plus a gather over the interactions (csrc/loss_dense.hip) -- the dense
    interaction matrix of ``tf.sparse_add`` is never built."""
    is_dense = True

    def connect_loss_graph(self, tf_interactions, tf_prediction, **kwargs):
        return ops.rmse_dense_loss(tf_prediction, tf_interactions)


class SeparationLossGraph(AbstractLossGraph):
    """Overlap of the normal fits of positive and non-positive interaction predictions (loss_graphs.py:75-97):
    1 - Normal(mu_neg - mu_pos, sqrt(var_neg + var_pos)).cdf(0) with tf.nn.moments' population variance -- two reduction passes
    (means, centred second moments) and a closed-form backward pass, no boolean-mask copies."""

    def connect_loss_graph(self, tf_prediction_serial, tf_interactions_serial, **kwargs):
        return ops.separation_loss(tf_prediction_serial, tf_interactions_serial)


class SeparationDenseLossGraph(AbstractLossGraph):
    """Separation loss over the dense matrix, non-interacted pairs counting as negatives (loss_graphs.py:100-134): the
    negatives' moments are "all predictions minus the positive interactions'", so neither the dense interaction matrix nor the
    masks exist."""
    is_dense = True

    def connect_loss_graph(self, tf_prediction, tf_interactions, **kwargs):
        return ops.separation_dense_loss(tf_prediction, tf_interactions)


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
