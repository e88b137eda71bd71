"""
Prediction graphs -- same classes and methods as tensorrec/prediction_graphs.py:7-117.

``connect_dense_prediction_graph`` maps (user_repr [U, d], item_repr [I, d]) to scores [U, I];
``connect_serial_prediction_graph`` additionally takes index tensors and returns [n_pairs] scores.  Both are pure
functions of their arguments, also reused for item-item similarity (recommendation_graphs.py:133-136).

The built-in graphs additionally describe themselves to the fused engine through ``engine_mode`` /
``engine_normalize`` so that ``TensorRec.predict*`` can run the MFMA score kernel with the bias, top-k or rank
epilogue fused; a user-defined graph without those attributes is executed as written (torch ops).
"""
import abc

from . import ops


class AbstractPredictionGraph(object):
    __metaclass__ = abc.ABCMeta

    # description for the fused score kernel; None = run connect_dense_prediction_graph as written
    engine_mode = None
    engine_normalize = False

    @abc.abstractmethod
    def connect_dense_prediction_graph(self, tf_user_representation, tf_item_representation):
        """:return: tensor [n_users, n_items] (prediction_graphs.py:10-22)"""
        pass

    @abc.abstractmethod
    def connect_serial_prediction_graph(self, tf_user_representation, tf_item_representation, tf_x_user, tf_x_item):
        """:return: tensor [n_interactions] (prediction_graphs.py:24-40)"""
        pass


def _dense(user_repr, item_repr, mode, normalize, precision):
    dtype = ops.DTYPE_BF16 if precision == 'bf16' else ops.DTYPE_F32
    return ops.dense_scores(user_repr, item_repr, dtype, normalize, mode)


class DotProductPredictionGraph(AbstractPredictionGraph):
    """Prediction = user_repr . item_repr  (prediction_graphs.py:43-55)"""
    engine_mode = ops.MODE_DOT

    def __init__(self, precision='fp32'):
        self.precision = precision

    def connect_dense_prediction_graph(self, tf_user_representation, tf_item_representation):
        return _dense(tf_user_representation, tf_item_representation, ops.MODE_DOT, False, self.precision)

    def connect_serial_prediction_graph(self, tf_user_representation, tf_item_representation, tf_x_user, tf_x_item):
        return ops.pair_score(tf_user_representation, tf_item_representation, tf_x_user, tf_x_item, ops.MODE_DOT)


class CosineSimilarityPredictionGraph(AbstractPredictionGraph):
    """Prediction = cos(user_repr, item_repr)  (prediction_graphs.py:58-72)"""
    engine_mode = ops.MODE_DOT
    engine_normalize = True

    def __init__(self, precision='fp32'):
        self.precision = precision

    def connect_dense_prediction_graph(self, tf_user_representation, tf_item_representation):
        return _dense(tf_user_representation, tf_item_representation, ops.MODE_DOT, True, self.precision)

    def connect_serial_prediction_graph(self, tf_user_representation, tf_item_representation, tf_x_user, tf_x_item):
        normalized_users = ops.l2_normalize_rows(tf_user_representation)
        normalized_items = ops.l2_normalize_rows(tf_item_representation)
        return ops.pair_score(normalized_users, normalized_items, tf_x_user, tf_x_item, ops.MODE_DOT)


class EuclideanSimilarityPredictionGraph(AbstractPredictionGraph):
    """Prediction = -1 * sqrt(sum((user_repr - item_repr)^2))  (prediction_graphs.py:75-117).
    As in the reference the dense form uses r_u - 2 u.i + r_i and the serial form sum((u - i)^2), both clipped at
    1e-16 (``epsilon``), so they differ in the last bits for near-identical vectors."""
    engine_mode = ops.MODE_EUCLIDEAN
    epsilon = 1e-16

    def __init__(self, precision='fp32'):
        self.precision = precision

    def connect_dense_prediction_graph(self, tf_user_representation, tf_item_representation):
        return _dense(tf_user_representation, tf_item_representation, ops.MODE_EUCLIDEAN, False, self.precision)

    def connect_serial_prediction_graph(self, tf_user_representation, tf_item_representation, tf_x_user, tf_x_item):
        return ops.pair_score(tf_user_representation, tf_item_representation, tf_x_user, tf_x_item,
                              ops.MODE_EUCLIDEAN)
