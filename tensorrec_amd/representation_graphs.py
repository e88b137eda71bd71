"""
Representation graphs -- same class names, method names and argument meaning as
tensorrec/representation_graphs.py:5-159.  ``tf_features`` is a ``sparse.SparseFeatures`` (device CSR) instead of a
``tf.SparseTensor``; the return value is still ``(repr [n, n_components], [weights...])`` where the weights list is
what gets L2-regularised (tensorrec/tensorrec.py:313, :346, :487).  Methods are executed every step (eagerly).
"""
import abc

from . import ops
from .framework import Variable, random_normal, zeros, ones


class AbstractRepresentationGraph(object):
    __metaclass__ = abc.ABCMeta

    @abc.abstractmethod
    def connect_representation_graph(self, tf_features, n_components, n_features, node_name_ending):
        """
        Connects the user/item features to their latent representations (representation_graphs.py:8-23).
        :param tf_features: SparseFeatures of shape [ n_users, n_features ]
        :param n_components: int -- size of the latent representation
        :param n_features: int -- number of input features
        :param node_name_ending: str -- 'user_<taste>', 'item' or 'attn_<taste>'; use it in Variable names
        :return: (tensor [ n_users, n_components ], list of weight tensors to regularise)
        """
        pass


class LinearRepresentationGraph(AbstractRepresentationGraph):
    """Linear embedding: repr = features . W  (representation_graphs.py:26-43).  K1 gather-SpMM."""

    def _weights(self, n_components, n_features, node_name_ending):
        # random_normal(stddev=1) rows, L2-normalised at initialisation only (:35-36)
        return Variable(lambda: ops.l2_normalize_rows(random_normal([n_features, n_components], stddev=1.0)),
                        name='linear_weights_{}'.format(node_name_ending))

    def connect_representation_graph(self, tf_features, n_components, n_features, node_name_ending):
        tf_linear_weights = self._weights(n_components, n_features, node_name_ending)
        tf_repr = ops.sparse_dense_matmul(tf_features, tf_linear_weights)
        return tf_repr, [tf_linear_weights]


class NormalizedLinearRepresentationGraph(LinearRepresentationGraph):
    """Linear embedding followed by a row L2-normalisation (representation_graphs.py:46-58), fused in K1's epilogue."""

    def connect_representation_graph(self, tf_features, n_components, n_features, node_name_ending):
        tf_linear_weights = self._weights(n_components, n_features, node_name_ending)
        normalized_repr = ops.sparse_dense_matmul_l2norm(tf_features, tf_linear_weights)
        return normalized_repr, [tf_linear_weights]


class FeaturePassThroughRepresentationGraph(AbstractRepresentationGraph):
    """Uses the features as the representation (representation_graphs.py:61-74)."""

    def connect_representation_graph(self, tf_features, n_components, n_features, node_name_ending):
        if n_components != n_features:
            raise ValueError('{} requires n_features and n_components to be equal. Either adjust n_components or use a '
                             'different representation graph. n_features = {}, n_components = {}'.format(
                                self.__class__.__name__, n_features, n_components
                             ))
        return ops.sparse_to_dense(tf_features), []


class WeightedFeaturePassThroughRepresentationGraph(FeaturePassThroughRepresentationGraph):
    """Pass-through multiplied by a weight row.  In the reference the weights are ``tf.ones`` -- a constant, not a
    variable (representation_graphs.py:87) -- so they are regularised but never trained; reproduced as such."""

    def connect_representation_graph(self, tf_features, n_components, n_features, node_name_ending):
        dense_repr, _ = super(WeightedFeaturePassThroughRepresentationGraph, self).connect_representation_graph(
            tf_features=tf_features, n_components=n_components, n_features=n_features, node_name_ending=node_name_ending
        )
        weights = ones([1, n_components])
        weighted_repr = dense_repr * weights
        return weighted_repr, [weights]


class ReLURepresentationGraph(AbstractRepresentationGraph):
    """Single hidden ReLU layer: relu(features . W1 + b1) . W2  (representation_graphs.py:92-124).
    :param relu_size: int or None -- hidden width; None means 4 * n_components."""

    def __init__(self, relu_size=None):
        self.relu_size = relu_size

    def connect_representation_graph(self, tf_features, n_components, n_features, node_name_ending):
        relu_size = 4 * n_components if self.relu_size is None else self.relu_size

        tf_relu_weights = Variable(lambda: random_normal([n_features, relu_size], stddev=.5),
                                   name='relu_weights_{}'.format(node_name_ending))
        tf_relu_biases = Variable(lambda: zeros([1, relu_size]), name='relu_biases_{}'.format(node_name_ending))
        tf_linear_weights = Variable(lambda: random_normal([relu_size, n_components], stddev=.5),
                                     name='linear_weights_{}'.format(node_name_ending))

        # SpMM + bias + ReLU is one kernel (K1 epilogue 2); the dense layer runs on fp32 MFMA
        tf_relu = ops.sparse_dense_matmul_bias_relu(tf_features, tf_relu_weights, tf_relu_biases)
        tf_repr = ops.matmul(tf_relu, tf_linear_weights)
        return tf_repr, [tf_relu_weights, tf_linear_weights, tf_relu_biases]


class AbstractTorchRepresentationGraph(AbstractRepresentationGraph):
    """Counterpart of AbstractKerasRepresentationGraph (representation_graphs.py:127-159; Keras is not available
    here): override ``create_layers`` to return torch.nn modules.  The first layer receives the DENSE feature
    matrix; every parameter of every layer is trained and regularised, as Keras ``layer.weights`` are."""
    __metaclass__ = abc.ABCMeta

    def connect_representation_graph(self, tf_features, n_components, n_features, node_name_ending):
        from .framework import current_store
        store = current_store()
        key = '_torch_layers_{}'.format(node_name_ending)
        if not hasattr(store, key):
            layers = self.create_layers(n_features=n_features, n_components=n_components)
            weights = []
            for li, layer in enumerate(layers):
                layer.to(store.device)
                for pname, p in layer.named_parameters():
                    v = Variable(p.detach().float(), name='torch_{}_{}_{}'.format(node_name_ending, li, pname))
                    weights.append((layer, pname, v))
            setattr(store, key, (layers, weights))
        layers, weights = getattr(store, key)
        last_layer = ops.sparse_to_dense(tf_features)
        import torch
        for layer in layers:
            params = {pname: v for (ly, pname, v) in weights if ly is layer}
            last_layer = torch.func.functional_call(layer, params, (last_layer,))
        return last_layer, [v for (_, _, v) in weights]

    @abc.abstractmethod
    def create_layers(self, n_features, n_components):
        """Returns a list of torch.nn.Module layers mapping n_features -> n_components."""
        pass
