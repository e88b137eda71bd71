"""Exception types of the public API -- same names and messages as tensorrec/errors.py:4-41."""


class TensorRecException(Exception):
    msg = None

    def __init__(self, **kwargs):
        self.kwargs = kwargs
        super(TensorRecException, self).__init__(self.msg.format(**kwargs))

    @property
    def message(self):
        return str(self)

    def __str__(self):
        return self.msg.format(**self.kwargs)

    __repr__ = __str__


class ModelNotBiasedException(TensorRecException):
    msg = 'Cannot predict {actor} bias for unbiased model'


class ModelNotFitException(TensorRecException):
    msg = "{method}() has been called before model fitting. Call fit() or fit_partial() before calling {method}()."


class ModelWithoutAttentionException(TensorRecException):
    msg = "This TensorRec model does not use attention. Try re-building TensorRec with a valid 'attention_graph' arg."


class BatchNonSparseInputException(TensorRecException):
    msg = 'In order to support user batching at fit time, interactions and user_features must both be scipy.sparse ' \
          'matrices.'
