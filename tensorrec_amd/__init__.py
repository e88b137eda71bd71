"""
tensorrec_amd -- an MI355X-native engine behind TensorRec's ``fit / predict / predict_rank`` API and its pluggable
RepresentationGraph / PredictionGraph / LossGraph classes (module layout of tensorrec/__init__.py:1-18).

Importing the package does not need a GPU; running any model does (there is no CPU execution path).
"""
from .tensorrec import TensorRec, DeviceSampler, HostSampler, ReplaySampler
from . import errors
from . import eval  # noqa: A004
from . import framework
from . import input_utils
from . import loss_graphs
from . import prediction_graphs
from . import recommendation_graphs
from . import representation_graphs
from . import util

__version__ = '0.1.0'

__all__ = [
    "TensorRec", "DeviceSampler", "HostSampler", "ReplaySampler", "errors", "eval", "framework", "input_utils", "loss_graphs",
    "prediction_graphs", "recommendation_graphs", "representation_graphs", "util",
]
