"""Shared tolerance logic of the model-level parity tests: lives with the checker (oracle/parity.py) so that bench.py's
parity_fit record applies the same bar."""
from oracle.parity import check_weights_after_adam  # noqa: F401
