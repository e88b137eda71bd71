"""
Launchers and autograd bindings for the HIP kernels (include/tensorrec_hip.h).

Each differentiable op is a ``torch.autograd.Function`` whose forward AND backward are hand-written kernels; torch
autograd only strings them together, which is what lets user-defined graphs (plain torch ops) mix with the built-in
ones -- the reference gets the same composability from TF's autodiff.  Nothing here computes on the CPU.

The implementation lives in two modules along the section lines this file used to have -- ops_base.py (K1, GEMM, K3, grouping, K9,
K6 losses, K4, K7, K8) and ops_topk.py (K2 and every exact top-k route) -- and this module re-exports every name of both, so
``from tensorrec_amd import ops; ops.anything`` works as before.  Module-level switches (``ops.KERNEL_EVENTS = []``,
``ops.FILTER_DEBUG = {}``, ``ops.CANDIDATE_STATS = True``, a constant a test overrides) are forwarded to the module that reads them.
"""
import sys as _sys
import types as _types

from . import ops_base as _base
from . import ops_topk as _topk

for _m in (_base, _topk):
    for _n, _v in vars(_m).items():
        if not _n.startswith("__"):
            globals()[_n] = _v


class _OpsModule(_types.ModuleType):
    """Assignments to a name one of the implementation modules defines reach that module (the functions read THEIR module's
    globals): ``ops.KERNEL_EVENTS = []`` switches the event brackets of ops_base._timed on, exactly as it did in the single file."""

    def __setattr__(self, name, value):
        for m in (_base, _topk):
            if name in vars(m):
                setattr(m, name, value)
        super().__setattr__(name, value)


_sys.modules[__name__].__class__ = _OpsModule
