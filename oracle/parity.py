"""Shared tolerance logic of the model-level parity checks (tests/test_gpu_model.py, tests/test_gpu_shapes.py,
bench.py parity_fit).  Test infrastructure, like everything under oracle/: never imported by the product."""
import numpy as np


def check_weights_after_adam(got, ref, g_gpu, g_ref, lr, steps, exempt=(), label=""):
    """Weights after ``steps`` Adam steps from identical initial weights.

    TF-form Adam moves an entry by ~lr * g / |g| whatever the size of g, so an entry's weight error is lr times the
    RELATIVE error of its gradient: entries with a gradient far above the fp32 summation noise must agree to 1e-4 * lr
    per step, entries whose gradient is small against that noise get a bound that grows like noise / |g| (capped at one
    full step per step), and tensors whose gradient is ZERO in exact arithmetic (``exempt``, named by the caller with
    the reason) move by pure noise -- up to lr per step on either side.  ``g_gpu`` / ``g_ref``: the raw first-step
    gradients of both sides (the tight check on those is the caller's 1e-4 * gmax assertion); their entrywise
    difference, floored at 1e-7 * gmax, is the noise estimate.  Returns {tensor: (max |dw|, share of entries beyond
    1e-4 * lr per step, entries beyond the Adam-aware bound)} for the log."""
    gmax = max(np.abs(g).max() for g in g_ref.values() if g is not None)
    report = {}
    for k, r in ref.items():
        d = np.abs(got[k] - r)
        report[k] = (float(d.max()), float((d > 1e-4 * lr * steps).mean()))
        if k in exempt:
            assert d.max() <= 2.0 * lr * steps + 1e-6, "%s %s: exempt tensor moved by more than two full steps per step" % (label, k)
            continue
        if g_ref.get(k) is None:
            assert d.max() <= 1e-6, "%s %s: untrained tensor changed" % (label, k)
            continue
        gg = g_gpu.get(k)
        noise = np.maximum(np.abs(gg - g_ref[k]) if gg is not None else 0.0, 1e-7 * gmax)
        tol = lr * steps * np.minimum(1.0, 1e-4 + 8.0 * noise / (np.abs(g_ref[k]) + 1e-30)) + 1e-7
        bad = d > tol
        # Non-differentiable points: a hinge max(0, 1 - y_ui + y_us) (or a ReLU unit) that sits within ~1e-7 of its kink
        # switches on one side and not on the other once the weights differ in the last bit (steps 2, 3); each switch
        # moves the two rows involved (one user row, one item row) by O(lr).  With millions of hinge terms per step a
        # few switches are expected, so up to 0.2% of a tensor's entries may leave the bound -- by at most the full steps.
        assert bad.mean() <= 2e-3, "%s %s: %d entries (%.3f%%) beyond the Adam-aware bound (worst |dw| %g at |g| %g, tol %g)" % (
            label, k, int(bad.sum()), 100 * bad.mean(), float(d[bad].max()), float(np.abs(g_ref[k])[bad].min()),
            float(tol[bad].min()))
        assert d.max() <= 2.0 * lr * steps + 1e-6, "%s %s: an entry moved by more than two full steps per step" % (label, k)
        report[k] = report[k] + (int(bad.sum()),)
    return report
