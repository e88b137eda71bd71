"""
oracle/model.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

CPU restatement of ``TensorRec._build_tf_graph`` (tensorrec/tensorrec.py:270-492) and of
one ``session.run(tf_optimizer)`` (tensorrec.py:617-622), including the mixture of tastes
(``n_tastes > 1``: one user representation per taste, max over tastes) and attention
(softmax over per-taste attention scores; for the SAMPLED pairs the reference feeds the
user representation where the attention representation was meant, tensorrec.py:367-372,
so the sampled attentions equal the sampled predictions -- reproduced).  torch-CPU float32
tensors + torch.autograd stand in for the TF graph and TF's autodiff; the optimiser is the TF-1.x Adam form from oracle/oracle.py.

PARITY UNPINNED: the reference has no known-answer test for a fit step.
The quirks of SURVEY.md 3.4 are reproduced on purpose:
  * WMRB returns a [P+] vector, ``tf_loss = vector + alpha * reg`` broadcasts, and
    ``minimize`` differentiates the SUM  -> objective sum_p loss_p + P+ * alpha * reg;
  * dense Adam: every weight element is updated every step;
  * sampled items are shared per user and may include positives;
  * interactions take their shape from the FEATURE matrices (tensorrec.py:294-295).
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp
import torch

from . import oracle as O


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))


def _sparse(features):
    m = sp.coo_matrix(features)
    idx = torch.from_numpy(np.stack([m.row, m.col]).astype(np.int64))
    return torch.sparse_coo_tensor(idx, torch.from_numpy(m.data.astype(np.float32)), m.shape).coalesce()


def _l2n(x, eps=1e-12):
    ss = (x * x).sum(dim=1, keepdim=True)
    return x * torch.rsqrt(torch.clamp(ss, min=eps))


class OracleTensorRec(object):
    """Weights are plain float32 NumPy arrays in ``self.weights`` (name -> array); ``adam_m`` /
    ``adam_v`` hold the slots.  ``repr_kind``: 'linear' | 'normalized_linear' | 'relu' |
    'passthrough' | 'weighted_passthrough'.  ``pred_kind``: 'dot' | 'cosine' | 'euclidean'.
    ``loss_kind``: 'rmse' | 'wmrb' | 'balanced_wmrb'."""

    def __init__(self, n_components, user_repr="linear", item_repr="linear", pred_kind="dot",
                 loss_kind="rmse", biased=True, n_tastes=1, attention=None):
        self.d = n_components
        self.n_tastes, self.attention = n_tastes, attention
        # node_name_ending per taste (tensorrec.py:344, :352); a single taste keeps the short name "user"
        self.user_sides = ["user"] if n_tastes == 1 else ["user_%d" % t for t in range(n_tastes)]
        self.attn_sides = ["attn_%d" % t for t in range(n_tastes)] if attention is not None else []
        self.user_repr, self.item_repr = user_repr, item_repr
        self.pred_kind, self.loss_kind, self.biased = pred_kind, loss_kind, biased
        self.weights, self.adam_m, self.adam_v = {}, {}, {}
        self.t = 0

    # ---- weights -------------------------------------------------------------------------
    def init_weights(self, n_user_features, n_item_features, rng):
        sides = [("item", self.item_repr, n_item_features)]
        for t in range(self.n_tastes):
            sides.append((self.user_sides[t], self.user_repr, n_user_features))
            if self.attention is not None:
                sides.append((self.attn_sides[t], self.attention, n_user_features))
        for side, kind, nf in sides:
            if kind in ("linear", "normalized_linear"):
                self.weights["linear_weights_" + side] = O.init_linear_weights(nf, self.d, rng)
            elif kind == "relu":
                w1, b1, w2 = O.init_relu_weights(nf, self.d, rng)
                self.weights["relu_weights_" + side] = w1
                self.weights["linear_weights_" + side] = w2
                self.weights["relu_biases_" + side] = b1
            elif kind == "weighted_passthrough":
                # tf.ones constant (representation_graphs.py:87): regularised, never trained
                self.weights["passthrough_weights_" + side] = np.ones((1, self.d), np.float32)
        if self.biased:
            self.weights["user_feature_biases"] = np.zeros((n_user_features, 1), np.float32)
            self.weights["item_feature_biases"] = np.zeros((n_item_features, 1), np.float32)
        self.reset_optimizer()

    def reset_optimizer(self):
        self.adam_m = {k: np.zeros_like(v) for k, v in self.weights.items()}
        self.adam_v = {k: np.zeros_like(v) for k, v in self.weights.items()}
        self.t = 0

    def trainable(self, name):
        return not name.startswith("passthrough_weights_")

    # ---- graph (torch) ---------------------------------------------------------------------
    def _repr(self, kind, side, feats, W):
        if kind == "linear":
            return torch.sparse.mm(feats, W["linear_weights_" + side])
        if kind == "normalized_linear":
            return _l2n(torch.sparse.mm(feats, W["linear_weights_" + side]))
        if kind == "relu":
            h = torch.relu(torch.sparse.mm(feats, W["relu_weights_" + side]) + W["relu_biases_" + side])
            return h @ W["linear_weights_" + side]
        dense = feats.to_dense()
        if dense.shape[1] != self.d:
            raise ValueError("FeaturePassThroughRepresentationGraph requires n_features == n_components")
        if kind == "passthrough":
            return dense
        if kind == "weighted_passthrough":
            return dense * W["passthrough_weights_" + side]
        raise ValueError(kind)

    def _serial(self, u, v, xu, xi):
        if self.pred_kind == "dot":
            return (u[xu] * v[xi]).sum(dim=1)
        if self.pred_kind == "cosine":
            return (_l2n(u)[xu] * _l2n(v)[xi]).sum(dim=1)
        if self.pred_kind == "euclidean":
            dist = ((u[xu] - v[xi]) ** 2).sum(dim=1)
            return -1.0 * torch.sqrt(torch.clamp(dist, min=1e-16))
        raise ValueError(self.pred_kind)

    def _dense(self, u, v):
        if self.pred_kind == "dot":
            return u @ v.t()
        if self.pred_kind == "cosine":
            return _l2n(u) @ _l2n(v).t()
        r_u = (u ** 2).sum(1, keepdim=True)
        r_v = (v ** 2).sum(1, keepdim=True)
        dist = (r_u - 2.0 * (u @ v.t())) + r_v.t()
        return -1.0 * torch.sqrt(torch.clamp(dist, min=1e-16))

    @staticmethod
    def _collapse(preds, attns):                                  # recommendation_graphs.py:85-109
        stacked = torch.stack(preds)
        if attns is not None:
            return (stacked * torch.softmax(torch.stack(attns), dim=0)).sum(dim=0)
        return torch.amax(stacked, dim=0)                         # ties share the gradient evenly, as tf.reduce_max

    def _serial_all(self, o, xu, xi, sampled=False):
        preds = [self._serial(u, o["item_repr"], xu, xi) for u in o["user_reprs"]]
        attns = None
        if self.attention is not None:
            # tensorrec.py:367-372: the sampled attention is built from tf_user_representation
            attns = preds if sampled else [self._serial(a, o["item_repr"], xu, xi) for a in o["attn_reprs"]]
        s = self._collapse(preds, attns)
        if self.biased:
            s = s + o["user_bias"][xu] + o["item_bias"][xi]
        return s

    def forward(self, W, user_features, item_features, xu=None, xi=None):
        uf, itf = _sparse(user_features), _sparse(item_features)
        item_repr = self._repr(self.item_repr, "item", itf, W)
        user_reprs = [self._repr(self.user_repr, side, uf, W) for side in self.user_sides]
        attn_reprs = [self._repr(self.attention, side, uf, W) for side in self.attn_sides]
        out = {"user_repr": user_reprs[0], "user_reprs": user_reprs, "attn_reprs": attn_reprs, "item_repr": item_repr}
        if self.biased:
            out["user_bias"] = torch.sparse.mm(uf, W["user_feature_biases"]).sum(dim=1)
            out["item_bias"] = torch.sparse.mm(itf, W["item_feature_biases"]).sum(dim=1)
        if xu is not None:
            out["serial"] = self._serial_all(out, xu, xi)
        return out

    # ---- predict ----------------------------------------------------------------------------
    def _W(self, requires_grad=False):
        W = {}
        for k, v in self.weights.items():
            t = _t(v).clone()
            if requires_grad and self.trainable(k):
                t.requires_grad_(True)
            W[k] = t
        return W

    def predict(self, user_features, item_features):
        with torch.no_grad():
            W = self._W()
            o = self.forward(W, user_features, item_features)
            preds = [self._dense(u, o["item_repr"]) for u in o["user_reprs"]]
            attns = [self._dense(a, o["item_repr"]) for a in o["attn_reprs"]] if self.attention is not None else None
            pred = self._collapse(preds, attns)
            if self.biased:
                pred = pred + o["user_bias"][:, None] + o["item_bias"][None, :]
            return pred.numpy()

    def predict_rank(self, user_features, item_features):
        return O.rank_predictions_exact(self.predict(user_features, item_features))

    def representations(self, user_features, item_features):
        with torch.no_grad():
            o = self.forward(self._W(), user_features, item_features)
            return {k: (v.numpy() if torch.is_tensor(v) else [x.numpy() for x in v]) for k, v in o.items()}

    # ---- one optimiser step (tensorrec.py:617-622) --------------------------------------------
    def loss_and_grads(self, interactions, user_features, item_features, alpha, sample_items=None):
        """``sample_items``: int array [n_users, S] (the [U*S, 2] pairs of util.sample_items
        reshaped user-major).  Returns (basic_loss ndarray, weight_reg_loss float, grads dict)."""
        rows, cols, vals, _ = O.to_coo_like_reference(interactions)
        n_users = sp.coo_matrix(user_features).shape[0]      # tensorrec.py:294-295
        n_items = sp.coo_matrix(item_features).shape[0]
        xu, xi = torch.from_numpy(rows), torch.from_numpy(cols)
        y = torch.from_numpy(vals)
        W = self._W(requires_grad=True)
        o = self.forward(W, user_features, item_features, xu, xi)
        pred_serial = o["serial"]

        if self.loss_kind == "rmse":                                   # loss_graphs.py:58-59
            basic = torch.sqrt(torch.mean((y - pred_serial) ** 2))
        elif self.loss_kind in ("wmrb", "balanced_wmrb"):               # loss_graphs.py:153-227
            S = sample_items.shape[1]
            su = torch.arange(n_users).repeat_interleave(S)
            si = torch.from_numpy(np.ascontiguousarray(sample_items, np.int64).reshape(-1))
            samp = self._serial_all(o, su, si, sampled=True)
            samp = samp.reshape(n_users, S)                              # recommendation_graphs.py:68-69
            mask = y > 0.0
            pos_pred = pred_serial[mask]
            mapped = samp[xu[mask]]
            summation = torch.clamp(1.0 - pos_pred[:, None] + mapped, min=0.0)
            ratio = torch.tensor(float(n_items), dtype=torch.float32) / torch.tensor(float(S), dtype=torch.float32)
            smr = ratio * summation.sum(dim=1)
            if self.loss_kind == "balanced_wmrb":
                pos_vals = y[mask]
                per_item = torch.zeros(n_items, dtype=torch.float32).index_add_(0, xi[mask], pos_vals)
                smr = smr * pos_vals / per_item[xi[mask]]
            basic = torch.log(smr + 1.0)
        elif self.loss_kind in ("rmse_dense", "separation", "separation_dense"):     # loss_graphs.py:62-134
            def separation(pos, neg):
                loc = neg.mean() - pos.mean()
                scale = torch.sqrt(((neg - neg.mean()) ** 2).mean() + ((pos - pos.mean()) ** 2).mean())
                return 1.0 - 0.5 * (1.0 + torch.erf((0.0 - loc) / (scale * 1.4142135623730951)))

            if self.loss_kind == "separation":
                basic = separation(pred_serial[y > 0.0], pred_serial[y <= 0.0])
            else:
                preds = [self._dense(u_, o["item_repr"]) for u_ in o["user_reprs"]]
                attns = [self._dense(a_, o["item_repr"]) for a_ in o["attn_reprs"]] if self.attention is not None else None
                dense_pred = self._collapse(preds, attns)
                if self.biased:
                    dense_pred = dense_pred + o["user_bias"][:, None] + o["item_bias"][None, :]
                dense_inter = torch.zeros((n_users, n_items), dtype=torch.float32).index_put_((xu, xi), y, accumulate=True)
                if self.loss_kind == "rmse_dense":
                    basic = torch.sqrt(torch.mean((dense_inter - dense_pred) ** 2))
                else:
                    flat_p, flat_y = dense_pred.reshape(-1), dense_inter.reshape(-1)
                    basic = separation(flat_p[flat_y > 0.0], flat_p[flat_y <= 0.0])
        else:
            raise ValueError(self.loss_kind)

        reg = sum(0.5 * (w ** 2).sum() for w in W.values())              # tensorrec.py:487 (tf.nn.l2_loss)
        total = basic + torch.tensor(alpha, dtype=torch.float32) * reg    # :488 (broadcasts for vector losses)
        total.sum().backward()                                            # TF sums gradients of non-scalar ys
        grads = {k: (w.grad.numpy().copy() if w.grad is not None else None) for k, w in W.items()}
        return basic.detach().numpy().copy(), float(reg.detach()), grads, pred_serial.detach().numpy().copy()

    def step(self, interactions, user_features, item_features, learning_rate, alpha, sample_items=None):
        basic, reg, grads, pred_serial = self.loss_and_grads(interactions, user_features, item_features, alpha,
                                                             sample_items)
        self.last_grads = grads                  # d(loss + alpha * reg) / dw of this step (bench.py's parity_fit reads them)
        self.t += 1
        lr_t = O.adam_lr_t(learning_rate, self.t)
        for k, g in grads.items():
            if g is None:
                continue
            O.adam_tf_step(self.weights[k], self.adam_m[k], self.adam_v[k], g.astype(np.float32), lr_t)
        return basic, reg, pred_serial
