"""oracle/train_oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

CPU restatement of ONE optimizer step of the reference's training loop (SURVEY.md section 8
row f2): /root/reference/src/chessrl/selfplay.py:98-108 -> agent.py:64-89 ->
model.py:83-99 (``fit_generator``), compiled at model.py:69-72 with ``Adam(lr=0.002)`` and the
losses ``categorical_crossentropy`` (policy_out) + ``mean_squared_error`` (value_out), every conv /
dense kernel carrying ``kernel_regularizer='l2'`` (model.py:33-58,113-118).

Written independently of chessrl_amd/train.py: weights stay in their Keras layouts as leaf
tensors (conv HWIO, dense (in,out)); BatchNormalization in training mode is spelled out (batch
mean, biased variance for the normalisation, moving statistics with momentum 0.99 and the
unbiased variance); the Adam update is spelled out per tensor in float64 scalars / float32
tensors.  Gradients come from torch CPU autograd in float32.

PARITY STATUS: "parity unpinned" -- TensorFlow / Keras are absent from this image, the
reference holds no test or fixture for its training step, and the Keras semantics restated here
(l2(0.01), crossentropy clipping at 1e-7, Adam epsilon 1e-7 outside the bias correction, fused
batch-norm moving variance) are from recollection of TF 2.x.  The GPU trainer is held to THIS
restatement (loss, gradients, moving statistics, updated weights).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import numpy as np
import torch
import torch.nn.functional as F

BN_EPS, BN_MOMENTUM = 1e-3, 0.99
L2 = 0.01
LR, B1, B2, EPS = 0.002, 0.9, 0.999, 1e-7
IN_PLANES = 127

_STATS = (".mean", ".var")


def trainable_names(w):
    """Kernels, biases, BN gamma/beta -- everything but the moving statistics and the meta keys."""
    return [k for k in w if not k.startswith("meta.") and not k.endswith(_STATS)]


def regularized_names(w):
    return [k for k in w if k.endswith(".kernel")]


def _bn_train(x, gamma, beta, stats, name, new_stats):
    mean = x.mean(dim=(0, 2, 3))
    var_b = ((x - mean.view(1, -1, 1, 1)) ** 2).mean(dim=(0, 2, 3))          # biased
    n = x.shape[0] * x.shape[2] * x.shape[3]
    var_u = var_b * (n / max(n - 1, 1))                                      # unbiased, moving stat
    new_stats[name + ".mean"] = (BN_MOMENTUM * stats[name + ".mean"] +
                                 (1 - BN_MOMENTUM) * mean.detach().numpy()).astype(np.float32)
    new_stats[name + ".var"] = (BN_MOMENTUM * stats[name + ".var"] +
                                (1 - BN_MOMENTUM) * var_u.detach().numpy()).astype(np.float32)
    sh = (1, -1, 1, 1)
    return (x - mean.view(sh)) / torch.sqrt(var_b.view(sh) + BN_EPS) * gamma.view(sh) + beta.view(sh)


def loss_and_grads(w, planes, move_index, result):
    """Training-mode forward + backward.

    planes (B,8,8,127) 0/1 floats, move_index (B,) ints in [0,1968), result (B,) in {-1,0,1}.
    Returns (losses dict, grads {name: ndarray in Keras layout}, new moving stats {name: ndarray},
    policy (B,1968), value (B,)).
    """
    blocks = int(w["meta.blocks"])
    t = {k: torch.tensor(np.asarray(w[k], np.float32), requires_grad=True) for k in trainable_names(w)}
    new_stats = {}

    def conv(x, name, pad):
        return F.conv2d(x, t[name + ".kernel"].permute(3, 2, 0, 1), t[name + ".bias"], padding=pad)

    def bn(x, name):
        return _bn_train(x, t[name + ".gamma"], t[name + ".beta"], w, name, new_stats)

    x = torch.as_tensor(np.asarray(planes))[..., :IN_PLANES].to(torch.float32).permute(0, 3, 1, 2)
    x = conv(x, "stem", 1)
    for i in range(blocks):
        y = F.relu(bn(conv(x, "block%d.conv1" % i, 1), "block%d.bn1" % i))
        y = bn(conv(y, "block%d.conv2" % i, 1), "block%d.bn2" % i)
        x = F.relu(x + y)
    p = F.relu(bn(conv(x, "policy.conv", 0), "policy.bn"))
    p = p.permute(0, 2, 3, 1).reshape(p.shape[0], -1)
    p = torch.softmax(p @ t["policy.dense.kernel"] + t["policy.dense.bias"], dim=-1)
    v = F.relu(bn(conv(x, "value.conv", 0), "value.bn"))
    v = v.permute(0, 2, 3, 1).reshape(v.shape[0], -1)
    v = F.relu(v @ t["value.dense1.kernel"] + t["value.dense1.bias"])
    v = torch.tanh(v @ t["value.dense2.kernel"] + t["value.dense2.bias"])[:, 0]

    idx = torch.as_tensor(np.asarray(move_index), dtype=torch.int64)
    z = torch.as_tensor(np.asarray(result), dtype=torch.float32)
    # keras.backend.categorical_crossentropy on probabilities
    q = p / p.sum(dim=-1, keepdim=True)
    q = torch.clamp(q, 1e-7, 1.0 - 1e-7)
    onehot = F.one_hot(idx, p.shape[1]).to(torch.float32)
    cce = (-(onehot * torch.log(q)).sum(dim=-1)).mean()
    mse = ((z - v) ** 2).mean()
    reg = sum(L2 * (t[k] ** 2).sum() for k in regularized_names(w))
    total = cce + mse + reg
    total.backward()
    grads = {k: (t[k].grad.numpy().copy() if t[k].grad is not None else np.zeros_like(w[k]))
             for k in t}
    losses = {"loss": total.item(), "policy_out_loss": cce.item(), "value_out_loss": mse.item(),
              "reg_loss": reg.item()}
    return losses, grads, new_stats, p.detach(), v.detach()


class AdamState(object):
    def __init__(self, w):
        self.t = 0
        self.m = {k: np.zeros_like(np.asarray(w[k], np.float32)) for k in trainable_names(w)}
        self.v = {k: np.zeros_like(np.asarray(w[k], np.float32)) for k in trainable_names(w)}


def adam_update(w, grads, state):
    """TF2 Keras Adam: lr_t = lr*sqrt(1-b2^t)/(1-b1^t); w -= lr_t * m / (sqrt(v) + eps)."""
    state.t += 1
    lr_t = np.float32(LR * np.sqrt(1.0 - B2 ** state.t) / (1.0 - B1 ** state.t))
    out = dict(w)
    for k in trainable_names(w):
        g = grads[k].astype(np.float32)
        state.m[k] = (np.float32(B1) * state.m[k] + np.float32(1.0 - B1) * g).astype(np.float32)
        state.v[k] = (np.float32(B2) * state.v[k] + np.float32(1.0 - B2) * g * g).astype(np.float32)
        out[k] = (np.asarray(w[k], np.float32) -
                  lr_t * state.m[k] / (np.sqrt(state.v[k]) + np.float32(EPS))).astype(np.float32)
    return out


def train_step(w, planes, move_index, result, state):
    """One ``train_on_batch``: returns (new weight dict, losses, grads)."""
    losses, grads, new_stats, _, _ = loss_and_grads(w, planes, move_index, result)
    out = adam_update(w, grads, state)
    out.update(new_stats)
    return out, losses, grads
