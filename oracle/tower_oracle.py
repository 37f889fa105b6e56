"""oracle/tower_oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Plain PyTorch fp32 CPU restatement of the reference's residual policy/value
tower (/root/reference/src/chessrl/model.py:31-63, 111-122) with Keras
inference semantics: NHWC input (B,8,8,127); stem
Conv3x3 'same' with bias, no BN/activation; N x [Conv3x3 -> BN -> ReLU ->
Conv3x3 -> BN -> +input -> ReLU]; policy head Conv1x1(2) -> BN -> ReLU ->
Flatten(h,w,c) -> Dense(1968) softmax; value head Conv1x1(1) -> BN -> ReLU ->
Flatten -> Dense(256) ReLU -> Dense(1) tanh.  BatchNormalization eps = 1e-3,
moving statistics at inference.

PARITY STATUS: TensorFlow is absent from this image and the reference ships no
weights, so the NUMERICS of this restatement are "parity unpinned" against Keras
itself; its TOPOLOGY is pinned: tests/golden/model_graph.json is the layer graph
the reference's own constructor builds (recorded by oracle/ref_loader.
record_model_graph), and tests/test_model_graph.py executes that graph with these
weights and gets this module's outputs.  It is the fp32 reference the fp16 MFMA
tower is held to (|diff| <= 1e-3).

Weights live in a flat dict of numpy arrays in Keras layouts (conv HWIO, dense
(in,out)), generated with Keras-default initialisers (Glorot-uniform kernels,
zero biases, BN gamma=1 beta=0 mean=0 var=1).  ``randomize_bn`` perturbs the BN
statistics so that tests are sensitive to BN handling.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.
"""
import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-3
N_POLICY = 1968
IN_PLANES = 127


def _glorot(rng, shape, fan_in, fan_out):
    lim = np.sqrt(6.0 / (fan_in + fan_out))
    return rng.uniform(-lim, lim, size=shape).astype(np.float32)


def init_weights(blocks, filters, seed=0, randomize_bn=False):
    """Keras-default random init of the tower, as a flat name -> ndarray dict."""
    rng = np.random.default_rng(seed)
    w = {}

    def conv(name, k, cin, cout):
        w[name + ".kernel"] = _glorot(rng, (k, k, cin, cout), k * k * cin, k * k * cout)
        w[name + ".bias"] = np.zeros(cout, np.float32)

    def bn(name, c):
        if randomize_bn:
            w[name + ".gamma"] = rng.uniform(0.5, 1.5, c).astype(np.float32)
            w[name + ".beta"] = rng.uniform(-0.2, 0.2, c).astype(np.float32)
            w[name + ".mean"] = rng.uniform(-0.2, 0.2, c).astype(np.float32)
            w[name + ".var"] = rng.uniform(0.5, 1.5, c).astype(np.float32)
        else:
            w[name + ".gamma"] = np.ones(c, np.float32)
            w[name + ".beta"] = np.zeros(c, np.float32)
            w[name + ".mean"] = np.zeros(c, np.float32)
            w[name + ".var"] = np.ones(c, np.float32)

    def dense(name, cin, cout):
        w[name + ".kernel"] = _glorot(rng, (cin, cout), cin, cout)
        w[name + ".bias"] = np.zeros(cout, np.float32)

    conv("stem", 3, IN_PLANES, filters)
    for i in range(blocks):
        conv("block%d.conv1" % i, 3, filters, filters)
        bn("block%d.bn1" % i, filters)
        conv("block%d.conv2" % i, 3, filters, filters)
        bn("block%d.bn2" % i, filters)
    conv("policy.conv", 1, filters, 2)
    bn("policy.bn", 2)
    dense("policy.dense", 128, N_POLICY)
    conv("value.conv", 1, filters, 1)
    bn("value.bn", 1)
    dense("value.dense1", 64, 256)
    dense("value.dense2", 256, 1)
    w["meta.blocks"] = np.array(blocks)
    w["meta.filters"] = np.array(filters)
    return w


@torch.no_grad()
def calibrated_weights(blocks, filters, planes, seed=0, policy_logit_std=3.0, value_preact_std=1.0):
    """Weights that make the tower's outputs SENSITIVE to its input, the way a trained tower's are
    (a random-init or barely trained tower is nearly constant: uniform policy, value ~ 0, so a small
    absolute output error proves nothing).  Data-dependent initialisation over ``planes`` (B,8,8,127),
    a sample of real positions: random kernels and biases, and per BatchNorm layer

      * moving mean / variance := the statistics of that layer's input over the sample, so every
        normalised activation has zero mean and unit variance -- each layer has unit gain, errors
        neither die out (contractive net) nor blow up (random statistics);
      * gamma in [0.6, 1.4], beta in [-0.4, 0.4] at random: channels differ, ReLUs cut at different
        points;

    then the policy Dense kernel is scaled so that the logits have standard deviation
    ``policy_logit_std`` over the sample (peaked softmax: best moves of 0.3 and more) and the last
    value Dense so that the tanh pre-activation has standard deviation ``value_preact_std`` (values
    spread over (-1, 1))."""
    rng = np.random.default_rng(seed)
    w = init_weights(blocks, filters, seed=seed)
    x = torch.as_tensor(planes)[..., :IN_PLANES].to(torch.float32).permute(0, 3, 1, 2)

    def rand_bias(name):
        n = w[name + ".bias"].shape[0]
        w[name + ".bias"] = rng.normal(0, 0.05, n).astype(np.float32)

    def calibrate(name, z):
        c = z.shape[1]
        w[name + ".mean"] = z.mean(dim=(0, 2, 3)).numpy().astype(np.float32)
        w[name + ".var"] = np.maximum(z.var(dim=(0, 2, 3), unbiased=False).numpy(), 1e-4).astype(np.float32)
        w[name + ".gamma"] = rng.uniform(0.6, 1.4, c).astype(np.float32)
        w[name + ".beta"] = rng.uniform(-0.4, 0.4, c).astype(np.float32)
        return _bn(z, w, name)

    rand_bias("stem")
    x = _conv(x, w, "stem", 1)
    for i in range(blocks):
        rand_bias("block%d.conv1" % i)
        rand_bias("block%d.conv2" % i)
        y = F.relu(calibrate("block%d.bn1" % i, _conv(x, w, "block%d.conv1" % i, 1)))
        y = calibrate("block%d.bn2" % i, _conv(y, w, "block%d.conv2" % i, 1))
        x = F.relu(x + y)
    rand_bias("policy.conv")
    p = F.relu(calibrate("policy.bn", _conv(x, w, "policy.conv", 0)))
    p = p.permute(0, 2, 3, 1).reshape(p.shape[0], -1)
    w["policy.dense.bias"] = rng.normal(0, 1.0, N_POLICY).astype(np.float32)
    logits = p @ torch.from_numpy(w["policy.dense.kernel"])
    w["policy.dense.kernel"] = (w["policy.dense.kernel"] * (policy_logit_std / float(logits.std()))).astype(np.float32)
    rand_bias("value.conv")
    v = F.relu(calibrate("value.bn", _conv(x, w, "value.conv", 0)))
    v = v.permute(0, 2, 3, 1).reshape(v.shape[0], -1)
    w["value.dense1.bias"] = rng.normal(0, 0.1, 256).astype(np.float32)
    v = F.relu(v @ torch.from_numpy(w["value.dense1.kernel"]) + torch.from_numpy(w["value.dense1.bias"]))
    pre = v @ torch.from_numpy(w["value.dense2.kernel"])
    w["value.dense2.kernel"] = (w["value.dense2.kernel"] * (value_preact_std / float(pre.std()))).astype(np.float32)
    w["value.dense2.bias"] = np.array([-float(pre.mean()) * value_preact_std / float(pre.std())], np.float32)
    return w


def _conv(x, w, name, pad):
    k = torch.from_numpy(w[name + ".kernel"]).permute(3, 2, 0, 1)       # HWIO -> OIHW
    return F.conv2d(x, k, torch.from_numpy(w[name + ".bias"]), padding=pad)


def _bn(x, w, name):
    g, b = torch.from_numpy(w[name + ".gamma"]), torch.from_numpy(w[name + ".beta"])
    m, v = torch.from_numpy(w[name + ".mean"]), torch.from_numpy(w[name + ".var"])
    sh = (1, -1, 1, 1)
    return (x - m.view(sh)) / torch.sqrt(v.view(sh) + BN_EPS) * g.view(sh) + b.view(sh)


@torch.no_grad()
def forward(w, planes):
    """planes (B,8,8,127) any float dtype -> (policy (B,1968) f32, value (B,) f32)."""
    blocks = int(w["meta.blocks"])
    x = torch.as_tensor(planes)[..., :IN_PLANES].to(torch.float32).permute(0, 3, 1, 2)
    x = _conv(x, w, "stem", 1)
    for i in range(blocks):
        y = F.relu(_bn(_conv(x, w, "block%d.conv1" % i, 1), w, "block%d.bn1" % i))
        y = _bn(_conv(y, w, "block%d.conv2" % i, 1), w, "block%d.bn2" % i)
        x = F.relu(x + y)
    p = F.relu(_bn(_conv(x, w, "policy.conv", 0), w, "policy.bn"))
    p = p.permute(0, 2, 3, 1).reshape(p.shape[0], -1)                  # Keras Flatten on NHWC
    p = p @ torch.from_numpy(w["policy.dense.kernel"]) + torch.from_numpy(w["policy.dense.bias"])
    p = torch.softmax(p, dim=-1)
    v = F.relu(_bn(_conv(x, w, "value.conv", 0), w, "value.bn"))
    v = v.permute(0, 2, 3, 1).reshape(v.shape[0], -1)
    v = F.relu(v @ torch.from_numpy(w["value.dense1.kernel"]) + torch.from_numpy(w["value.dense1.bias"]))
    v = torch.tanh(v @ torch.from_numpy(w["value.dense2.kernel"]) + torch.from_numpy(w["value.dense2.bias"]))
    return p, v[:, 0]


class TowerNet(object):
    """net callable for OracleAgent: planes -> (policy, value), fp32 CPU."""

    def __init__(self, weights):
        self.w = weights

    def __call__(self, planes):
        return forward(self.w, planes)
