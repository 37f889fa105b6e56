"""Keras ``.h5`` weight files <-> the flat weight dict.

The reference saves / loads its tower with ``model.save_weights(path)`` / ``load_weights(path)``
(/root/reference/src/chessrl/model.py:77-81) into ``model-<v>.h5`` (selfplay.py:33-56).  Layout of
such a file (tf.keras ``hdf5_format.save_weights_to_hdf5_group``; *from recollection* -- TensorFlow
is absent here): root attributes ``layer_names``, ``backend``, ``keras_version``; one group per
layer with attribute ``weight_names`` (e.g. ``conv2d_3/kernel:0``) and the datasets at that path
below the layer group (``/conv2d_3/conv2d_3/kernel:0``); BatchNormalization stores gamma, beta,
moving_mean, moving_variance.  A full-model save (``model.save``) nests the same under
``model_weights``.  The container format is handled by chessrl_amd/h5lite.py.

Reading is shape-driven so it does not depend on Keras' auto-generated layer names: layers are
taken in ``layer_names`` order (the trunk is a chain, so that order is its topological order) and
classified by their weights -- 3x3 convs: stem (127 input planes) then conv1/conv2 of each block;
BatchNorms of F channels pair up with the block convs; the heads are recognised by shape
(1x1 conv to 2 / 1 channels, BN of 2 / 1 channels, Dense 128x1968, 64x256, 256x1).
Writing emits the names a fresh Keras session would generate for model.py:31-63.
"""
import numpy as np

from . import h5lite

N_POLICY = 1968


def _layers(tree):
    root = tree["model_weights"] if "model_weights" in tree else tree
    names = root.attrs.get("layer_names")
    if names is None:                                   # Keras splits long attributes into chunks
        parts, i = [], 0
        while "layer_names%d" % i in root.attrs:
            parts.append(root.attrs["layer_names%d" % i])
            i += 1
        if not parts:
            raise ValueError("not a Keras weight file: no layer_names attribute")
        names = np.concatenate(parts)
    out = []
    for raw in np.atleast_1d(names):
        lname = raw.decode("utf8") if isinstance(raw, bytes) else str(raw)
        g = root[lname]
        wn = g.attrs.get("weight_names")
        weights = {}
        for w in (np.atleast_1d(wn) if wn is not None else []):
            path = w.decode("utf8") if isinstance(w, bytes) else str(w)
            node = g
            for part in path.split("/"):
                node = node[part]
            short = path.split("/")[-1].split(":")[0]
            weights[short] = np.asarray(node, dtype=np.float32)
        if weights:
            out.append((lname, weights))
    return out


def load_keras_h5(path):
    """Keras weight file of the reference's tower -> flat ``name -> ndarray`` dict (model.py layout)."""
    layers = _layers(h5lite.read(path))
    convs3 = [w for _, w in layers if "kernel" in w and w["kernel"].ndim == 4 and w["kernel"].shape[0] == 3]
    convs1 = [w for _, w in layers if "kernel" in w and w["kernel"].ndim == 4 and w["kernel"].shape[0] == 1]
    bns = [w for _, w in layers if "gamma" in w]
    denses = [w for _, w in layers if "kernel" in w and w["kernel"].ndim == 2]
    if not convs3 or convs3[0]["kernel"].shape[2] != 127 or len(convs3) % 2 != 1:
        raise ValueError("unexpected trunk: %d 3x3 convolutions" % len(convs3))
    filters, blocks = convs3[0]["kernel"].shape[3], (len(convs3) - 1) // 2
    out = {}

    def conv(name, w):
        out[name + ".kernel"], out[name + ".bias"] = w["kernel"], w["bias"]

    def bn(name, w):
        out[name + ".gamma"], out[name + ".beta"] = w["gamma"], w["beta"]
        out[name + ".mean"], out[name + ".var"] = w["moving_mean"], w["moving_variance"]

    conv("stem", convs3[0])
    trunk_bns = [w for w in bns if w["gamma"].shape[0] == filters]
    if len(trunk_bns) != 2 * blocks:
        raise ValueError("expected %d trunk BatchNorm layers, found %d" % (2 * blocks, len(trunk_bns)))
    for i in range(blocks):
        conv("block%d.conv1" % i, convs3[1 + 2 * i])
        bn("block%d.bn1" % i, trunk_bns[2 * i])
        conv("block%d.conv2" % i, convs3[2 + 2 * i])
        bn("block%d.bn2" % i, trunk_bns[2 * i + 1])

    def one(items, pred, what):
        hit = [w for w in items if pred(w)]
        if len(hit) != 1:
            raise ValueError("expected exactly one %s, found %d" % (what, len(hit)))
        return hit[0]
    conv("policy.conv", one(convs1, lambda w: w["kernel"].shape[3] == 2, "policy 1x1 conv"))
    conv("value.conv", one(convs1, lambda w: w["kernel"].shape[3] == 1, "value 1x1 conv"))
    bn("policy.bn", one(bns, lambda w: w["gamma"].shape[0] == 2 and filters != 2, "policy BatchNorm"))
    bn("value.bn", one(bns, lambda w: w["gamma"].shape[0] == 1 and filters != 1, "value BatchNorm"))
    conv("policy.dense", one(denses, lambda w: w["kernel"].shape == (128, N_POLICY), "policy Dense"))
    conv("value.dense1", one(denses, lambda w: w["kernel"].shape[0] == 64 and w["kernel"].shape[1] != 1,
                             "value hidden Dense"))
    conv("value.dense2", one(denses, lambda w: w["kernel"].shape[1] == 1, "value output Dense"))
    out["meta.blocks"], out["meta.filters"] = np.array(blocks), np.array(filters)
    return out


def save_keras_h5(weights, path):
    """Flat weight dict -> Keras-layout ``.h5`` (``load_weights`` of model.py:77-78 reads it back)."""
    blocks, filters = int(weights["meta.blocks"]), int(weights["meta.filters"])
    root = h5lite.Group()
    order = []
    counters = {}

    def auto(kind):                                     # conv2d, conv2d_1, conv2d_2, ...
        n = counters.get(kind, 0)
        counters[kind] = n + 1
        return kind if n == 0 else "%s_%d" % (kind, n)

    def layer(lname, pairs):
        g = h5lite.Group()
        inner = h5lite.Group()
        for short, arr in pairs:
            inner[short + ":0"] = np.asarray(arr, dtype=np.float32)
        if pairs:
            g[lname] = inner
        g.attrs["weight_names"] = (np.array([("%s/%s:0" % (lname, s)).encode("utf8") for s, _ in pairs])
                                   if pairs else np.zeros(0, "S1"))
        root[lname] = g
        order.append(lname)

    def conv(src):
        return [("kernel", weights[src + ".kernel"]), ("bias", weights[src + ".bias"])]

    def bn(src):
        return [("gamma", weights[src + ".gamma"]), ("beta", weights[src + ".beta"]),
                ("moving_mean", weights[src + ".mean"]), ("moving_variance", weights[src + ".var"])]

    # creation order of model.py:31-63 fixes the auto-generated names ...
    n_in, n_stem = "input_1", auto("conv2d")
    trunk = []
    for i in range(blocks):
        c1, b1, a1 = auto("conv2d"), auto("batch_normalization"), auto("activation")
        c2, b2, ad, a2 = auto("conv2d"), auto("batch_normalization"), auto("add"), auto("activation")
        trunk += [(c1, conv("block%d.conv1" % i)), (b1, bn("block%d.bn1" % i)), (a1, []),
                  (c2, conv("block%d.conv2" % i)), (b2, bn("block%d.bn2" % i)), (ad, []), (a2, [])]
    pc, pb, pa, pf = auto("conv2d"), auto("batch_normalization"), auto("activation"), auto("flatten")
    vc, vb, va, vf, vd = (auto("conv2d"), auto("batch_normalization"), auto("activation"),
                          auto("flatten"), auto("dense"))
    # ... and the layer order is the network's topological order: by decreasing distance from the
    # outputs, ties in traversal order from [policy_out, value_out] (the value head is one layer
    # longer, so each of its layers comes one step earlier than the policy head's)
    layer(n_in, [])
    layer(n_stem, conv("stem"))
    for lname, pairs in trunk:
        layer(lname, pairs)
    layer(vc, conv("value.conv"))
    layer(pc, conv("policy.conv"))
    layer(vb, bn("value.bn"))
    layer(pb, bn("policy.bn"))
    layer(va, [])
    layer(pa, [])
    layer(vf, [])
    layer(pf, [])
    layer(vd, conv("value.dense1"))
    layer("policy_out", conv("policy.dense"))
    layer("value_out", conv("value.dense2"))
    root.attrs["layer_names"] = np.array([n.encode("utf8") for n in order])
    root.attrs["backend"] = b"tensorflow"
    root.attrs["keras_version"] = b"2.2.4-tf"
    h5lite.write(path, root)
