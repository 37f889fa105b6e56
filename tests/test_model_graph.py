"""CPU: the topology of the tower against the reference's own constructor.

tests/golden/model_graph.json is what ``ChessModel.__init__`` / ``__res_block`` (model.py:17-72,
111-122) build when they are executed over recording stand-ins of the Keras layer constructors
(oracle/ref_loader.record_model_graph; made by oracle/make_golden.py): every layer, its arguments
and its inputs, and the ``compile`` call.  Here that graph is (a) checked for the facts the build
relies on and (b) EXECUTED, node by node, with the oracle tower's weights assigned in layer-creation
order -- the order Keras also names ``.h5`` groups in -- and compared with ``tower_oracle.forward``.
What a node computes (Conv2D 'same' / 'valid', BatchNormalization(axis=-1) at inference with Keras'
default epsilon 1e-3, Flatten of an NHWC tensor, Dense + activation) is Keras documentation; which
nodes exist, with which arguments, wired how, is the reference's code.
"""
import json
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import tower_oracle


@pytest.fixture(scope="module")
def graph(golden_dir):
    return json.load(open(os.path.join(golden_dir, "model_graph.json")))["graph"]


def test_the_reference_graph_is_the_topology_the_build_assumes(graph):
    ops = [n["op"] for n in graph["nodes"]]
    assert ops.count("Conv2D") == 23 and ops.count("BatchNormalization") == 22 and ops.count("Add") == 10
    assert graph["nodes"][0] == {"op": "Input", "config": {"shape": [8, 8, 127]}, "inputs": []}
    stem = graph["nodes"][1]
    assert stem["op"] == "Conv2D" and stem["config"]["filters"] == 256 and stem["config"]["padding"] == "same"
    # the stem is linear: its output feeds the first block's conv AND its Add directly
    users = [n["op"] for n in graph["nodes"] if 1 in n["inputs"]]
    assert sorted(users) == ["Add", "Conv2D"]
    for n in graph["nodes"]:
        if n["op"] in ("Conv2D", "Dense"):
            assert n["config"]["kernel_regularizer"] == "l2"           # every kernel, no bias term
            assert "use_bias" not in n["config"]                       # Keras default: biases everywhere
        if n["op"] == "BatchNormalization":
            assert n["config"] == {"axis": -1}                         # all other arguments: Keras defaults
    pol, val = (graph["nodes"][i] for i in graph["outputs"])
    assert pol["config"]["args"] == [1968] and pol["config"]["activation"] == "softmax"
    assert val["config"]["args"] == [1] and val["config"]["activation"] == "tanh"
    c = graph["compile"]
    assert c["optimizer"] == {"class": "Adam", "lr": 0.002}
    assert c["loss"] == ["categorical_crossentropy", "mean_squared_error"]     # policy, value; unit weights


def _run_graph(graph, w, planes):
    """Execute the recorded graph on NHWC planes with the oracle's weights in creation order."""
    names = {"Conv2D": ["stem"], "BatchNormalization": [], "Dense": ["policy.dense", "value.dense1", "value.dense2"]}
    blocks = int(w["meta.blocks"])
    for i in range(blocks):
        names["Conv2D"] += ["block%d.conv1" % i, "block%d.conv2" % i]
        names["BatchNormalization"] += ["block%d.bn1" % i, "block%d.bn2" % i]
    names["Conv2D"] += ["policy.conv", "value.conv"]
    names["BatchNormalization"] += ["policy.bn", "value.bn"]
    seen = {k: 0 for k in names}
    vals = []
    for n in graph["nodes"]:
        op, cfg = n["op"], n["config"]
        x = [vals[i] for i in n["inputs"]]
        if op in names:
            name = names[op][seen[op]]
            seen[op] += 1
        if op == "Input":
            y = planes                                                   # [B, 8, 8, 127]
        elif op == "Conv2D":
            k = torch.from_numpy(w[name + ".kernel"])                   # HWIO
            assert k.shape[0] == cfg["kernel_size"] and k.shape[3] == cfg["filters"] and cfg.get("strides", 1) == 1
            pad = {"same": cfg["kernel_size"] // 2, "valid": 0}[cfg["padding"]]
            y = F.conv2d(x[0].permute(0, 3, 1, 2), k.permute(3, 2, 0, 1), torch.from_numpy(w[name + ".bias"]),
                         padding=pad).permute(0, 2, 3, 1)
        elif op == "BatchNormalization":
            g, b, m, v = (torch.from_numpy(w[name + s]) for s in (".gamma", ".beta", ".mean", ".var"))
            y = (x[0] - m) / torch.sqrt(v + 1e-3) * g + b
        elif op == "Activation":
            assert cfg["args"] == ["relu"]
            y = torch.relu(x[0])
        elif op == "Add":
            y = x[0] + x[1]
        elif op == "Flatten":
            y = x[0].reshape(x[0].shape[0], -1)                          # NHWC memory order, as Keras
        elif op == "Dense":
            k = torch.from_numpy(w[name + ".kernel"])
            assert k.shape[1] == cfg["args"][0]
            y = x[0] @ k + torch.from_numpy(w[name + ".bias"])
            y = {"softmax": lambda t: torch.softmax(t, -1), "relu": torch.relu, "tanh": torch.tanh}[cfg["activation"]](y)
        else:
            raise AssertionError(op)
        vals.append(y)
    assert all(seen[k] == len(names[k]) for k in names)                 # every weight found its layer
    return [vals[i] for i in graph["outputs"]]


@pytest.mark.parametrize("randomize_bn", [False, True])
def test_executing_the_reference_graph_gives_the_oracle_tower(graph, randomize_bn):
    """The reference model is 10 blocks x 256 filters: run it at that size (two boards)."""
    w = tower_oracle.init_weights(10, 256, seed=5, randomize_bn=randomize_bn)
    rng = np.random.default_rng(9)
    w["policy.dense.bias"] = rng.normal(0, 0.1, 1968).astype(np.float32)
    w["value.dense2.bias"] = np.array([0.05], np.float32)
    planes = torch.from_numpy((rng.random((2, 8, 8, 127)) < 0.12).astype(np.float32))
    with torch.no_grad():
        pol, val = _run_graph(graph, w, planes)
        ref_pol, ref_val = tower_oracle.forward(w, planes)
    assert pol.shape == (2, 1968) and val.shape == (2, 1)
    assert (pol - torch.as_tensor(ref_pol)).abs().max().item() <= 1e-6
    assert (val.reshape(-1) - torch.as_tensor(ref_val).reshape(-1)).abs().max().item() <= 1e-6


def test_keras_h5_group_names_follow_the_reference_creation_order(graph, tmp_path):
    """Keras names un-named layers by class and creation order (conv2d, conv2d_1, ...,
    batch_normalization_3, dense; 'policy_out' / 'value_out' are given in model.py:47,60).  Derive
    those names from the recorded graph and check that the ``.h5`` the product writes
    (chessrl_amd/keras_h5.py, no GPU involved) stores every weight under the group of ITS layer."""
    from chessrl_amd import h5lite
    from chessrl_amd.keras_h5 import save_keras_h5
    w = tower_oracle.init_weights(10, 256, seed=2, randomize_bn=True)
    path = str(tmp_path / "model.h5")
    save_keras_h5(w, path)
    tree = h5lite.read(path)
    src = {"Conv2D": ["stem"], "BatchNormalization": [], "Dense": ["policy.dense", "value.dense1", "value.dense2"]}
    for i in range(10):
        src["Conv2D"] += ["block%d.conv1" % i, "block%d.conv2" % i]
        src["BatchNormalization"] += ["block%d.bn1" % i, "block%d.bn2" % i]
    src["Conv2D"] += ["policy.conv", "value.conv"]
    src["BatchNormalization"] += ["policy.bn", "value.bn"]
    snake = {"Conv2D": "conv2d", "BatchNormalization": "batch_normalization", "Dense": "dense"}
    count = {k: 0 for k in snake}
    used = {k: 0 for k in snake}
    checked = 0
    for n in graph["nodes"]:
        if n["op"] not in snake:
            continue
        k = count[n["op"]]
        count[n["op"]] += 1
        if "name" in n["config"]:
            lname = n["config"]["name"]
        else:                                            # auto names count the UN-named layers of the class
            lname = snake[n["op"]] if used[n["op"]] == 0 else "%s_%d" % (snake[n["op"]], used[n["op"]])
            used[n["op"]] += 1
        inner = tree[lname][lname]
        first = "gamma" if n["op"] == "BatchNormalization" else "kernel"
        assert np.array_equal(np.asarray(inner[first + ":0"]), w[src[n["op"]][k] + "." + first]), (lname, src[n["op"]][k])
        checked += 1
    assert checked == 23 + 22 + 3
