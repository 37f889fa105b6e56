"""Keras ``.h5`` weight files without h5py.

The container layer (chessrl_amd/h5lite.py) is pinned both ways against the real HDF5 library:
reading a fixture that library wrote (tests/golden/keras_layout_libhdf5.h5, generator committed
next to it) and -- where the library's ``h5dump`` tool exists -- having it read files written here.
The Keras naming layer (chessrl_amd/keras_h5.py) is from recollection of tf.keras: parity unpinned.
"""
import os
import shutil
import subprocess

import numpy as np
import pytest

from chessrl_amd import h5lite
from chessrl_amd.keras_h5 import load_keras_h5, save_keras_h5
from chessrl_amd.model import init_weights

H5DUMP = shutil.which("h5dump") or ("/opt/conda/bin/h5dump" if os.path.exists("/opt/conda/bin/h5dump") else None)


def test_reads_a_file_written_by_the_real_hdf5_library(golden_dir):
    t = h5lite.read(os.path.join(golden_dir, "keras_layout_libhdf5.h5"))
    assert list(t.attrs["layer_names"]) == [b"input_1", b"conv2d", b"batch_normalization", b"activation",
                                            b"policy_out"]
    assert t.attrs["backend"] == b"tensorflow" and t.attrs["keras_version"] == b"2.2.4-tf"
    assert sorted(t) == ["activation", "batch_normalization", "conv2d", "input_1", "policy_out"]
    assert t["activation"].attrs["weight_names"].shape == (0,) and len(t["activation"]) == 0
    assert list(t["conv2d"].attrs["weight_names"]) == [b"conv2d/kernel:0", b"conv2d/bias:0"]
    k = t["conv2d"]["conv2d"]["kernel:0"]
    assert k.shape == (3, 3, 2, 4) and k.dtype == np.float32
    assert np.array_equal(k.ravel(), 1.0 + 0.25 * np.arange(72, dtype=np.float32))
    assert np.array_equal(t["conv2d"]["conv2d"]["bias:0"], [-2.0, -1.75, -1.5, -1.25])
    bn = t["batch_normalization"]["batch_normalization"]
    assert [float(bn[n][0]) for n in ("gamma:0", "beta:0", "moving_mean:0", "moving_variance:0")] == [10, 20, 30, 40]
    assert t["policy_out"]["policy_out"]["kernel:0"].shape == (8, 5)
    assert float(t["policy_out"]["policy_out"]["bias:0"][4]) == 201.0


def _same(a, b):
    assert set(a) == set(b) and set(a.attrs) == set(b.attrs)
    for k in a.attrs:
        assert np.array_equal(a.attrs[k], b.attrs[k]), k
    for k in a:
        if isinstance(a[k], dict):
            _same(a[k], b[k])
        else:
            x, y = np.asarray(a[k]), np.asarray(b[k])
            assert x.dtype == y.dtype and x.shape == y.shape and np.array_equal(x, y), k


def test_container_roundtrip_and_many_members(tmp_path, golden_dir):
    t = h5lite.read(os.path.join(golden_dir, "keras_layout_libhdf5.h5"))
    p = str(tmp_path / "copy.h5")
    h5lite.write(p, t)
    _same(t, h5lite.read(p))
    g = h5lite.Group()                                  # 70 members: several symbol nodes
    for i in range(70):
        sub = h5lite.Group()
        sub["kernel:0"] = np.arange(6, dtype=np.float32).reshape(2, 3) + i
        sub.attrs["weight_names"] = np.array([b"x%d/kernel:0" % i])
        g["x%d" % i] = sub
    g["scalar"] = np.float32(3.5)
    g["ints"] = np.arange(5, dtype=np.int64)
    g["f64"] = np.linspace(0, 1, 7)
    g.attrs["layer_names"] = np.array([b"x%d" % i for i in range(70)])
    g.attrs["rate"] = np.float32(2.5)
    p = str(tmp_path / "many.h5")
    h5lite.write(p, g)
    _same(g, h5lite.read(p))


def test_rejects_what_it_cannot_read(tmp_path):
    p = tmp_path / "x.h5"
    p.write_bytes(b"not an hdf5 file" * 100)
    with pytest.raises(h5lite.H5Error):
        h5lite.read(str(p))
    sb = bytearray(h5lite.SIGNATURE + bytes([2]) + bytes(87))       # superblock version 2
    p.write_bytes(bytes(sb))
    with pytest.raises(h5lite.H5Error):
        h5lite.read(str(p))


def test_tower_weights_roundtrip_through_keras_layout(tmp_path):
    w = init_weights(2, 16, seed=5)
    rng = np.random.default_rng(1)
    for k in w:                                          # make every tensor distinguishable
        if not k.startswith("meta."):
            w[k] = (w[k] + rng.normal(size=w[k].shape)).astype(np.float32)
    p = str(tmp_path / "model-0.h5")
    save_keras_h5(w, p)
    back = load_keras_h5(p)
    assert set(back) == set(w)
    for k in w:
        assert np.array_equal(back[k], w[k]), k
    t = h5lite.read(p)
    names = [n.decode() for n in t.attrs["layer_names"]]
    assert names[:3] == ["input_1", "conv2d", "conv2d_1"] and names[-2:] == ["policy_out", "value_out"]
    assert len(names) == 2 + 7 * 2 + 11 and "dense" in names and "add_1" in names
    assert t["conv2d_2"]["conv2d_2"]["kernel:0"].shape == (3, 3, 16, 16)
    # a full-model save nests the same groups under model_weights
    nested = h5lite.Group()
    nested["model_weights"] = t
    p2 = str(tmp_path / "full.h5")
    h5lite.write(p2, nested)
    assert np.array_equal(load_keras_h5(p2)["value.dense2.kernel"], w["value.dense2.kernel"])


@pytest.mark.skipif(H5DUMP is None, reason="HDF5 command-line tools not installed")
def test_real_hdf5_library_reads_files_written_here(tmp_path):
    w = init_weights(1, 8, seed=2)
    p = str(tmp_path / "model-0.h5")
    save_keras_h5(w, p)
    r = subprocess.run([H5DUMP, "-H", p], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "error" not in r.stderr.lower(), r.stderr[-500:]
    assert r.stdout.count("DATASET") == 2 * 5 + 4 * 4 + 2 * 3          # 5 convs, 4 BNs, 3 denses
    assert 'ATTRIBUTE "layer_names"' in r.stdout and 'GROUP "batch_normalization_3"' in r.stdout
    r = subprocess.run([H5DUMP, "-d", "/conv2d/conv2d/bias:0", "-d", "/value_out/value_out/kernel:0", p],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-500:]
    assert "DATASPACE  SIMPLE { ( 8 ) / ( 8 ) }" in r.stdout and "DATASPACE  SIMPLE { ( 256, 1 ) / ( 256, 1 ) }" in r.stdout
    first = float(np.asarray(w["value.dense2.kernel"]).ravel()[0])
    assert ("%.6g" % first)[:6] in r.stdout


def test_model_path_prefers_the_newest_version(tmp_path):
    from chessrl_amd.selfplay import get_model_path
    d = str(tmp_path)
    assert get_model_path(d).endswith("model-0.npz")
    open(os.path.join(d, "model-0.h5"), "wb").close()
    open(os.path.join(d, "model-1.h5"), "wb").close()
    assert get_model_path(d).endswith("model-1.h5")


@pytest.mark.gpu
def test_chess_model_saves_and_loads_keras_h5(tmp_path):
    import torch
    from chessrl_amd.model import ChessModel
    m = ChessModel(blocks=2, filters=64, seed=3)
    p = str(tmp_path / "model-0.h5")
    m.save_weights(p)
    m2 = ChessModel(weights=p)
    assert (m2.blocks, m2.filters) == (2, 64)
    x = torch.zeros((4, 8, 8, 128), dtype=torch.float16, device="cuda:0")
    x[..., :127] = (torch.rand((4, 8, 8, 127), device="cuda:0") < 0.1).half()
    (p1, v1), (p2, v2) = m(x), m2(x)
    assert torch.equal(p1, p2) and torch.equal(v1, v2)
    m3 = ChessModel(blocks=2, filters=64, seed=9)
    m3.load_weights(p)
    assert torch.equal(m3(x)[0], p1)


def test_container_roundtrip_property():
    """hypothesis: random trees (nesting, names, dtypes, shapes incl. scalars and empty arrays,
    byte-string and numeric attributes) survive write -> read unchanged."""
    import tempfile
    from hypothesis import HealthCheck, given, settings, strategies as st
    from hypothesis.extra import numpy as hnp

    names = st.text(alphabet="abcdefghijklmnopqrstuvwxyz_0123456789:", min_size=1, max_size=24)
    dtypes = st.sampled_from([np.float32, np.float64, np.int32, np.int64, np.uint8, np.float16])
    arrays = dtypes.flatmap(lambda d: hnp.arrays(d, hnp.array_shapes(min_dims=0, max_dims=3, min_side=0, max_side=5),
                                                 elements=st.integers(0, 100)))
    attr_vals = st.one_of(st.binary(min_size=1, max_size=12).filter(lambda b: b"\0" not in b),
                          arrays.filter(lambda a: a.size > 0),
                          st.lists(st.binary(min_size=1, max_size=9).filter(lambda b: b"\0" not in b),
                                   min_size=1, max_size=5).map(lambda l: np.array(l)))

    def groups(depth):
        leaf = st.dictionaries(names, arrays, max_size=4)
        if depth == 0:
            return leaf
        return st.dictionaries(names, st.one_of(arrays, groups(depth - 1)), max_size=4)

    def build(d, attrs):
        g = h5lite.Group()
        for k, v in d.items():
            g[k] = build(v, {}) if isinstance(v, dict) else v
        g.attrs.update(attrs)
        return g

    @settings(max_examples=40, deadline=None, derandomize=True, database=None,
              suppress_health_check=list(HealthCheck))
    @given(groups(2), st.dictionaries(names, attr_vals, max_size=3))
    def run(tree, attrs):
        g = build(tree, attrs)
        with tempfile.TemporaryDirectory() as d:
            p = os.path.join(d, "t.h5")
            h5lite.write(p, g)
            _same(g, h5lite.read(p))
            if H5DUMP is not None and len(tree) % 3 == 0:          # the real library parses it too
                r = subprocess.run([H5DUMP, p], capture_output=True, text=True, timeout=60)
                assert r.returncode == 0 and "unable" not in r.stderr.lower(), r.stderr[-300:]

    run()
