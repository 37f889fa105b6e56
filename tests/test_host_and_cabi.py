"""CPU: the C-ABI library loads and exports every declared symbol, fails loudly without a GPU;
host-side logic (labels, compute_policy, records, conversions, tower BN folding)."""
import json
import os
import re

import numpy as np
import pytest
import torch

from oracle import encoder_oracle, mcts_oracle, tower_oracle
from oracle.chess_oracle import board_from_fen, board_to_array
from oracle import chess_oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HAS_GPU = torch.cuda.is_available()


def test_library_exports_every_symbol_declared_in_the_header():
    from chessrl_amd import _lib
    text = open(os.path.join(ROOT, "include", "chessrl_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(crl_[a-z_0-9]+)\s*\(", text)))
    assert declared and sorted(_lib.SYMBOLS) == declared
    L = _lib.lib()
    for name in declared:
        assert hasattr(L, name), name


def test_library_identity_and_refusal_of_a_stale_library(monkeypatch):
    """The library carries the sha256 of the sources it was built from and an ABI version; a library built
    from other sources that cannot be rebuilt is refused, and so is one of another ABI version (a ctypes
    call through a stale signature would hand the GPU garbage pointers)."""
    from chessrl_amd import _lib
    L = _lib.lib()
    text = open(os.path.join(ROOT, "include", "chessrl_hip.h")).read()
    assert L.crl_abi_version() == _lib.ABI_VERSION == int(re.search(r"#define CRL_ABI_VERSION (\d+)", text).group(1))
    assert L.crl_source_hash().decode() == _lib.source_hash() == _lib.embedded_hash() and not _lib.is_stale()
    # the sources change and hipcc is not there: no silent load of the old library
    monkeypatch.setattr(_lib, "source_hash", lambda: "0" * 64)
    assert _lib.is_stale()
    monkeypatch.setattr(_lib, "_lib", None)

    def no_compiler(*a, **k):
        raise OSError("hipcc: not found")
    monkeypatch.setattr(_lib, "build", no_compiler)
    with pytest.raises(_lib.HipLibraryError, match="older than csrc"):
        _lib.lib()
    # another ABI version than the binding's
    monkeypatch.undo()
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "ABI_VERSION", _lib.ABI_VERSION + 1)
    with pytest.raises(_lib.HipLibraryError, match="ABI version"):
        _lib.lib()
    monkeypatch.undo()
    assert _lib.lib() is not None


@pytest.mark.skipif(HAS_GPU, reason="checks the no-GPU failure path")
def test_product_path_fails_loudly_without_gpu():
    from chessrl_amd import _lib
    from chessrl_amd.engine import LockstepEngine
    from chessrl_amd.model import ChessModel
    with pytest.raises(_lib.HipLibraryError, match="no HIP device|crl_create failed"):
        _lib.Context(4, 8)
    with pytest.raises(_lib.HipLibraryError):
        LockstepEngine(lambda p: None, 4, 8)
    with pytest.raises(RuntimeError):
        ChessModel(blocks=1, filters=8)


def test_product_never_imports_the_oracle():
    """A product path that routes through oracle/ would void every parity claim."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "chessrl_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), f
                assert "oracle/" not in src or f.endswith(".md"), f


def test_label_table_matches_reference_golden(golden_dir):
    from chessrl_amd import _lib
    from chessrl_amd.game import move_to_uci
    import hashlib
    gold = json.load(open(os.path.join(golden_dir, "uci_labels.json")))
    assert len(gold["labels"]) == 1968 and len(set(gold["labels"])) == 1968
    assert gold["sha256"] == "e67a413cdbce60252cbcf4714d6e4b88549ecaee5e5a613ee4cdb8d5098d8f7b"
    assert hashlib.sha256("\n".join(gold["labels"]).encode()).hexdigest() == gold["sha256"]
    assert [move_to_uci(m) for m in _lib.uci_label_moves()] == gold["labels"]     # C++ generator
    assert encoder_oracle.get_uci_labels() == gold["labels"]                      # oracle restatement
    for u, i in (("e2e4", 930), ("g1f3", 1402), ("e1g1", 901), ("e7e8q", 1881), ("h2h1b", 1960)):
        assert gold["labels"][i] == u


def test_uci_and_fen_conversions_agree_with_oracle():
    from chessrl_amd import game
    for u in ["e2e4", "a7a8q", "h2g1n", "e1g1", "b7c8r"]:
        assert game.uci_to_move(u) == chess_oracle.uci_to_move(u)
        assert game.move_to_uci(game.uci_to_move(u)) == u
    for bad in ["00000", "e2", "e2e9", "i1a1", "e7e8k", "e7e8x", None, 5]:
        assert game.uci_to_move(bad) is None
    fen = "r3k2r/p1ppqpb1/bn2pnp1/3PN3/1p2P3/2N2Q1p/PPPBBPPP/R3K2R w KQkq e3 12 30"
    row = game.board_row_from_fen(fen)
    assert np.array_equal(row, board_to_array(board_from_fen(fen)))
    assert game.board_fen_from_row(row) == fen.split()[0]
    # FEN roots carry python-chess's clean_castling_rights() (game.py:17-21 -> chess.Board(fen))
    from tests.test_oracle_chess import UNCLEAN_FENS, _rights
    for fen, want in UNCLEAN_FENS:
        row = game.board_row_from_fen(fen)
        assert _rights(int(row[7])) == want, fen
        assert np.array_equal(row, board_to_array(board_from_fen(fen)))


def test_compute_policy_matches_oracle_bitwise():
    from chessrl_amd.engine import compute_policy
    rng = np.random.default_rng(0)
    for nb in (0, 12, 29, 30, 31, 77, 200):
        v = rng.integers(0, 40, size=rng.integers(1, 60)).tolist()
        rv = sum(v) + 1
        a = compute_policy(v, rv, nb, noise=False)
        b = mcts_oracle.compute_policy(v, rv, nb, noise=False)
        assert np.array_equal(a, b)
        a = compute_policy(v, rv, nb, noise=True, rng=np.random.default_rng(5))
        b = mcts_oracle.compute_policy(v, rv, nb, noise=True, rng=np.random.default_rng(5))
        assert np.array_equal(a, b)
    np.random.seed(9)
    a = compute_policy([5, 2, 1], 9, 3, noise=True)
    np.random.seed(9)
    b = 0.75 * np.array([5, 2, 1]) / 9 + np.random.dirichlet([0.03] * 3)    # noise NOT scaled by eps
    assert np.array_equal(a, b)


def test_records_roundtrip():
    from chessrl_amd import records
    from chessrl_amd.game import uci_to_move
    recs = [records.GameRecord(7, [uci_to_move(u) for u in ["e2e4", "e7e5", "d1h5"]], None, True, "d"),
            records.GameRecord(2**33 + 1, [uci_to_move("a7a8q")], -1, False),
            records.GameRecord(0, [], 0, True)]
    back = records.unpack(records.pack(recs, 16))
    assert back == recs and back[1].game_id == 2**33 + 1
    h = json.loads(records.dumps(recs))
    assert h[0] == {"moves": ["e2e4", "e7e5", "d1h5"], "result": None, "player_color": True, "date": "d"}
    again = records.loads(records.dumps(recs))
    assert [r.get_history()["moves"] for r in again] == [r.get_history()["moves"] for r in recs]
    with pytest.raises(ValueError):
        records.pack(recs, 2)


def test_game_color_and_model_path(tmp_path):
    from chessrl_amd import selfplay
    cols = [selfplay.game_color(3, g) for g in range(64)]
    assert cols == [selfplay.game_color(3, g) for g in range(64)] and 10 < sum(cols) < 54
    assert selfplay.get_model_path(str(tmp_path)).endswith("model-0.npz")
    for v in (1, 3, 2):
        (tmp_path / ("model-%d.npz" % v)).write_bytes(b"")
    assert selfplay.get_model_path(str(tmp_path)).endswith("model-3.npz")


def test_tower_module_folding_matches_oracle_on_cpu():
    """BN folding, HWIO->OIHW, the 128-plane pad and Keras Flatten order, in fp32 on the CPU
    (the fp16 MFMA numerics are checked on the GPU)."""
    from chessrl_amd import model
    w = tower_oracle.init_weights(3, 16, seed=2, randomize_bn=True)
    net = model.Tower(3, 16)
    net.load_keras_dict(w)
    net = net.to(memory_format=torch.channels_last).eval()
    rng = np.random.default_rng(1)
    planes = (rng.random((5, 8, 8, 127)) < 0.15).astype(np.float32)
    x = torch.zeros((5, 8, 8, 128))
    x[..., :127] = torch.from_numpy(planes)
    with torch.no_grad():
        p, v = net(x.permute(0, 3, 1, 2))
    ep, ev = tower_oracle.forward(w, planes)
    assert (p - ep).abs().max() < 1e-6 and (v - ev).abs().max() < 1e-5
    assert abs(p.sum(dim=1) - 1).max() < 1e-5 and p.shape == (5, 1968)
    own = model.init_weights(3, 16, seed=0)
    assert set(own) == set(w) and all(own[k].shape == w[k].shape for k in w)
    f, b = 128, 10
    assert 73152 * f + 1152 * f * f * b + 192 * f + 268544 == 198400256


def test_encoder_oracle_known_answers():
    """Hand-derived: python-chess is absent, so these pin get_game_state."""
    from oracle.chess_oracle import OracleGame
    g = OracleGame()
    s = encoder_oracle.get_game_state(g)
    assert s.shape == (8, 8, 127)
    assert s[6, :, 8].all() and s[1, :, 1].all()                # white pawns row 6, black pawns row 1
    assert s[7, 4, 7 + 6] == 1 and s[0, 4, 6] == 1              # kings: e1 -> row 7 col 4, e8 -> row 0
    assert s[2:, :, 0].all() and not s[:2, :, 0].any()          # "no black piece here" plane
    assert s[:, :, 126].all() and not s[:, :, 14:126].any()
    g.move("e2e4")
    s2 = encoder_oracle.get_game_state(g)
    assert np.array_equal(s2[:, :, 14:28], s[:, :, 0:14])       # previous position in history slot 0
    assert not s2[:, :, 126].any() and s2[4, 4, 8] == 1 and s2[6, 4, 8] == 0
    assert np.array_equal(encoder_oracle.get_game_state(g, flipped=True), np.rot90(s2, k=2))


def test_choose_children_equals_per_game_compute_policy():
    """The batched move choice must pick exactly what argmax(compute_policy(...)) picks per game."""
    from chessrl_amd.engine import choose_children, compute_policy
    rng = np.random.default_rng(1)
    G = 200
    nchild = rng.integers(0, 45, size=G)
    nchild[:5] = 0
    visits = rng.integers(0, 30, size=(G, 256)).astype(np.int32)
    plies = rng.integers(0, 300, size=G)
    plies[:40] = rng.integers(25, 35, size=40)                 # around the tau switch at 30 plies
    root = np.array([visits[g, :nchild[g]].sum() + 1 for g in range(G)])
    for noise in (False, True):
        rngs = [np.random.default_rng([7, g]) for g in range(G)]
        got = choose_children(visits, nchild, root, plies, noise=noise, rngs=rngs)
        rngs = [np.random.default_rng([7, g]) for g in range(G)]
        for g in range(G):
            if nchild[g] == 0:
                assert got[g] == -1
                continue
            p = compute_policy(visits[g, :nchild[g]], root[g], int(plies[g]), noise=noise, rng=rngs[g])
            assert got[g] == int(np.argmax(p)), (g, noise)
            q = mcts_oracle.compute_policy(list(visits[g, :nchild[g]]), int(root[g]), int(plies[g]), noise=False)
            if not noise:
                assert got[g] == int(np.argmax(q))


def test_bench_refuses_to_run_without_a_gpu():
    """bench.py measures the HIP path only: on a box without a GPU it must exit with an error,
    not fall back to anything."""
    import subprocess
    import sys
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode != 0
    assert "MI355X" in (r.stderr + r.stdout) and "{" not in r.stdout


def test_bench_gpus_2_starts_two_ranks_and_checks_the_world(tmp_path):
    """``bench.py --gpus 2`` without a launcher starts two rank processes itself; the ranks join
    one process group (gloo here), prove the world size with an all_reduce and run the record
    gather.  CRL_BENCH_DRYRUN=1 stops before the GPU workload (nothing is measured: value null)."""
    import json
    import subprocess
    import sys
    env = dict(os.environ, CRL_BENCH_DRYRUN="1", CRL_BENCH_BACKEND="gloo")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3",
                        "--warmup", "1"], capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1                                  # rank 0 only
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["dry_run"] is True and out["value"] is None
    assert out["record_gather"] == {"records": 5 + 6, "backend": "gloo"}
    # the process group carries an EXPLICIT timeout (rank 0's post-window work runs while the others wait in a
    # collective): the default of --dist-timeout-min here, the argument's value in the 8-rank run below
    assert out["process_group_timeout_s"] == 30 * 60
    # SCALE-day shape: 8 ranks (gloo, no GPU): one line from rank 0, every rank's step and trunk times on it,
    # and the precision mode agreed by all ranks (one rank on f16x3 puts all eight there)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "3",
                        "--warmup", "1", "--dist-timeout-min", "7"], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and out["record_gather"]["records"] == sum(5 + k for k in range(8))
    assert out["mode_agreed"] == "f16x3" and out["process_group_timeout_s"] == 7 * 60
    pr = out["per_rank"]["ms_per_step"]
    assert len(pr["ranks"]) == 8 and pr["min"] == 2.0 and abs(pr["max"] - 2.07) < 1e-9 and pr["min"] < pr["mean"] < pr["max"]
    assert len(out["per_rank"]["trunk_launch_ms"]["ranks"]) == 8
    # a launcher that started a different number of ranks than --gpus says is refused
    env2 = dict(env, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29400")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"],
                       capture_output=True, text=True, timeout=300, cwd=ROOT, env=env2)
    assert r.returncode != 0 and "--gpus 2" in r.stderr and "{" not in r.stdout


def test_dense_head_weight_packing_is_the_fragment_order_the_kernel_reads():
    """ChessModel._pack_split -> [tile][k-step][hi|lo][lane = 16 q + r][8 halves] with the lane
    holding x[16 tile + r][32 s + 8 q + e] (csrc/heads.hpp), and hi + lo == x to 2^-22."""
    import torch
    from chessrl_amd.model import ChessModel
    g = torch.Generator().manual_seed(0)
    x = torch.randn((48, 64), generator=g) * 0.3                  # 3 tiles of 16 units, 2 k-steps
    packed = ChessModel._pack_split(x, 3, 2).reshape(3, 2, 2, 64, 8).float()
    for t in range(3):
        for s in range(2):
            for lane in (0, 5, 16, 37, 63):
                r, q = lane & 15, lane >> 4
                want = x[16 * t + r, 32 * s + 8 * q: 32 * s + 8 * q + 8]
                hi, lo = packed[t, s, 0, lane], packed[t, s, 1, lane]
                assert torch.equal(hi, want.half().float())
                assert (hi + lo - want).abs().max() <= want.abs().max() * 2.0 ** -21


def test_trunk_weight_image_is_the_plane_order_the_header_documents():
    """ChessModel._pack_fused (CPU tensors, no GPU): the fp16 image handed to crl_trunk_forward is
    [conv][tap][in-ch/32][F rows][4 chunks][8 in] with row r holding output channel
    (r & ~31) + 8*((r & 15) >> 2) + 4*((r >> 4) & 1) + (r & 3) and input channels 8c..8c+7 at chunk
    position c ^ ((-(r >> 2)) & 3) (include/chessrl_hip.h) -- checked element by element against the
    BatchNorm-folded kernels, stem (128 input planes incl. the zero pad) and residual convs."""
    import torch
    from chessrl_amd import model as M
    from oracle import tower_oracle
    F_, blocks = 64, 1
    w = tower_oracle.init_weights(blocks, F_, seed=3, randomize_bn=True)
    m = M.ChessModel.__new__(M.ChessModel)
    m.filters, m.blocks, m.device = F_, blocks, torch.device("cpu")
    m.precision_requested = "auto"                                   # packs both images
    m._pack_fused(w)
    img = m._wtiles.float().numpy()
    assert m._wtiles.dtype == torch.float16
    convs = [("stem", None), ("block0.conv1", "block0.bn1"), ("block0.conv2", "block0.bn2")]
    assert img.size == 9 * 128 * F_ + 2 * 9 * F_ * F_
    rng = np.random.default_rng(0)
    off = 0
    for conv, bn in convs:
        k, _ = M._fold(w, conv, bn)                                  # [O][I][ky][kx] fp32
        k = k.half().float().numpy()
        cin = 128 if conv == "stem" else F_
        planes = img[off:off + 9 * cin * F_].reshape(9, cin // 32, F_, 4, 8)
        for _ in range(400):
            tap, g, r, c, e = (int(rng.integers(n)) for n in (9, cin // 32, F_, 4, 8))
            chan = (r & ~31) + 8 * ((r & 15) >> 2) + 4 * ((r >> 4) & 1) + (r & 3)
            pos = c ^ ((-(r >> 2)) & 3)
            i = 32 * g + 8 * c + e
            want = k[chan, i, tap // 3, tap % 3] if i < k.shape[1] else 0.0    # stem: plane 127 is a zero pad
            assert planes[tap, g, r, pos, e] == want, (conv, tap, g, r, c, e)
        off += 9 * cin * F_
    assert sorted({(r & ~31) + 8 * ((r & 15) >> 2) + 4 * ((r >> 4) & 1) + (r & 3) for r in range(F_)}) == list(range(F_))
    # the split-precision image (CRL_TRUNK_SPLIT): per tap the planes of Whi, then of Wlo = fp16(W - Whi),
    # each in the same plane order; hi + lo carries W to ~2^-22
    img3 = m._wtiles3.float().numpy()
    assert img3.size == 2 * (9 * 128 * F_ + 2 * 9 * F_ * F_)
    off = 0
    for conv, bn in convs:
        k, _ = M._fold(w, conv, bn)
        k = k.numpy()
        hi = k.astype(np.float16).astype(np.float32)
        lo = (k - hi).astype(np.float16).astype(np.float32)
        cin = 128 if conv == "stem" else F_
        parts = [hi, lo]
        planes = img3[off:off + len(parts) * 9 * cin * F_].reshape(9, len(parts), cin // 32, F_, 4, 8)
        for _ in range(400):
            tap, part, g, r, c, e = (int(rng.integers(n)) for n in (9, len(parts), cin // 32, F_, 4, 8))
            chan = (r & ~31) + 8 * ((r & 15) >> 2) + 4 * ((r >> 4) & 1) + (r & 3)
            i = 32 * g + 8 * c + e
            want = parts[part][chan, i, tap // 3, tap % 3] if i < k.shape[1] else 0.0
            assert planes[tap, part, g, r, c ^ ((-(r >> 2)) & 3), e] == want, (conv, tap, part, g, r, c, e)
        assert np.abs(hi + lo - k).max() <= 2.0 ** -20 * np.abs(k).max()
        off += len(parts) * 9 * cin * F_
    # 256 filters: the split-precision image of the LAYER-WISE kernels (csrc/tower_layer.hpp), K-chunk-major:
    # [conv][in-ch/32][tap][Whi, Wlo][256 rows][4 chunks][8 in] (include/chessrl_hip.h: crl_trunk_workspace_bytes)
    F_ = 256
    w = tower_oracle.init_weights(1, F_, seed=5, randomize_bn=True)
    m = M.ChessModel.__new__(M.ChessModel)
    m.filters, m.blocks, m.device = F_, 1, torch.device("cpu")
    m.precision_requested = "f16x3"
    m._pack_fused(w)
    img3 = m._wtiles3.float().numpy()
    assert img3.size == 2 * (9 * 128 * F_ + 2 * 9 * F_ * F_)
    off = 0
    for conv, bn in convs:
        k = M._fold(w, conv, bn)[0].numpy()
        hi = k.astype(np.float16).astype(np.float32)
        parts = [hi, (k - hi).astype(np.float16).astype(np.float32)]
        cin = 128 if conv == "stem" else F_
        planes = img3[off:off + 2 * 9 * cin * F_].reshape(cin // 32, 9, 2, F_, 4, 8)
        for _ in range(400):
            g, tap, part, r, c, e = (int(rng.integers(n)) for n in (cin // 32, 9, 2, F_, 4, 8))
            chan = (r & ~31) + 8 * ((r & 15) >> 2) + 4 * ((r >> 4) & 1) + (r & 3)
            i = 32 * g + 8 * c + e
            want = parts[part][chan, i, tap // 3, tap % 3] if i < k.shape[1] else 0.0
            assert planes[g, tap, part, r, c ^ ((-(r >> 2)) & 3), e] == want, (conv, g, tap, part, r, c, e)
        off += 2 * 9 * cin * F_


@pytest.fixture(scope="module")
def device_asm(tmp_path_factory):
    """The gfx950 assembly of the library's device code, compiled with the library's own flags."""
    import shutil
    import subprocess
    from chessrl_amd import _lib
    if shutil.which("hipcc") is None:
        pytest.skip("hipcc not available")
    asm = str(tmp_path_factory.mktemp("isa") / "api.s")
    flags = [f for f in _lib.HIPCC_FLAGS if f not in ("-fPIC", "-shared")]
    subprocess.check_call(["hipcc"] + flags + ["-S", "--offload-device-only",
                                               os.path.join(ROOT, "chessrl_amd", "csrc", "api.hip"), "-o", asm],
                          stderr=subprocess.DEVNULL)
    return asm


def test_trunk_kernels_never_read_a_register_with_an_lds_read_in_flight(device_asm):
    """The trunk kernels issue their fragment reads as inline-asm ``ds_read_b128`` with hand-counted
    ``s_waitcnt lgkmcnt`` (hipcc would otherwise undo the software pipeline), so hipcc does not know
    that those registers are not valid yet and may copy them -- phi moves at a loop back-edge -- before
    the data has landed: results then depend on LDS latency (found in round 3: the 64-filter
    split-precision kernels differed from run to run once a second process shared the GPU).
    tools/check_asm_hazards.py walks the ISA of every k_trunk_x16 kernel the library contains and
    reports any instruction reading a register whose ds_read is still in flight."""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_asm_hazards.py"), device_asm],
                       capture_output=True, text=True)
    kernels = [l for l in r.stdout.splitlines() if l.startswith(("k_trunk_x16<", "k_layer_conv<"))]
    # every dispatched k_trunk_x16<F, NB, BITS, PAIR, GROUP, SPLIT, IDX> + the layer-wise k_layer_conv<CHUNKS, KIND, IDX>
    assert sum(l.startswith("k_trunk_x16<") for l in kernels) >= 18 and sum(l.startswith("k_layer_conv<") for l in kernels) >= 18, r.stdout
    assert r.returncode == 0 and all(l.endswith(": ok") for l in kernels), r.stdout


def _race_checker():
    import importlib
    import sys
    tools = os.path.join(ROOT, "tools")
    if tools not in sys.path:
        sys.path.insert(0, tools)
    return importlib.import_module("lds_race_check")


def test_trunk_kernels_lds_traffic_is_race_free_under_emulation(device_asm):
    """The other half of the hand-counted pipeline: weight and bias tiles arrive by LDS-DMA
    (``global_load_lds``) and are retired by counted ``s_waitcnt vmcnt(N)`` + ``s_barrier`` before a ring
    slot is read, and a slot is refilled only behind a barrier after its last reader.
    tools/lds_race_check.py EXECUTES all 8 waves of a workgroup of every dispatched trunk kernel (20:
    F x NB x plane format x PAIR / GROUP x SPLIT) on an emulator of the integer / control subset of the
    gfx950 ISA -- stem + 2 residual blocks: prologue, steady state, both parities of the layer hand-over,
    drain -- and checks every LDS byte epoch by epoch: no byte read or written while a DMA transfer into it
    is in flight, no two transfers into one byte, no byte written by one wave and touched by another without
    a barrier between, no instruction reading OR overwriting a register with an asm ds_read in flight."""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "lds_race_check.py"), device_asm, "2"],
                       capture_output=True, text=True)
    kernels = [l for l in r.stdout.splitlines() if l.startswith(("k_trunk_x16<", "k_layer_conv<"))]
    # (round 5: also the layer-wise convolution kernels of csrc/tower_layer.hpp, one whole convolution each: 72 taps,
    # the two-buffer activation chunks and the four-slot plane ring with its ONE barrier per tap)
    assert sum(l.startswith("k_trunk_x16<") for l in kernels) >= 18 and sum(l.startswith("k_layer_conv<") for l in kernels) >= 18, r.stdout + r.stderr
    assert r.returncode == 0 and all(": ok " in l for l in kernels), r.stdout
    for l in kernels:                                        # the emulation really ran the pipeline
        stats = eval(l.split(": ok ", 1)[1])
        assert stats["dma_transfers"] > 400 and stats["lds_reads"] > 3000 and stats["epochs"] >= 30, l


@pytest.mark.parametrize("kernel", ["128,4,1,0,1,0,0,0", "64,2,1,0,0,0,0,0", "64,4,1,0,0,1,0,0", "128,2,1,0,1,0,1,0",
                                    "layer:8,1,0,4", "layer:8,2,3,2", "layer:8,1,2,1"])
def test_lds_race_check_catches_seeded_pipeline_bugs(device_asm, kernel):
    """Negative controls of the emulation: loosening a steady-state ``vmcnt`` wait by one, removing a tile
    barrier and loosening a fragment ``lgkmcnt`` wait by one must each be reported (pair ring, plain ring,
    group ring, a split-precision kernel and the layer-wise convolution)."""
    chk = _race_checker()
    family = "k_layer_conv" if kernel.startswith("layer:") else "k_trunk_x16"
    kernel = kernel.split(":")[-1]
    seg = {(f, t): sg for f, t, sg in chk.kernels_of(device_asm)}[(family, kernel)]
    indexed = family == "k_layer_conv" and kernel.split(",")[2] != "0"     # a listed launch: one listed board + its padding
    kernarg = chk.layer_kernarg(int(kernel.split(",")[1]), indexed) if family == "k_layer_conv" else None
    listed = chk.layer_listed(kernel.split(",")[2]) if family == "k_layer_conv" else None   # (IDX 3: a list beyond 256 boards)

    def run(lines):
        ins, labels = chk.parse_kernel(lines)
        try:
            return chk.check_workgroup(ins, labels, 1, kernarg=kernarg, listed=listed)[0]
        except chk.EmuError as e:
            return ["stopped: %s" % e]

    assert run(seg) == []
    in_asm, vm, lg, bars = False, [], [], []
    for i, l in enumerate(seg):
        if "#ASMSTART" in l:
            in_asm = True
        elif "#ASMEND" in l:
            in_asm = False
        t = l.split(";")[0].strip()
        if in_asm and t.startswith("s_waitcnt"):
            (vm if "vmcnt" in t else lg).append(i)
        if t == "s_barrier":
            bars.append(i)

    def loosened(i, counter):
        m = list(seg)
        n = int(re.search(counter + r"\((\d+)\)", m[i]).group(1))
        m[i] = m[i].replace("%s(%d)" % (counter, n), "%s(%d)" % (counter, n + 1))
        return m

    # the hand-counted waits of the loop body (the first ones of each list belong to the prologue)
    hits = [bool(run(loosened(i, "vmcnt"))) for i in vm[1:4]]
    assert any(hits), "no loosened vmcnt wait was reported"
    found = [f for i in lg[3:9] for f in run(loosened(i, "lgkmcnt"))]
    assert any("asm ds_read in flight" in f for f in found), "no loosened lgkmcnt wait was reported"
    no_barrier = list(seg)
    no_barrier[bars[1]] = "\ts_nop 0"
    assert run(no_barrier), "a removed tile barrier was not reported"


def test_dataset_accepts_game_records_and_refuses_unknown_entries():
    """DatasetGame.append / += take a slot-free GameRecord (anything answering get_history()) like a
    Game, another dataset's games, and raise on anything else instead of dropping it silently."""
    from chessrl_amd import records
    from chessrl_amd.dataset import DatasetGame
    from chessrl_amd.game import uci_to_move
    rec = records.GameRecord(3, [uci_to_move(u) for u in ["e2e4", "e7e5"]], 1, True, "d")
    ds = DatasetGame()
    ds.append(rec)
    ds += rec
    other = DatasetGame([rec])
    ds += other
    assert len(ds) == 3 and json.loads(str(ds))[0]["moves"] == ["e2e4", "e7e5"]
    with pytest.raises(TypeError):
        ds.append({"moves": ["e2e4"]})
    with pytest.raises(TypeError):
        ds += 5


def test_a_truncated_game_is_not_trained_on_after_a_round_trip_through_the_stored_records():
    """``GameRecord.truncated`` does not survive ``get_history()`` / gameplays.json (the reference's
    record has no such key): a reloaded truncated game is ``result: null`` with moves.  The training jobs pick
    their games by RESULT, so neither form reaches ``DataGameSequence`` (which refuses unfinished games)."""
    from chessrl_amd import records
    from chessrl_amd.game import uci_to_move
    from chessrl_amd.selfplay import trainable_records
    mv = [uci_to_move(u) for u in ["e2e4", "e7e5", "g1f3"]]
    recs = [records.GameRecord(0, mv, 1, True), records.GameRecord(1, mv, None, False, truncated=True),
            records.GameRecord(2, [], 0, True), records.GameRecord(3, mv, 0, False)]
    assert [r.game_id for r in trainable_records(recs)] == [0, 3]
    back = records.loads(records.dumps(recs))                      # what a later run reads from gameplays.json
    assert [b.truncated for b in back] == [False] * 4 and back[1].result is None
    assert [r.game_id for r in trainable_records(back)] == [0, 3]


def test_board_list_header_is_one_number_in_the_header_the_kernels_and_the_binding():
    """The hybrid mode's board list is int32 [CRL_LIST_HEADER + n]: the header's define, the kernels' LIST_HEADER
    (csrc/tower_common.hpp) and the ctypes binding's constant must be the same number (round 5 widened the header from 2 to 4
    words for the 64-bit running total)."""
    from chessrl_amd import _lib
    header = open(os.path.join(ROOT, "include", "chessrl_hip.h")).read()
    common = open(os.path.join(ROOT, "chessrl_amd", "csrc", "tower_common.hpp")).read()
    h = int(re.search(r"#define\s+CRL_LIST_HEADER\s+(\d+)", header).group(1))
    k = int(re.search(r"constexpr int LIST_HEADER = (\d+);", common).group(1))
    assert h == k == _lib.LIST_HEADER == 4
    assert int(re.search(r"#define\s+CRL_ABI_VERSION\s+(\d+)", header).group(1)) == _lib.ABI_VERSION
