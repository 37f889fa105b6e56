"""ctypes binding of libchessrl_hip.so (the C-ABI in include/chessrl_hip.h).

There is deliberately NO CPU fallback: if the HIP library is missing or no GPU
is visible, every entry point raises.  ``build()`` compiles the library in-tree
with hipcc for gfx950 (works without a GPU: hipcc cross-compiles).
"""
import ctypes
import hashlib
import os
import subprocess

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_PKG)
_CSRC = os.path.join(_PKG, "csrc")
SO_PATH = os.path.join(_PKG, "libchessrl_hip.so")
HEADER = os.path.join(_ROOT, "include", "chessrl_hip.h")
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared"]


def sources():
    """Every file the library is compiled from (all of csrc/ and the public header)."""
    files = sorted(os.path.join(_CSRC, f) for f in os.listdir(_CSRC) if f.endswith((".hip", ".hpp", ".h")))
    return files + [HEADER]


def source_hash():
    h = hashlib.sha256(" ".join(HIPCC_FLAGS).encode())
    for f in sources():
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()


MAX_MOVES = 256
N_LABELS = 1968
PLANES = 128
NO_MOVE = 0xFFFF
RESULT_NONE = 2
FLAG_NUMPY_LEGACY = 1
POLICY_FULL, POLICY_LEGAL, POLICY_LEGAL_RAW = 0, 1, 2
TRUNK_BITPLANES = 1
TRUNK_SPLIT = 2
LIST_HEADER = 4          # include/chessrl_hip.h: CRL_LIST_HEADER (int32 words in front of a board list)

# every symbol include/chessrl_hip.h declares (tests check the .so exports all of them)
SYMBOLS = [
    "crl_create", "crl_destroy", "crl_set_stream", "crl_sync", "crl_last_error", "crl_max_games",
    "crl_max_sims", "crl_set_window", "crl_set_plane_format", "crl_set_policy_format", "crl_eval_labels",
    "crl_copy_game", "crl_copy_game_from", "crl_uci_label_moves", "crl_reset_games", "crl_set_positions",
    "crl_get_positions", "crl_legal_moves", "crl_push_moves", "crl_push_sequences", "crl_results", "crl_records",
    "crl_encode", "crl_greedy_moves", "crl_search_begin", "crl_search_root_priors",
    "crl_sim_select_expand", "crl_sim_reply", "crl_sim_backup", "crl_root_children",
    "crl_advance", "crl_counters", "crl_trunk_forward",
    "crl_trunk_forward_bitplanes", "crl_trunk_forward_x", "crl_trunk_set_small_batch", "crl_trunk_kernel_name", "crl_heads_forward",
    "crl_heads_forward_legal", "crl_heads_forward_legal_raw", "crl_heads_raw_supported", "crl_heads_set_sliced_max",
    "crl_set_policy_stats", "crl_abi_version", "crl_source_hash", "crl_reply_margin", "crl_trunk_forward_indexed", "crl_trunk_workspace_bytes",
    "crl_end_move_fetch", "crl_advance_fetch",
    "crl_im2col3x3_f32", "crl_col2im3x3_f32", "crl_stamp", "crl_stamp_clock_khz",
]


class HipLibraryError(RuntimeError):
    pass


ABI_VERSION = 8          # include/chessrl_hip.h: CRL_ABI_VERSION (checked against the loaded library)
_HASH_MARK = b"CRL_SRC_HASH="


def embedded_hash(so=SO_PATH):
    """The source hash compiled INTO the library (csrc/api.hip: g_source_hash), read from the file
    without loading it; None when the file is missing or carries no marker."""
    try:
        data = open(so, "rb").read()
    except OSError:
        return None
    i = data.find(_HASH_MARK)
    if i < 0:
        return None
    tail = data[i + len(_HASH_MARK):i + len(_HASH_MARK) + 64]
    return tail.split(b"\0")[0].decode("ascii", "replace")


def have_sources():
    return os.path.isdir(_CSRC) and os.path.exists(HEADER)


def is_stale(so=SO_PATH):
    """True when `so` is missing or was compiled from other sources than the ones present: the
    library carries the sha256 of its sources inside (no side file that a deploy can lose or an
    interrupted build can leave behind).  A deploy of the library WITHOUT csrc/ has nothing to be
    compared with and is taken as shipped."""
    if not os.path.exists(so):
        return True
    if not have_sources():
        return False
    return embedded_hash(so) != source_hash()


def build(force=False, verbose=False):
    """hipcc --offload-arch=gfx950 -> chessrl_amd/libchessrl_hip.so (in-tree; works without a GPU).
    Recompiles whenever any file of csrc/ or the header changed since the library was built
    (``force`` recompiles regardless).  The sources' hash is compiled in (``crl_source_hash``)."""
    so = SO_PATH
    if not force and not is_stale(so):
        return so
    # several ranks of one node may arrive here together (torchrun starts one process per GPU):
    # one compiles, the others wait on the lock and find the library fresh; the library appears
    # atomically (rename), so nobody ever loads a half-written file
    import fcntl
    with open(so + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if not force and not is_stale(so):
            return so
        tmp = "%s.tmp.%d" % (so, os.getpid())
        cmd = (["hipcc"] + HIPCC_FLAGS + ['-DCRL_SOURCE_HASH="%s"' % source_hash()] +
               ["-o", tmp, os.path.join(_CSRC, "api.hip")])
        if verbose:
            print(" ".join(cmd))
        try:
            subprocess.check_call(cmd)
            os.replace(tmp, so)
        finally:
            if os.path.exists(tmp):
                os.remove(tmp)
    return so


_lib = None


def lib():
    """Load the shared library.  It is compiled first when it is missing or was built from other
    sources than the csrc/ present; a stale library that cannot be rebuilt is REFUSED (its entry points
    may have other signatures than this binding's), and so is a library of another ABI version.  No
    CPU fallback: without a loadable library this raises."""
    global _lib
    if _lib is not None:
        return _lib
    # torch bundles its own libamdhip64; it must be the copy this process binds, or the tower and
    # the search kernels would sit on two HIP runtimes (and the second one sees no device)
    import torch  # noqa: F401
    so = SO_PATH
    if is_stale(so):
        try:
            build()
        except Exception as e:  # pragma: no cover
            raise HipLibraryError(
                "%s is %s and could not be built with hipcc (%s); the HIP path has no CPU fallback and a "
                "library built from other sources is not loaded"
                % (os.path.basename(so), "missing" if not os.path.exists(so) else "older than csrc/", e))
    try:
        L = ctypes.CDLL(so)
    except OSError as e:
        raise HipLibraryError("cannot load %s: %s" % (so, e))
    try:
        L.crl_abi_version.restype = ctypes.c_int
        L.crl_source_hash.restype = ctypes.c_char_p
        abi, built_from = L.crl_abi_version(), (L.crl_source_hash() or b"").decode()
    except AttributeError:
        raise HipLibraryError("%s exports no crl_abi_version: it predates this binding" % so)
    if abi != ABI_VERSION:
        raise HipLibraryError("%s has ABI version %d, this binding needs %d" % (so, abi, ABI_VERSION))
    if have_sources() and built_from != source_hash():
        raise HipLibraryError("%s was built from other sources (%s...) than csrc/ (%s...)"
                              % (so, built_from[:12], source_hash()[:12]))
    vp, i32, u32 = ctypes.c_void_p, ctypes.c_int, ctypes.c_uint32
    L.crl_create.argtypes = [ctypes.POINTER(vp), i32, i32, i32, i32, u32]
    L.crl_destroy.argtypes = [vp]
    L.crl_destroy.restype = None
    L.crl_set_stream.argtypes = [vp, vp]
    L.crl_sync.argtypes = [vp]
    L.crl_last_error.argtypes = [vp]
    L.crl_last_error.restype = ctypes.c_char_p
    L.crl_max_games.argtypes = [vp]
    L.crl_max_sims.argtypes = [vp]
    L.crl_set_window.argtypes = [vp, i32, i32]
    L.crl_copy_game.argtypes = [vp, i32, i32]
    L.crl_copy_game_from.argtypes = [vp, i32, vp, i32]
    L.crl_uci_label_moves.argtypes = [vp]
    L.crl_reset_games.argtypes = [vp, vp]
    L.crl_set_positions.argtypes = [vp, vp, i32]
    L.crl_get_positions.argtypes = [vp, vp, i32]
    L.crl_legal_moves.argtypes = [vp, vp, vp]
    L.crl_push_moves.argtypes = [vp, vp, vp]
    L.crl_push_sequences.argtypes = [vp, vp, vp, i32, vp]
    L.crl_results.argtypes = [vp, vp]
    L.crl_records.argtypes = [vp, vp, vp, vp]
    L.crl_encode.argtypes = [vp, vp]
    L.crl_greedy_moves.argtypes = [vp, vp, vp, i32, vp]
    L.crl_search_begin.argtypes = [vp, vp]
    L.crl_search_root_priors.argtypes = [vp, vp]
    L.crl_sim_select_expand.argtypes = [vp, vp, vp, vp]
    L.crl_sim_reply.argtypes = [vp, vp, vp]
    L.crl_sim_backup.argtypes = [vp, vp, vp]
    L.crl_root_children.argtypes = [vp] * 8
    L.crl_advance.argtypes = [vp, vp, vp, vp]
    L.crl_counters.argtypes = [vp, vp]
    L.crl_end_move_fetch.argtypes = [vp, vp, vp, vp, vp, vp, vp]
    L.crl_advance_fetch.argtypes = [vp, vp, vp, vp, vp, vp]
    L.crl_trunk_forward.argtypes = [vp, i32, vp, vp, vp, vp, i32, i32, vp, vp, vp]
    L.crl_trunk_forward_bitplanes.argtypes = [vp, i32, vp, vp, vp, vp, i32, i32, vp, vp, vp]
    L.crl_trunk_forward_x.argtypes = [vp, i32, i32, vp, vp, vp, vp, i32, i32, vp, vp, vp, vp, ctypes.c_size_t]
    L.crl_trunk_workspace_bytes.argtypes = [i32, i32, i32]
    L.crl_trunk_workspace_bytes.restype = ctypes.c_size_t
    L.crl_set_plane_format.argtypes = [vp, i32]
    L.crl_reply_margin.argtypes = [vp, vp, vp, i32, vp, i32, vp]
    L.crl_trunk_forward_indexed.argtypes = [vp, i32, vp, vp, vp, i32, i32, vp, vp, vp, vp, vp, ctypes.c_size_t]
    L.crl_trunk_set_small_batch.argtypes = [i32]
    L.crl_trunk_kernel_name.argtypes = [i32, i32, i32, ctypes.c_char_p, i32]
    L.crl_heads_forward.argtypes = [vp, vp, i32, vp, vp, vp, vp, vp, vp, vp, vp]
    L.crl_heads_forward_legal.argtypes = [vp, vp, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    L.crl_heads_forward_legal_raw.argtypes = [vp, vp, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    L.crl_heads_raw_supported.argtypes = [i32]
    L.crl_heads_set_sliced_max.argtypes = [i32]
    L.crl_set_policy_format.argtypes = [vp, i32]
    L.crl_set_policy_stats.argtypes = [vp, i32, vp]
    L.crl_eval_labels.argtypes = [vp, i32, ctypes.POINTER(vp), ctypes.POINTER(vp)]
    L.crl_im2col3x3_f32.argtypes = [vp, vp, vp, i32, i32]
    L.crl_col2im3x3_f32.argtypes = [vp, vp, vp, i32, i32]
    L.crl_stamp.argtypes = [vp, vp, u32, u32]
    L.crl_stamp_clock_khz.argtypes = [i32]
    for name in SYMBOLS:
        if name not in ("crl_destroy", "crl_last_error", "crl_source_hash"):
            getattr(L, name).restype = ctypes.c_int
    _lib = L
    return L


def _ptr(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def uci_label_moves():
    out = np.zeros(N_LABELS, dtype=np.uint16)
    rc = lib().crl_uci_label_moves(_ptr(out))
    if rc != 0:
        raise HipLibraryError("crl_uci_label_moves failed (%d)" % rc)
    return out


class Context(object):
    """One crl_ctx (one GPU).  Thin, numpy-in/numpy-out; device pointers are ints."""

    def __init__(self, max_games, max_sims, max_plies=4096, device=0, numpy_legacy=False):
        self._h = ctypes.c_void_p()
        self._L = lib()
        flags = FLAG_NUMPY_LEGACY if numpy_legacy else 0
        rc = self._L.crl_create(ctypes.byref(self._h), device, max_games, max_sims, max_plies, flags)
        if rc != 0:
            msg = self._L.crl_last_error(None)
            self._h = None
            raise HipLibraryError("crl_create failed (%d): %s" % (rc, (msg or b"").decode()))
        self.G, self.max_sims, self.max_plies, self.device = max_games, max_sims, max_plies, device
        self.n_slots = max_games

    def close(self):
        if getattr(self, "_h", None):
            self._L.crl_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc, what):
        if rc != 0:
            msg = self._L.crl_last_error(self._h)
            raise HipLibraryError("%s failed (%d): %s" % (what, rc, (msg or b"").decode()))

    def set_stream(self, stream_ptr):
        self._ck(self._L.crl_set_stream(self._h, ctypes.c_void_p(stream_ptr)), "crl_set_stream")

    def set_window(self, first, count):
        """Later calls act on slots [first, first+count); arrays become `count` rows."""
        self._ck(self._L.crl_set_window(self._h, first, count), "crl_set_window")
        self.G = count

    def set_policy_format(self, legal):
        """What the simulation entry points take as "policy": False / 0 = full policy[row][1968] rows,
        True / 1 = priors[row][256] of the legal moves (CRL_POLICY_LEGAL), 2 = the same rows holding
        logits that the kernels normalise on read (CRL_POLICY_LEGAL_RAW; ``set_policy_stats`` first)."""
        self._ck(self._L.crl_set_policy_format(self._h, int(legal)), "crl_set_policy_format")

    def set_policy_stats(self, which, dev_stats):
        """Device address of the slice statistics (float [rows][16]) of tower call ``which`` (0: S1, 1: S2)."""
        self._ck(self._L.crl_set_policy_stats(self._h, which, ctypes.c_void_p(dev_stats)), "crl_set_policy_stats")

    def eval_labels(self, which):
        """(labels, counts) device addresses for the position tower call `which` evaluates
        (0: S1 after sim_select_expand, 1: S2 after sim_reply); rows are window-relative."""
        lab, cnt = ctypes.c_void_p(), ctypes.c_void_p()
        self._ck(self._L.crl_eval_labels(self._h, which, ctypes.byref(lab), ctypes.byref(cnt)), "crl_eval_labels")
        return lab.value, cnt.value

    def set_plane_format(self, bits):
        """Encoders write fp16 NHWC planes (False, default) or 128 plane bitboards per position (True)."""
        self._ck(self._L.crl_set_plane_format(self._h, 1 if bits else 0), "crl_set_plane_format")

    def copy_game(self, dst, src):
        self._ck(self._L.crl_copy_game(self._h, dst, src), "crl_copy_game")

    def copy_game_from(self, dst, src_ctx, src):
        """Slot ``dst`` becomes a deep copy of slot ``src`` of another context on the same GPU."""
        self._ck(self._L.crl_copy_game_from(self._h, dst, src_ctx._h, src), "crl_copy_game_from")

    def sync(self):
        self._ck(self._L.crl_sync(self._h), "crl_sync")

    # ---- Game seam -----------------------------------------------------------------
    def reset_games(self, mask=None):
        m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
        self._ck(self._L.crl_reset_games(self._h, _ptr(m)), "crl_reset_games")

    def set_positions(self, boards):
        """boards: np.uint64 [n][8] rows (bb0..5, white, state)."""
        b = np.ascontiguousarray(boards, dtype=np.uint64).reshape(-1, 8)
        self._ck(self._L.crl_set_positions(self._h, _ptr(b), b.shape[0]), "crl_set_positions")

    def get_positions(self, n=None):
        n = self.G if n is None else n
        b = np.zeros((n, 8), dtype=np.uint64)
        self._ck(self._L.crl_get_positions(self._h, _ptr(b), n), "crl_get_positions")
        return b

    def legal_moves(self):
        moves = np.zeros((self.G, MAX_MOVES), dtype=np.uint16)
        counts = np.zeros(self.G, dtype=np.int32)
        self._ck(self._L.crl_legal_moves(self._h, _ptr(moves), _ptr(counts)), "crl_legal_moves")
        return moves, counts

    def legal_counts(self):
        """len(get_legal_moves()) per game without copying the move lists back."""
        counts = np.zeros(self.G, dtype=np.int32)
        self._ck(self._L.crl_legal_moves(self._h, None, _ptr(counts)), "crl_legal_moves")
        return counts

    def push_moves(self, moves):
        m = np.ascontiguousarray(moves, dtype=np.uint16)
        assert m.shape == (self.G,)
        ok = np.zeros(self.G, dtype=np.uint8)
        self._ck(self._L.crl_push_moves(self._h, _ptr(m), _ptr(ok)), "crl_push_moves")
        return ok

    def push_sequences(self, moves, counts):
        """Every game replays its own move list (one launch); returns the number applied per game."""
        m = np.ascontiguousarray(moves, dtype=np.uint16)
        c = np.ascontiguousarray(counts, dtype=np.int32)
        assert m.ndim == 2 and m.shape[0] == self.G and c.shape == (self.G,)
        pushed = np.zeros(self.G, dtype=np.int32)
        self._ck(self._L.crl_push_sequences(self._h, _ptr(m), _ptr(c), m.shape[1], _ptr(pushed)),
                 "crl_push_sequences")
        return pushed

    def results(self):
        r = np.zeros(self.G, dtype=np.int8)
        self._ck(self._L.crl_results(self._h, _ptr(r)), "crl_results")
        return r

    def records(self, with_moves=True):
        moves = np.zeros((self.G, self.max_plies), dtype=np.uint16) if with_moves else None
        plies = np.zeros(self.G, dtype=np.int32)
        res = np.zeros(self.G, dtype=np.int8)
        self._ck(self._L.crl_records(self._h, _ptr(moves), _ptr(plies), _ptr(res)), "crl_records")
        return moves, plies, res

    # ---- encoder / agent seam (device pointers) ----------------------------------------
    def encode(self, dev_planes):
        self._ck(self._L.crl_encode(self._h, ctypes.c_void_p(dev_planes)), "crl_encode")

    def greedy_moves(self, dev_policy, mask=None, push=False, want_moves=True):
        m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
        out = np.zeros(self.G, dtype=np.uint16) if want_moves else None
        self._ck(self._L.crl_greedy_moves(self._h, ctypes.c_void_p(dev_policy), _ptr(m),
                                          1 if push else 0, _ptr(out)), "crl_greedy_moves")
        return out

    # ---- SelfPlayTree seam -------------------------------------------------------------
    def search_begin(self, dev_planes):
        self._ck(self._L.crl_search_begin(self._h, ctypes.c_void_p(dev_planes)), "crl_search_begin")

    def search_root_priors(self, dev_policy):
        self._ck(self._L.crl_search_root_priors(self._h, ctypes.c_void_p(dev_policy)),
                 "crl_search_root_priors")

    def sim_select_expand(self, dev_policy_s2, dev_value_s2, dev_planes_s1):
        self._ck(self._L.crl_sim_select_expand(
            self._h, ctypes.c_void_p(dev_policy_s2), ctypes.c_void_p(dev_value_s2),
            ctypes.c_void_p(dev_planes_s1)), "crl_sim_select_expand")

    def sim_reply(self, dev_policy_s1, dev_planes_s2):
        self._ck(self._L.crl_sim_reply(self._h, ctypes.c_void_p(dev_policy_s1),
                                       ctypes.c_void_p(dev_planes_s2)), "crl_sim_reply")

    def sim_backup(self, dev_policy_s2, dev_value_s2):
        self._ck(self._L.crl_sim_backup(self._h, ctypes.c_void_p(dev_policy_s2),
                                        ctypes.c_void_p(dev_value_s2)), "crl_sim_backup")

    def root_children(self, fields=None):
        """Root children statistics in CHILDREN order; ``fields`` limits what is copied back."""
        G = self.G
        spec = {
            "nchild": ((G,), np.int32), "visits": ((G, MAX_MOVES), np.int32),
            "values": ((G, MAX_MOVES), np.float64), "priors": ((G, MAX_MOVES), np.float32),
            "moves": ((G, MAX_MOVES), np.uint16), "replies": ((G, MAX_MOVES), np.uint16),
            "root_visits": ((G,), np.int32),
        }
        out = {k: (np.zeros(sh, dt) if (fields is None or k in fields) else None)
               for k, (sh, dt) in spec.items()}
        self._ck(self._L.crl_root_children(
            self._h, _ptr(out["nchild"]), _ptr(out["visits"]), _ptr(out["values"]), _ptr(out["priors"]),
            _ptr(out["moves"]), _ptr(out["replies"]), _ptr(out["root_visits"])), "crl_root_children")
        return out

    def advance(self, chosen):
        c = np.ascontiguousarray(chosen, dtype=np.int32)
        assert c.shape == (self.G,)
        bm = np.zeros(self.G, np.uint16)
        am = np.zeros(self.G, np.uint16)
        self._ck(self._L.crl_advance(self._h, _ptr(c), _ptr(bm), _ptr(am)), "crl_advance")
        return bm, am

    def end_move_fetch(self, dev_policy_s2, dev_value_s2):
        """sim_backup + root children (nchild, visits, root_visits) + plies in one synchronising call."""
        G = self.G
        nchild, root_visits, plies = (np.zeros(G, np.int32) for _ in range(3))
        visits = np.zeros((G, MAX_MOVES), np.int32)
        self._ck(self._L.crl_end_move_fetch(self._h, ctypes.c_void_p(dev_policy_s2), ctypes.c_void_p(dev_value_s2),
                                            _ptr(nchild), _ptr(visits), _ptr(root_visits), _ptr(plies)),
                 "crl_end_move_fetch")
        return nchild, visits, root_visits, plies

    def advance_fetch(self, chosen):
        """advance + results + legal-move counts of the next roots in one synchronising call."""
        c = np.ascontiguousarray(chosen, dtype=np.int32)
        assert c.shape == (self.G,)
        res = np.zeros(self.G, np.int8)
        counts = np.zeros(self.G, np.int32)
        self._ck(self._L.crl_advance_fetch(self._h, _ptr(c), None, None, _ptr(res), _ptr(counts)), "crl_advance_fetch")
        return res, counts

    def counters(self):
        c = np.zeros(6, dtype=np.uint64)
        self._ck(self._L.crl_counters(self._h, _ptr(c)), "crl_counters")
        return dict(zip(("sims", "nodes", "depth_sum", "branch_sum", "evals", "terminal_hits"),
                        (int(x) for x in c)))
