"""Lockstep search engine: G independent self-play games advance one MCTS
simulation per step on one MI355X.

This is the host-side driver of the HIP kernels behind the reference's
``SelfPlayTree.search_move`` (/root/reference/src/chessrl/mctree.py:159-214).
One *step* = one ``explore_tree`` for every game:

    crl_sim_select_expand   backprop(previous sim) + select + expand up to S1
    evaluator(planes S1)    tower forward #1  -> policy(S1)      (opponent reply)
    crl_sim_reply           greedy reply, S2, Node(S2), encode S2
    evaluator(planes S2)    tower forward #2  -> policy(S2), value(S2)

All four are enqueued on one HIP stream with no host synchronisation and are
captured once into a hipGraph (torch.cuda.CUDAGraph) that is replayed S times
per move.  Tensors never leave HBM: the encoder kernels write the tower's input
in place -- 128 plane bitboards per position for the fused HIP trunk, which
expands them on chip, or fp16 NHWC planes for any other evaluator -- and the
search kernels read the tower's fp32 outputs in place
(``torch.Tensor.data_ptr()`` across the C-ABI).

Sequential ``threads=1`` semantics of the reference (the only deterministic
mode): per game, simulations are strictly ordered.
"""
import numpy as np
import torch

from . import _lib


def compute_policy(visits, root_visits, nb_moves, noise=True, rng=None):
    """SelfPlayTree.compute_policy (mctree.py:305-322) on host numpy.

    ``visits`` in children order.  ``rng``: a numpy Generator/RandomState or
    None for the global ``np.random`` stream the reference draws from.
    """
    tau = 1
    if nb_moves >= 30:
        tau = nb_moves / (1 + np.power(nb_moves, 1.3))
    policy = np.array([np.power(v, 1 / tau) for v in visits]) / np.power(root_visits, 1 / tau)
    if noise:
        epsilon = 0.25
        policy = (1 - epsilon) * policy + (rng or np.random).dirichlet([0.03] * len(visits))
    return policy


_ALPHA = {}


def dirichlet_row(src, n):
    """``np.random.dirichlet([0.03] * n)`` (mctree.py:319-320) from the stream ``src``."""
    a = _ALPHA.get(n)
    if a is None:
        a = _ALPHA[n] = np.full(n, 0.03)
    return src.dirichlet(a)


def choose_children(visits, nchild, root_visits, plies, noise=True, rngs=None, noise_rows=None):
    """``np.argmax(compute_policy(...))`` for every game at once (-1 where nchild == 0).

    Same arithmetic as ``compute_policy`` element for element (numpy's array ``power``, multiply, add
    and division are the scalar ones applied per element), so the chosen child is identical; only
    the per-game Dirichlet draw stays a loop because every game owns its random stream.
    ``visits`` [G, >=max(nchild)] int, children order; ``rngs`` one generator per game (or None
    for the global ``np.random`` stream, drawn in game order); ``noise_rows``: the draws already
    made for this move (``SelfPlayRunner`` makes them while the GPU searches) -- either a list with
    one row of ``nchild[g]`` entries per game with children, or a pair (matrix float64 [G, >=max
    nchild] holding the rows left-aligned, counts int [G]): the latter is consumed without a Python
    loop (4096 per-game argmax calls were most of a move boundary).
    """
    G = len(nchild)
    nchild = np.asarray(nchild)
    plies_f = np.asarray(plies, dtype=np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        tau = np.where(plies_f >= 30, plies_f / (1 + np.power(plies_f, 1.3)), 1.0)
        inv_tau = 1 / tau
        w = int(nchild.max()) if G else 0
        pol = (np.power(np.asarray(visits)[:, :w].astype(np.float64), inv_tau[:, None]) /
               np.power(np.asarray(root_visits, dtype=np.float64), inv_tau)[:, None])
    chosen = np.full(G, -1, dtype=np.int32)
    cols = np.arange(w)
    live = nchild > 0
    if noise and isinstance(noise_rows, tuple):
        mat, cnt = noise_rows
        bad = np.nonzero(live & (np.asarray(cnt)[:G] != nchild))[0]
        if len(bad):
            g = int(bad[0])
            raise RuntimeError("noise drawn ahead for %d children, game %d has %d" % (int(cnt[g]), g, int(nchild[g])))
        if mat.shape[1] < w:
            raise RuntimeError("noise matrix narrower than the widest root")
        with np.errstate(invalid="ignore"):
            noisy = (1 - 0.25) * pol + mat[:G, :w]
        masked = np.where(cols[None, :] < nchild[:, None], noisy, -np.inf)
        chosen[live] = np.argmax(masked[live], axis=1)
    elif noise:
        for g in range(G):
            n = int(nchild[g])
            if n:
                if noise_rows is not None:
                    row = noise_rows[g]
                    if row is None or len(row) != n:
                        raise RuntimeError("noise drawn ahead for %s children, game %d has %d"
                                           % (None if row is None else len(row), g, n))
                else:
                    row = dirichlet_row(rngs[g] if rngs is not None else np.random, n)
                chosen[g] = int(np.argmax((1 - 0.25) * pol[g, :n] + row))
    else:
        masked = np.where(cols[None, :] < nchild[:, None], pol, -np.inf)
        chosen[live] = np.argmax(masked[live], axis=1)
    return chosen


def resolve_numpy_promotion(mode="auto"):
    """Which arithmetic ``Node.get_value``'s ``C * self.prior`` (mctree.py:79-87: python int x
    ``np.float32`` scalar) has.  Under the reference's pinned numpy==1.17.2 (requirements.txt:6;
    any numpy < 2) value-based scalar promotion makes it a float64 product: ``"legacy"``; under
    numpy >= 2 (NEP 50) it is rounded to float32: ``"nep50"``.  ``"auto"`` (the default
    everywhere) asks the numpy this process runs on -- the same numpy the reference would run on
    if it were started in this interpreter, so a drop-in replacement computes what the code it
    replaces computed there."""
    if mode == "auto":
        return "legacy" if (10 * np.float32(0.1)).dtype == np.float64 else "nep50"
    if mode not in ("nep50", "legacy"):
        raise ValueError("numpy_promotion must be 'auto', 'nep50' or 'legacy'")
    return mode


# stamp ids (StampRing): the engine's phase boundaries and the evaluator's trunk launches
STAMP_STEP, STAMP_SELECTED, STAMP_S1_DONE, STAMP_REPLIED, STAMP_GRAPH_END = 0, 1, 2, 3, 4
STAMP_TRUNK = {"f16": (8, 9), "f16x3": (10, 11), "f16x3 indexed": (12, 13)}      # (begin, end) per trunk arithmetic


class StampRing(object):
    """Device ring of (id, device wall clock) pairs written by ``crl_stamp`` (include/chessrl_hip.h): one-thread
    kernels that capture into a hipGraph like any other, so the time of every phase of a step and of every trunk
    launch comes from the REPLAYED graph -- consecutive stamps telescope to the step -- instead of from eager
    launches timed beside it (C5's eager phases added up to 1.03-1.10 x its graph-replayed step)."""

    def __init__(self, capacity, device):
        import ctypes
        self._ct = ctypes
        self.capacity = int(capacity)
        self.dev = torch.device(device)
        self.ring = torch.zeros(2 + 2 * self.capacity, dtype=torch.int64, device=self.dev)
        khz = _lib.lib().crl_stamp_clock_khz(self.dev.index or 0)
        if khz <= 0:
            raise _lib.HipLibraryError("crl_stamp_clock_khz failed (%d)" % khz)
        self.ticks_per_ms = float(khz)

    def stamp(self, sid):
        vp = self._ct.c_void_p
        rc = _lib.lib().crl_stamp(vp(torch.cuda.current_stream(self.dev).cuda_stream), vp(self.ring.data_ptr()),
                                  self.capacity, int(sid))
        if rc != 0:
            raise _lib.HipLibraryError("crl_stamp failed (%d)" % rc)

    def clear(self):
        self.ring.zero_()

    def read(self):
        """[(id, milliseconds since the first stamp)] in the order written (synchronises)."""
        torch.cuda.synchronize(self.dev)
        host = self.ring.cpu().numpy()
        n = int(host[0])
        if n > self.capacity:
            raise RuntimeError("stamp ring overflowed (%d stamps, capacity %d)" % (n, self.capacity))
        ids, clk = host[2:2 + 2 * n:2], host[3:3 + 2 * n:2]
        return [(int(i), float(c - clk[0]) / self.ticks_per_ms) for i, c in zip(ids, clk)]


def summarise_stamps(stamps):
    """Per-phase means of a stamped run (``StampRing.read()``).  Every interval between two consecutive stamps is
    attributed to what ran in it -- so the parts add up to the wall time between the first and the last stamp BY
    CONSTRUCTION; what the comparison with the un-stamped timed step then shows is the cost of the stamps themselves
    (one launch gap each).  An interval that starts at a trunk-begin stamp is that trunk launch (one kernel for the
    fused trunk, the 42 of a layer-wise forward); the rest of a tower phase (heads, the hybrid mode's margin kernel,
    the stamps' own gaps) is "tower_s1 other" / "tower_s2 other".
    Returns {steps, ms_per_step, stamps_per_step, parts: {name: ms per step}, trunk: {kind: {launch_ms,
    launches_per_step, min_ms, max_ms}}}."""
    phase_of = {STAMP_STEP: "select_expand", STAMP_SELECTED: "tower_s1", STAMP_S1_DONE: "reply",
                STAMP_REPLIED: "tower_s2", STAMP_GRAPH_END: "graph_launch_gap"}
    begin = {b: k for k, (b, e) in STAMP_TRUNK.items()}
    parts, trunk, steps, phase = {}, {}, 0, None
    first = next((k for k, (i, _) in enumerate(stamps) if i == STAMP_STEP), len(stamps))
    stamps = stamps[first:]                  # (evaluations in front of the first step -- warm-up, the root -- are not steps)
    for (i0, t0), (i1, t1) in zip(stamps[:-1], stamps[1:]):
        dt = t1 - t0
        if i0 == STAMP_STEP:
            steps += 1
        phase = phase_of.get(i0, phase)
        if i0 in begin:
            kind = begin[i0]
            if i1 != STAMP_TRUNK[kind][1]:
                raise RuntimeError("stamp %d (trunk begin) followed by %d" % (i0, i1))
            trunk.setdefault(kind, []).append(dt)
            key = "trunk %s (%s)" % (kind, "s1" if phase == "tower_s1" else "s2")
        elif phase in ("tower_s1", "tower_s2"):
            key = phase + " other"
        else:
            key = phase or "before the first step"
        parts[key] = parts.get(key, 0.0) + dt
    if steps == 0:
        raise RuntimeError("no complete step among the stamps")
    total = stamps[-1][1] - stamps[0][1]
    return {"steps": steps, "ms_per_step": total / steps, "stamps_per_step": (len(stamps) - 1) / steps,
            "parts": {k: v / steps for k, v in sorted(parts.items())},
            "trunk": {k: {"launch_ms": sum(v) / len(v), "launches_per_step": len(v) / steps,
                          "min_ms": min(v), "max_ms": max(v)} for k, v in trunk.items()}}


class LockstepEngine(object):
    """G games x one search tree each on one GPU.

    evaluator(planes[G,8,8,128] fp16 cuda) -> (policy[G,1968] f32, value[G] f32) cuda tensors; an
    evaluator with ``accepts_bitplanes`` is given int64 [G,128] plane bitboards instead.
    """

    def __init__(self, evaluator, n_games, max_sims, device=0, max_plies=4096,
                 numpy_promotion="auto", use_graph=True, bitplanes=None, legal_priors=None, raw_priors=None,
                 steps_per_graph=None):
        numpy_promotion = resolve_numpy_promotion(numpy_promotion)
        if not torch.cuda.is_available():
            raise _lib.HipLibraryError("LockstepEngine needs an MI355X: no CPU fallback exists")
        self.numpy_promotion = numpy_promotion
        self.G, self.max_sims = n_games, max_sims
        self.dev = torch.device("cuda", device)
        torch.cuda.set_device(self.dev)
        self.ctx = _lib.Context(n_games, max_sims, max_plies=max_plies, device=device,
                                numpy_legacy=(numpy_promotion == "legacy"))
        self.evaluator = evaluator
        G = n_games
        # An evaluator that runs the fused HIP trunk takes the encoder's compact form -- 128 plane
        # bitboards (1 KiB) per position -- and expands it on chip: the 16-KiB fp16 planes are then
        # never written to HBM.  Any other evaluator gets the fp16 NHWC planes.
        if bitplanes is None:
            bitplanes = bool(getattr(evaluator, "accepts_bitplanes", False))
        self.bitplanes = bitplanes
        if bitplanes:
            self.ctx.set_plane_format(True)
            self.planes_s1 = torch.zeros((G, _lib.PLANES), dtype=torch.int64, device=self.dev)
            self.planes_s2 = torch.zeros((G, _lib.PLANES), dtype=torch.int64, device=self.dev)
        else:
            self.planes_s1 = torch.zeros((G, 8, 8, _lib.PLANES), dtype=torch.float16, device=self.dev)
            self.planes_s2 = torch.zeros((G, 8, 8, _lib.PLANES), dtype=torch.float16, device=self.dev)
        self.pol_s1 = torch.zeros((G, _lib.N_LABELS), dtype=torch.float32, device=self.dev)
        self.pol_s2 = torch.zeros((G, _lib.N_LABELS), dtype=torch.float32, device=self.dev)
        self.val_s2 = torch.zeros((G,), dtype=torch.float32, device=self.dev)
        # An evaluator with ``accepts_legal_labels`` (the HIP heads) is told which labels the search
        # will read -- the legal moves of S1 / S2, listed by the kernel that generated them -- and
        # writes only those probabilities: priors [G,256] instead of policy [G,1968] per call
        # (CRL_POLICY_LEGAL).  Root priors and greedy openings still use the full vectors.
        if legal_priors is None:
            legal_priors = bool(getattr(evaluator, "accepts_legal_labels", False))
        self.legal_priors = legal_priors
        self.pri_s1 = self.pri_s2 = None
        self.raw_priors = False
        self._want_raw = raw_priors              # None: wherever the heads support it; False: never
        if legal_priors:
            self.pri_s1 = torch.zeros((G, _lib.MAX_MOVES), dtype=torch.float32, device=self.dev)
            self.pri_s2 = torch.zeros((G, _lib.MAX_MOVES), dtype=torch.float32, device=self.dev)
            self._lab_s1 = self.ctx.eval_labels(0)
            self._lab_s2 = self.ctx.eval_labels(1)
            # small batches: the heads leave logits + the softmax statistics of their label slices and
            # the search kernels normalise on read (CRL_POLICY_LEGAL_RAW): one launch fewer per tower
            # call.  Each evaluation slot owns its statistics (S2's are read by the NEXT step's select).
            self.stats_s1 = torch.zeros((G, 16), dtype=torch.float32, device=self.dev)
            self.stats_s2 = torch.zeros((G, 16), dtype=torch.float32, device=self.dev)
            self.ctx.set_policy_stats(0, self.stats_s1.data_ptr())
            self.ctx.set_policy_stats(1, self.stats_s2.data_ptr())
            self._pick_policy_format()
        else:
            self.pri_s1, self.pri_s2 = self.pol_s1, self.pol_s2
        self._full = (self.planes_s1, self.planes_s2, self.pol_s1, self.pol_s2, self.val_s2,
                      self.pri_s1, self.pri_s2)
        self.use_graph = use_graph
        if steps_per_graph is not None:
            if int(steps_per_graph) < 1:
                raise ValueError("steps_per_graph must be >= 1")
            self.STEPS_PER_GRAPH = int(steps_per_graph)      # (instance override of the class default)
        self.stamps = None                   # measurement: a StampRing -> every phase of a step is stamped (bench.py)
        self._stamped_ring = None            # ... the ring the cached stamped graphs write into
        self._graphs = {}                    # (steps per graph, stamped?) -> captured hipGraph
        self._graph_epoch = 0
        self._bind_stream()

    # ---- plumbing ---------------------------------------------------------------------
    def _pick_policy_format(self):
        """CRL_POLICY_LEGAL_RAW where the evaluator's heads can skip their normalising pass for this
        batch size, else CRL_POLICY_LEGAL.  Only called while no simulation is pending (construction,
        ``shrink`` at a move boundary): priors written in one format are never read in the other."""
        sup = getattr(self.evaluator, "raw_priors_supported", None)
        self.raw_priors = bool(self._want_raw is not False and sup is not None and sup(self.G))
        self.ctx.set_policy_format(_lib.POLICY_LEGAL_RAW if self.raw_priors else _lib.POLICY_LEGAL)

    def _bind_stream(self):
        self.ctx.set_stream(torch.cuda.current_stream(self.dev).cuda_stream)

    def close(self):
        self._graph = None
        self.ctx.close()

    def shrink(self, n):
        """Keep only slots [0, n) in the lockstep batch (finite runs: the batch thins out as games
        end; the caller compacts the running games into the first slots with ``ctx.copy_game``).
        Every later launch, tower batch and the re-captured hipGraph cover n games."""
        if not 0 < n <= self._full[0].shape[0] or n % 4:
            raise ValueError("shrink: n must be a multiple of 4 within the engine's capacity")
        self.ctx.set_window(0, n)
        self.G = n
        (self.planes_s1, self.planes_s2, self.pol_s1, self.pol_s2, self.val_s2,
         self.pri_s1, self.pri_s2) = (t[:n] for t in self._full)
        if self.legal_priors:
            self._pick_policy_format()
        self._graph = None

    def _eval_into(self, planes, pol_out, val_out):
        fi = getattr(self.evaluator, "forward_into", None)
        if fi is not None:                       # ChessModel: no extra copy of the 32 MB policy
            fi(planes, pol_out, val_out)
            return
        pol, val = self.evaluator(planes)
        pol_out.copy_(pol)
        if val_out is not None:
            val_out.copy_(val)

    # the four phases of one simulation step (bench.py times them one by one)
    def phase_select_expand(self):
        self.ctx.sim_select_expand(self.pri_s2.data_ptr(), self.val_s2.data_ptr(), self.planes_s1.data_ptr())

    def phase_tower_s1(self):
        if self.legal_priors:
            self.evaluator.forward_legal_into(self.planes_s1, self._lab_s1[0], self._lab_s1[1], self.pri_s1, None,
                                              **({"stats_out": self.stats_s1} if self.raw_priors else {}))
        else:
            self._eval_into(self.planes_s1, self.pol_s1, None)

    def phase_reply(self):
        self.ctx.sim_reply(self.pri_s1.data_ptr(), self.planes_s2.data_ptr())

    def phase_tower_s2(self):
        if self.legal_priors:
            self.evaluator.forward_legal_into(self.planes_s2, self._lab_s2[0], self._lab_s2[1], self.pri_s2, self.val_s2,
                                              **({"stats_out": self.stats_s2} if self.raw_priors else {}))
        else:
            self._eval_into(self.planes_s2, self.pol_s2, self.val_s2)

    def _step_body(self):
        st = self.stamps
        if st is None:
            self.phase_select_expand()
            self.phase_tower_s1()
            self.phase_reply()
            self.phase_tower_s2()
            return
        # measurement build of the same step (bench.py): one-thread stamp kernels between the phases, captured into
        # the graph with them; the evaluator stamps its own trunk launches (ChessModel.stamp_fn)
        st.stamp(STAMP_STEP)
        self.phase_select_expand()
        st.stamp(STAMP_SELECTED)
        self.phase_tower_s1()
        st.stamp(STAMP_S1_DONE)
        self.phase_reply()
        st.stamp(STAMP_REPLIED)
        self.phase_tower_s2()

    def set_stamps(self, ring):
        """Attach (or with None detach) a StampRing: ``run_steps`` then replays the STAMPED build of the step (captured
        on first use, cached beside the plain graphs: switching between the two costs nothing, so a stamped leg can
        follow a timed window without a capture in between).  Stamped graphs write into the ring they were captured
        with: another ring drops them."""
        if ring is not None and ring is not self._stamped_ring:
            self._graphs = {k: g for k, g in self._graphs.items() if not k[1]}
            self._stamped_ring = ring
        self.stamps = ring
        if hasattr(self.evaluator, "stamp_fn"):
            self.evaluator.stamp_fn = ring.stamp if ring is not None else None

    def drop_stamped_graphs(self):
        self.set_stamps(None)
        self._graphs = {k: g for k, g in self._graphs.items() if not k[1]}
        self._stamped_ring = None

    # One hipGraph launch costs ~12 us between the last kernel of one graph and the first of the next (measured,
    # tools/graph_unroll_probe.py): nothing at C3 (0.5 %), 9 % of a C2 step.  So the engine also keeps a graph of
    # STEPS_PER_GRAPH consecutive steps and ``run_steps(n)`` replays that one for as long as n allows
    # (C2: 133 -> 119 us per step, 3.84 -> 4.29 M simulations/s).
    STEPS_PER_GRAPH = 8

    def _capture(self, k=1):
        # warm the evaluator (library handles, autotuning, buffers it keeps between calls) outside of capture
        prepare = getattr(self.evaluator, "prepare", None)
        if prepare is not None:
            prepare(self.G)
        if not self._graphs:
            side = torch.cuda.Stream(self.dev)
            side.wait_stream(torch.cuda.current_stream(self.dev))
            with torch.cuda.stream(side):
                for _ in range(3):
                    self.evaluator(self.planes_s1)
            torch.cuda.current_stream(self.dev).wait_stream(side)
        torch.cuda.synchronize(self.dev)
        g = torch.cuda.CUDAGraph()
        # thread_local: another thread of this process (rank 0's background trainer) may allocate and launch
        # on its own stream while this thread captures
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            self._bind_stream()              # kernels must land on the capturing stream
            for _ in range(k):
                self._step_body()
            if self.stamps is not None:
                self.stamps.stamp(STAMP_GRAPH_END)
        self._bind_stream()
        self._graphs[(k, self.stamps is not None)] = g
        self._graph_epoch = getattr(self.evaluator, "graph_epoch", 0)
        return g

    @property
    def _graph(self):
        """the one-step graph (None until captured); assigning None drops every captured graph"""
        return self._graphs.get((1, False))

    @_graph.setter
    def _graph(self, value):
        if value is not None:
            raise ValueError("graphs are captured by the engine")
        self._graphs = {}

    def _graph_of(self, k):
        # an evaluator whose kernel choice changed (ChessModel precision "auto" after new weights)
        # says so through graph_epoch: the captured launches are stale, capture again
        if self._graphs and self._graph_epoch != getattr(self.evaluator, "graph_epoch", 0):
            self._graphs = {}
        g = self._graphs.get((k, self.stamps is not None))
        return g if g is not None else self._capture(k)

    def prepare_graphs(self, n_steps=None):
        """Capture now what ``run_steps(n_steps)`` will replay (bench.py: no capture inside a timed region).
        Capturing launches nothing: the search state is untouched."""
        if self.use_graph:
            if n_steps is None or n_steps >= self.STEPS_PER_GRAPH > 1:
                self._graph_of(self.STEPS_PER_GRAPH)
            self._graph_of(1)

    def run_steps(self, n):
        """``n`` simulations for every game (enqueue only, no host sync)."""
        if not self.use_graph:
            for _ in range(n):
                self._step_body()
            return
        K = self.STEPS_PER_GRAPH
        while K > 1 and n >= K:
            self._graph_of(K).replay()
            n -= K
        while n > 0:
            self._graph_of(1).replay()
            n -= 1

    def step(self):
        """One simulation for every game (enqueue only, no host sync)."""
        self.run_steps(1)

    # ---- SelfPlayTree surface -------------------------------------------------------------
    def search_begin(self):
        """Tree.__init__ for every slot + the root's policy (priors of its children)."""
        self._bind_stream()
        self.ctx.search_begin(self.planes_s2.data_ptr())
        self._eval_into(self.planes_s2, self.pol_s2, self.val_s2)
        self.ctx.search_root_priors(self.pol_s2.data_ptr())

    def search(self, n_sims):
        """search_begin + n_sims lockstep simulations + the last backprop."""
        if n_sims > self.max_sims:
            raise ValueError("n_sims exceeds the max_sims this engine was created with")
        self.search_begin()
        self.run_steps(n_sims)
        self.ctx.sim_backup(self.pri_s2.data_ptr(), self.val_s2.data_ptr())

    def root_children(self):
        return self.ctx.root_children()

    def advance(self, chosen):
        return self.ctx.advance(chosen)

    # ---- Game surface (batched) --------------------------------------------------------------
    def reset(self, mask=None):
        self._bind_stream()
        self.ctx.reset_games(mask)

    def greedy_move(self, mask=None, push=True):
        """agent.best_move(game, real_game=True) for the masked slots (+ gam.move(...))."""
        self._bind_stream()
        self.ctx.encode(self.planes_s1.data_ptr())
        self._eval_into(self.planes_s1, self.pol_s1, None)
        return self.ctx.greedy_moves(self.pol_s1.data_ptr(), mask=mask, push=push)

    def load_games(self, games):
        """Slot i becomes a deep copy of ``games[i]`` (chessrl_amd ``Game`` objects: position,
        move stack and history as they stand, whatever position they started from)."""
        from .game import arena
        self._bind_stream()
        src = arena().ctx
        for i, g in enumerate(games):
            self.ctx.copy_game_from(i, src, g._slot)

    def load_moves(self, move_lists):
        """Replay per-slot move id sequences from the start position (tests / Game copies)."""
        self.reset()
        n = max((len(m) for m in move_lists), default=0)
        if n == 0:
            return
        table = np.full((self.G, n), _lib.NO_MOVE, dtype=np.uint16)
        counts = np.zeros(self.G, dtype=np.int32)
        for g, ml in enumerate(move_lists):
            table[g, :len(ml)] = ml
            counts[g] = len(ml)
        pushed = self.ctx.push_sequences(table, counts)
        for g, ml in enumerate(move_lists):
            if pushed[g] != len(ml):
                raise ValueError("illegal move %d in sequence of slot %d" % (int(pushed[g]), g))
