"""Self-play driver -- host mirror of the reference's ``selfplay.py``.

``play_game(agent)`` is the reference's per-game loop
(/root/reference/src/chessrl/selfplay.py:59-84) on the drop-in ``Game`` / ``Agent``
objects.  ``SelfPlayRunner`` is the MI355X-first form of the same loop: thousands of
independent games in lockstep on one GPU (one rank), finished slots refilled at move
boundaries so the batch stays full, records gathered across ranks at the end
(``records.gather_records``).  Games shard by ``game_id % world``; every per-game random
stream (colour, Dirichlet noise) is keyed by the GLOBAL game id, so what a game plays
does not depend on the number of GPUs.

``train_model_job`` is the training half of the reference's ``main`` (selfplay.py:98-108,
157-163): one epoch over the games just played, one batch per game.
The CLI keeps ``modeldir --games --threads --debug`` and adds the knobs the reference hard-codes
(``--sims`` 900 in selfplay.py:76, net size, parallel games, ``--rounds`` of play+train).
"""
import argparse
import logging
import os
import random
import time

import numpy as np

from . import _lib
from .engine import LockstepEngine, choose_children, dirichlet_row
from .records import GameRecord

log = logging.getLogger("chessrl_amd.selfplay")


def get_model_path(directory):
    """Weights file a run continues from (selfplay.py:33-56): among the ``model-<v>.npz`` /
    ``model-<v>.h5`` files of ``directory`` the one whose ``<v>.<ext>`` part compares greatest AS A
    STRING (the reference's rule: "9" beats "10"); ``model-0.npz`` when there is none yet."""
    found = [name for name in os.listdir(directory) if name.endswith((".npz", ".h5"))]
    newest = max(found, key=lambda name: name.split("-")[1]) if found else "model-0.npz"
    return os.path.join(directory, newest)


def play_game(agent, max_iters=900):
    """One game, reference-shaped (selfplay.py:59-84): returns the finished ``Game``."""
    from .game import Game
    player_color = True if random.random() >= 0.5 else False
    gam = Game(player_color=player_color)
    agent.color = player_color
    if player_color is False:
        gam.move(agent.best_move(gam, real_game=True))
    while gam.get_result() is None:
        start = time.perf_counter()
        bm, am = agent.best_move(gam, real_game=False, ai_move=True, max_iters=max_iters)
        gam.move(bm)
        gam.move(am)
        log.debug("\tMade move: %s, took: %.2f secs", bm, time.perf_counter() - start)
    log.debug(gam.get_history())
    return gam


def game_color(seed, game_id):
    """Per-game colour stream: ``random.random() >= .5`` (selfplay.py:62) keyed by game id."""
    return random.Random((seed << 20) ^ game_id).random() >= 0.5


_numpy_warm = False


def _warm_numpy():
    """The first large numpy multiply / add of a process that has imported torch costs ~0.2 s (measured: 186 ms
    for a 4096 x 20 float64 ``0.75 * x + y`` against 0.2 ms afterwards); it used to land in the first full move
    boundary (``choose_children``), where the GPU waits.  Paid here, once, while the runner is being built."""
    global _numpy_warm
    if not _numpy_warm:
        x = np.ones((4096, 32))
        (1 - 0.25) * x + x
        np.argmax(np.where(x > 0, x, -np.inf), axis=1)
        _numpy_warm = True


class SelfPlayRunner(object):
    """Lockstep self-play of ``n_parallel`` games on one GPU (one rank of ``world``)."""

    def __init__(self, evaluator, n_parallel, sims, seed=0, noise=True, rank=0, world=1, device=0,
                 max_plies=4096, numpy_promotion="auto", use_graph=True, total_games=None,
                 compact=True, round_size=None, steps_per_graph=None):
        _warm_numpy()
        self.engine = LockstepEngine(evaluator, n_parallel, sims, device=device, max_plies=max_plies,
                                     numpy_promotion=numpy_promotion, use_graph=use_graph,
                                     steps_per_graph=steps_per_graph)
        self.G, self.sims, self.seed, self.noise = n_parallel, sims, seed, noise
        self.rank, self.world = rank, world
        self.total_games = total_games           # global cap on started games (None = endless)
        self.compact = compact                   # finite runs: shrink the batch as games end
        # rolling rounds: ids [r * round_size, (r+1) * round_size) form round r; the batch keeps
        # refilling from the next rounds' ids while the long games of round r finish
        self.round_size = round_size
        self._round_done = {}                    # round -> finished games of this rank
        self.max_plies = max_plies
        self.next_local = 0                      # k-th game of this rank has id rank + world*k
        self.game_id = np.full(n_parallel, -1, dtype=np.int64)
        self.color = np.zeros(n_parallel, dtype=bool)
        self.rngs = [None] * n_parallel
        self.finished = []
        self.moves_played = 0
        self.sims_run = 0
        self.truncated_games = 0                 # records handed over at max_plies (result None)
        self.boundaries = 0                      # move boundaries crossed
        self._sims_in_move = None
        self._noise_rows = None                  # this move's Dirichlet draws, made while the GPU searches
        self._noise_states = {}                  # ... and every drawn stream's state before its draw
        self._root_legal = None                  # len(get_legal_moves()) of every root, read at the boundary
        self._start(np.ones(n_parallel, dtype=bool))

    # ---- slot management ------------------------------------------------------------------
    def _start(self, mask):
        """(Re)start the masked slots with fresh games; black-player games get the opponent's
        greedy opening move first (selfplay.py:68-70)."""
        opening = np.zeros(self.G, dtype=np.uint8)
        reset = np.zeros(self.G, dtype=np.uint8)
        for g in np.nonzero(mask)[0]:
            gid = self.rank + self.world * self.next_local
            if self.total_games is not None and gid >= self.total_games:
                self.game_id[g] = -1
                continue
            self.next_local += 1
            self.game_id[g] = gid
            self.color[g] = game_color(self.seed, gid)
            self.rngs[g] = np.random.default_rng([self.seed, gid])
            reset[g] = 1
            opening[g] = 0 if self.color[g] else 1
        if reset.any():
            self.engine.reset(reset)
        if opening.any():
            self.engine.greedy_move(mask=opening, push=True)

    def active(self):
        return self.game_id >= 0

    # ---- one move for every game --------------------------------------------------------------
    def begin_move(self):
        """Fresh tree per slot (agentdistributed.py:61-63) + the root's priors."""
        if self.noise and self._root_legal is None:
            self._root_legal = self.engine.ctx.legal_counts()
        self.engine.search_begin()
        self._sims_in_move = 0
        self._noise_rows = None

    def step(self):
        """One lockstep simulation for every game; runs the move boundary when the budget of
        ``sims`` per move is reached.  Returns True when a move boundary was crossed."""
        return self.steps(1) > 0

    def steps(self, n):
        """``n`` lockstep simulations for every game, move boundaries included wherever a move's budget of
        ``sims`` completes; returns the number of boundaries crossed.  The same games as ``n`` calls of
        ``step()``: the steps in between are handed to the engine in one piece (``run_steps``: several steps per
        hipGraph launch), cut where the runner has something to do -- the noise draw half-way through a move,
        the boundary."""
        crossed = 0
        half = max(1, self.sims // 2)
        while n > 0:
            if self._sims_in_move is None:
                self.begin_move()
            stop = half if self._sims_in_move < half else self.sims
            k = min(n, stop - self._sims_in_move)
            self.engine.run_steps(k)
            self._sims_in_move += k
            n -= k
            if self._sims_in_move == half:
                self._draw_noise_ahead()
            if self._sims_in_move >= self.sims:
                self.end_move()
                crossed += 1
        return crossed

    def _draw_noise_ahead(self):
        """The Dirichlet draws of this move's ``compute_policy`` (mctree.py:317-320), made on the
        host WHILE the GPU works through the steps already enqueued (launches are asynchronous; the
        host is hundreds of steps ahead) instead of at the move boundary, where the GPU would wait
        for them: 4096 per-game ``dirichlet`` calls are ~25 ms.  A draw needs the number of root
        children the search will end with: every simulation expands one more child of the root until
        all of its legal moves are expanded (mctree.py:216-231), so that is min(sims, legal moves of
        the root), known since the last boundary.  Each game draws from its own stream, in the same
        order and with the same arguments as at the boundary: the values are identical."""
        if not self.noise or self._noise_rows is not None or self._root_legal is None:
            return
        n_final = np.where(self.game_id >= 0, np.minimum(self._root_legal[:self.G], self.sims), 0)
        mat = np.zeros((self.G, max(1, int(n_final.max()))), dtype=np.float64)
        # a move that is cut short after this point (end_move before `sims` simulations) ends with fewer
        # root children and must draw again FROM THE SAME STREAM POSITION: the states are kept until then
        self._noise_states = {}
        for g in np.nonzero(n_final > 0)[0]:
            self._noise_states[g] = self.rngs[g].bit_generator.state
            mat[g, :n_final[g]] = dirichlet_row(self.rngs[g], int(n_final[g]))
        self._noise_rows = (mat, n_final)

    def end_move(self):
        """Last backprop, compute_policy + argmax on the host, the two pushes, harvest of
        finished games and refill of their slots."""
        eng = self.engine
        # (backup + root statistics + plies: one synchronising call; advance + results + the next roots' legal
        # counts: another -- five calls of ~70 us each before)
        nchild_all, visits, root_visits, plies = eng.ctx.end_move_fetch(eng.pri_s2.data_ptr(), eng.val_s2.data_ptr())
        nchild = np.where(self.game_id >= 0, nchild_all, 0)
        rows = self._noise_rows if self._sims_in_move == self.sims else None    # (a shortened move draws here)
        if rows is None and self._noise_rows is not None:
            # shortened AFTER the draw ahead: rewind every stream to where it stood, so that the draw below
            # is the one the reference would make (one dirichlet per move and game, in stream order)
            for g, st in self._noise_states.items():
                self.rngs[g].bit_generator.state = st
        chosen = choose_children(visits, nchild, root_visits, plies, noise=self.noise,
                                 rngs=self.rngs, noise_rows=rows)
        self._noise_rows = None
        # A record that cannot take another full move (our move + the reply) has reached the engine's
        # max_plies: the game is ended HERE and handed over as it stands -- result None, as
        # Game.get_result() (game.py:92-109) says of a game the rules have not ended, flagged
        # ``truncated`` -- instead of overflowing the device's record array (a sticky device error that
        # would end a multi-hour run for one endless game).  Its slot is refilled like any finished one.
        full = self.active() & (chosen >= 0) & (np.asarray(plies) + 2 > self.max_plies)
        chosen[full] = -1
        live = int((chosen >= 0).sum())
        res, next_legal = eng.ctx.advance_fetch(chosen)
        self.moves_played += live
        self.sims_run += live * self._sims_in_move
        self._sims_in_move = None
        done = ((res != _lib.RESULT_NONE) | full) & self.active()
        if done.any():
            moves, plies, res = eng.ctx.records()
            for g in np.nonzero(done)[0]:
                ended = res[g] != _lib.RESULT_NONE
                self.finished.append(GameRecord(self.game_id[g], moves[g, :plies[g]], int(res[g]) if ended else None,
                                                bool(self.color[g]), truncated=not ended))
                if not ended:
                    self.truncated_games += 1
                    log.warning("game %d reached max_plies=%d after %d plies: handed over unfinished (result None)",
                                self.game_id[g], self.max_plies, plies[g])
                if self.round_size:
                    r = int(self.game_id[g]) // self.round_size
                    self._round_done[r] = self._round_done.get(r, 0) + 1
            self._start(done)
            self._maybe_compact()
            next_legal = None                               # slots were reset / moved: count again
        if self.noise:
            self._root_legal = next_legal if next_legal is not None else eng.ctx.legal_counts()
        self.boundaries += 1
        if self.GUARD_EVERY and self.boundaries % self.GUARD_EVERY == 0:
            # an evaluator that chose its arithmetic on a probe (ChessModel precision="auto" -> "f16") is shown the
            # tower inputs this search has just evaluated and may leave that mode (model.py: guard_check)
            check = getattr(eng.evaluator, "guard_check", None)
            if check is not None and eng.bitplanes:
                before = getattr(eng.evaluator, "precision", None)
                margin = getattr(eng.evaluator, "reply_margin", None)
                d = check(eng.planes_s2)
                rule = getattr(eng.evaluator, "reply_rule_check", None)
                if before == "hybrid" and rule is not None and eng.legal_priors:
                    # ... and the reply rule itself on the same positions: a board that chooses another reply in f16 than in
                    # f16x3 must be one the margin lists (model.py: reply_rule_check; widens the margin at once if not)
                    rule(eng.planes_s2, eng._lab_s2[0], eng._lab_s2[1])
                if before == "hybrid" and getattr(eng.evaluator, "reply_margin", None) != margin:
                    log.info("hybrid reply margin widened from %.3e to %.3e (log-policy distance of f16 on this run's own "
                             "tree leaves)", margin, eng.evaluator.reply_margin)
                if d is not None and getattr(eng.evaluator, "precision", None) != before:
                    log.warning("tower precision guard: |f16 - f16x3| = %.2e on this run's own tree leaves (tolerance "
                                "%.1e): %s -> %s from the next move on", d, eng.evaluator.GUARD_TOL, before,
                                eng.evaluator.precision)
        return live

    GUARD_EVERY = 8          # move boundaries between two guard checks (one f16 + one f16x3 evaluation of the batch:
                             # ~5 ms per 8 moves of ~1.9 s at C3)

    COMPACT_MIN = 64

    def _maybe_compact(self):
        """A finite run (``total_games``) stops refilling at some point and the batch thins out;
        whenever the running games fit a cheaper batch size, they are moved into the first
        slots (``crl_copy_game``: board, move stack and history) and the lockstep batch -- search
        kernels, tower batch, hipGraph -- shrinks to it.  What a game plays does not depend on its
        slot (random streams are keyed by the game id), so the records are unchanged."""
        if not self.compact or self.total_games is None or self.G <= self.COMPACT_MIN:
            return
        if self.rank + self.world * self.next_local < self.total_games:
            return                                               # still refilling
        act = np.nonzero(self.active())[0]
        # batch sizes worth switching to: the trunk kernel runs 4 boards per workgroup on 256 CUs, so
        # its cost steps at multiples of 1024 boards; below that the batch is halved
        need = max(len(act), self.COMPACT_MIN)
        if need > 1024:
            n_new = (need + 1023) // 1024 * 1024
        else:
            n_new = self.COMPACT_MIN
            while n_new < need:
                n_new *= 2
        if n_new >= self.G:
            return
        free = [s for s in range(n_new) if self.game_id[s] < 0]
        for src in act[act >= n_new]:
            dst = free.pop()
            self.engine.ctx.copy_game(int(dst), int(src))
            self.game_id[dst], self.color[dst], self.rngs[dst] = self.game_id[src], self.color[src], self.rngs[src]
            self.game_id[src] = -1
        self.engine.shrink(n_new)
        self.G = n_new
        self.game_id, self.color, self.rngs = self.game_id[:n_new], self.color[:n_new], self.rngs[:n_new]
        log.debug("compacted the lockstep batch to %d slots (%d games running)", n_new, len(act))

    def play_move(self):
        """search_move + the two pushes for every running game (``sims`` lockstep steps)."""
        self.begin_move()
        self.engine.run_steps(self.sims)
        self._sims_in_move = self.sims
        self._draw_noise_ahead()                 # the steps are enqueued; the GPU is busy with them
        return self.end_move() * self.sims

    def run(self, n_games=None, max_moves=None):
        """Play until ``n_games`` records exist on this rank (or ``max_moves`` move rounds)."""
        rounds = 0
        while self.active().any():
            if n_games is not None and len(self.finished) >= n_games:
                break
            if max_moves is not None and rounds >= max_moves:
                break
            self.play_move()
            rounds += 1
        return self.finished

    # ---- rolling rounds ---------------------------------------------------------------------
    def _round_share(self, r):
        """How many games of round r this rank plays (ids congruent to rank mod world, below
        total_games)."""
        lo, hi = r * self.round_size, (r + 1) * self.round_size
        if self.total_games is not None:
            hi = min(hi, self.total_games)
        first = lo + (self.rank - lo) % self.world
        return max(0, (hi - first + self.world - 1) // self.world)

    def rounds_complete(self):
        """Number of leading rounds whose games (this rank's share) have all finished."""
        r = 0
        last = None if self.total_games is None else (self.total_games + self.round_size - 1) // self.round_size
        while (last is None or r < last) and self._round_done.get(r, 0) >= self._round_share(r):
            if last is None and self._round_share(r) == 0:
                break
            r += 1
        return r

    def take_round(self, r):
        """Remove and return the finished records of round r."""
        lo, hi = r * self.round_size, (r + 1) * self.round_size
        mine = [x for x in self.finished if lo <= x.game_id < hi]
        self.finished = [x for x in self.finished if not lo <= x.game_id < hi]
        return mine

    def run_rolling(self, n_rounds, on_round=None, sync_every=8, poll=None, on_news=None, idle=None, news="max"):
        """The reference's ``play N games, train, repeat`` (selfplay.py:142-163) without its tail: a
        lockstep batch that stops refilling when a round's last game has STARTED runs ever emptier
        until that game ends (game lengths spread 9...788 plies; measured: 28 % of a 4096-game
        round's time).  Here the freed slots take the NEXT round's games at once; ``on_round(r,
        records)`` runs -- at a move boundary -- as soon as the last game of round r has finished.
        It may train and rewrite the evaluator's weights in place (the captured hipGraph stays
        valid): the games of round r+1 already under way continue on the new weights, the
        asynchronous self-play of AlphaZero instead of the reference's stop-and-train.  With
        ``total_games = n_rounds * round_size`` only the last round has a thinning tail.

        Ranks agree on completed rounds with one small all_reduce(MIN) every ``sync_every`` moves
        (~15 s apart at C3; nothing on the simulation path), so ``on_round`` may use collectives.

        ``poll()`` / ``on_news(k)``: a rank may have news for all ranks that arrives at no particular
        move -- rank 0's background trainer has finished another weight set.  ``poll()`` returns this
        rank's monotonic news counter (0 on ranks that never have any); the same all_reduce carries its
        maximum, and when that rises every rank calls ``on_news(k)`` at the SAME sync index (a
        collective inside it -- the weight broadcast -- is therefore safe) while nobody waited for it:
        every rank kept playing until the news was there.

        ``news="min"``: the counter every rank has REACHED (each rank has a trainer of its own, data-parallel
        training) instead of the highest any rank has.

        ``idle(seconds_of_the_last_move)``: called after every move; a rank that shares its GPU with something
        else (rank 0's background trainer) may pause there to hand it a share of the device."""
        if not self.round_size:
            raise ValueError("run_rolling needs round_size")
        if self.round_size < self.world:
            raise ValueError("round_size %d is smaller than the number of ranks %d: a rank without a share "
                             "of a round could never report it complete" % (self.round_size, self.world))
        done, moves = 0, 0
        self._news_seen = getattr(self, "_news_seen", 0)
        while done < n_rounds:
            t_move = time.perf_counter()
            if self.active().any():
                self.play_move()
            if idle is not None:
                idle(time.perf_counter() - t_move)
            moves += 1                               # (a rank whose batch ran dry idles to the next sync)
            agreed, any_active, seen = self._agree_rounds(moves, sync_every, done, poll, news)
            if on_news is not None and seen > self._news_seen:
                self._news_seen = seen
                on_news(seen)
            while done < min(agreed, n_rounds):
                recs = self.take_round(done)
                if on_round is not None:
                    on_round(done, recs)
                done += 1
            if not any_active and agreed <= done:
                break                                # nothing is running and no further round is complete
        if done < min(n_rounds, self.rounds_complete()):
            raise RuntimeError("run_rolling ended with %d rounds handed over of %d complete" % (done, self.rounds_complete()))
        return done

    def sync_news(self, poll, on_news, news="max"):
        """One agreement on the news counter outside the move loop (after the last round: the trainer's
        remaining weight sets).  Every rank calls it; returns the agreed counter."""
        _, _, seen = self._agree_rounds(0, 1, 0, poll, news)
        if on_news is not None and seen > getattr(self, "_news_seen", 0):
            self._news_seen = seen
            on_news(seen)
        return seen

    def _agree_rounds(self, moves, sync_every, done, poll=None, news="max"):
        """(rounds complete on EVERY rank, is any rank still playing, news counter: the highest of any rank, or
        with news="min" the one every rank has reached) -- as of the last sync."""
        local, active = self.rounds_complete(), bool(self.active().any())
        failure = getattr(self, "poll_failure", None)
        mine = getattr(self, "_news_mine", 0)
        if poll is not None and failure is None:
            try:
                mine = self._news_mine = int(poll())
            except Exception as e:                   # e.g. rank 0's background trainer died: the other ranks sit in
                failure = self.poll_failure = e      # the next all_reduce -- tell them there instead of leaving them
        if self.world == 1:                          # to the process group's timeout
            if failure is not None:
                raise failure
            return local, active, mine
        if moves % sync_every:
            return done, True, getattr(self, "_news_seen", 0)
        import torch
        import torch.distributed as dist
        dev = self.engine.dev if dist.get_backend() == "nccl" else torch.device("cpu")
        # word 4: the arithmetic this rank's tower runs (f16 < hybrid < f16x3).  A rank whose run-time guard has left
        # an auto-kept f16 must not play on beside ranks that are still in it with nothing recording that:
        # the strictest mode of any rank travels here and the others follow at this sync index.
        t = torch.tensor([local, -int(active), -mine if news == "max" else mine, -int(failure is not None),
                          -self._mode_index()], dtype=torch.int64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        if t[3].item() < 0:                          # every rank leaves at the same sync index, non-zero
            if failure is not None:
                raise failure
            raise RuntimeError("another rank reported a failure (its background trainer) in the periodic all_reduce")
        self._follow_strictest(int(-t[4].item()))
        return int(t[0].item()), bool(t[1].item() < 0), int(-t[2].item() if news == "max" else t[2].item())

    _MODE_ORDER = ("f16", "hybrid", "f16x3")

    def _mode_index(self):
        ev = getattr(getattr(self, "engine", None), "evaluator", None)
        p = getattr(ev, "precision", None)
        return self._MODE_ORDER.index(p) if p in self._MODE_ORDER else 0

    def _follow_strictest(self, strictest):
        """Another rank runs a stricter tower arithmetic than this one (its guard fired): leave an auto-kept f16 too."""
        if strictest <= self._mode_index():
            return
        ev = getattr(getattr(self, "engine", None), "evaluator", None)
        enter = getattr(ev, "enter_strict", None)
        if enter is not None and enter("another rank's run-time guard left f16 (agreed in the periodic all_reduce)"):
            self.mode_follows = getattr(self, "mode_follows", 0) + 1
            log.warning("rank %d: tower precision -> %s, following the strictest rank", self.rank, ev.precision)

    def close(self):
        self.engine.close()


def trainable_records(records):
    """The records a training round learns from: at least one move and a RESULT.  A game the runner cut off at
    ``max_plies`` has none (``Game.get_result()`` of a running game, game.py:92-109) -- whether it still carries
    ``GameRecord.truncated`` (in process, over the wire) or was stored and reloaded (``get_history()`` /
    gameplays.json keep only ``result: null``): ``DataGameSequence`` would refuse it as an unfinished game."""
    keep = []
    for r in records:
        h = r.get_history()
        if len(h["moves"]) > 0 and h.get("result") is not None and not getattr(r, "truncated", False):
            keep.append(r)
    return keep


def train_model_job(model, records, model_path, model_dir, epochs=1, batch_size=1):
    """selfplay.py:98-108: train on the recorded games and save the weights.  ``records`` are
    ``GameRecord``s (or ``Game``s): anything with ``get_history()``."""
    from .dataset import DatasetGame
    from .netencoder import DataGameSequence
    data_train = DatasetGame(trainable_records(records))
    if len(data_train) == 0:
        return None
    gen = DataGameSequence(data_train, batch_size=batch_size, random_flips=.1)   # agent.py:81-83
    model.reset_optimizer()             # the reference trains each round in a fresh process
    history = model.train_generator(gen, epochs=epochs, logdir=model_dir)
    model.save_weights(model_path)
    return history


def train_weights(weights, records, device, model_dir=None, epochs=1, batch_size=1, group=None):
    """The arithmetic of ``train_model_job`` without a ``ChessModel``: (trained weight dict, history) from
    a weight dict and the recorded games, on whatever stream is current -- what the background trainer runs
    beside the self-play that keeps using the model.  With ``group`` (a process group of trainer threads, one
    per rank) the ranks train together on their own games (``train.fit_data_parallel``)."""
    from .dataset import DatasetGame
    from .netencoder import DataGameSequence
    from .train import Trainer
    data_train = DatasetGame(trainable_records(records))
    if len(data_train) == 0 and group is None:
        return weights, None
    gen = DataGameSequence(data_train, batch_size=batch_size, random_flips=.1)   # agent.py:81-83
    trainer = Trainer(weights, device)              # a fresh optimizer per round, as in the reference
    log_fn = None
    if model_dir is not None:
        import json

        def log_fn(summary):
            with open(os.path.join(model_dir, "train_log.jsonl"), "a") as f:
                f.write(json.dumps(summary) + "\n")
    if group is not None:
        from .train import fit_data_parallel
        history = fit_data_parallel(trainer, gen, group=group, epochs=epochs, log=log_fn)
    else:
        history = trainer.fit_generator(gen, epochs=epochs, log=log_fn)
    return trainer.weights(), history


TRAINER_STREAM_PRIORITY = 0       # the background trainer's HIP stream (a high priority, -1, was measured: no effect)


class BackgroundTrainer(object):
    """Rank 0's trainer on a thread and a HIP stream of its own.

    The reference's loop is play -> train -> play in one process (selfplay.py:142-163).  With rolling
    rounds on several GPUs a blocking trainer stalls EVERY rank at the weight broadcast (8 x 4096 C3 games
    are ~416 s of training per ~300 s of play).  Here ``submit(round, records)`` returns at once; the
    thread trains the rounds in order, each from the weights the previous one produced (a fresh optimizer
    per round, as every ``train_model_job`` of the reference starts one), writes the weight file, and
    ``ready()`` -- a monotonic count of finished weight sets -- is what the ranks agree on in their
    periodic all_reduce (``SelfPlayRunner.run_rolling(poll=..., on_news=...)``).  The inference tensors
    are only ever rewritten by the main thread at a move boundary (``ChessModel.load_dict``), never here.
    ``train_fn(weights, records) -> (weights, history)`` defaults to ``train_weights`` on ``device``."""

    def __init__(self, weights, device=None, model_path=None, model_dir=None, train_fn=None, group=None):
        import queue
        import threading
        self._q = queue.Queue()
        self._lock = threading.Lock()
        self._weights = weights
        self._sets = {}                      # finished count -> that weight set (until taken)
        self.group = group                   # data-parallel: the trainer threads' own process group
        self._done = []                      # (round, seconds, last history entry)
        self.error = None
        self.device, self.model_path, self.model_dir = device, model_path, model_dir
        self._train_fn = train_fn
        self._thread = threading.Thread(target=self._loop, name="crl-trainer", daemon=True)
        self._thread.start()

    def submit(self, rnd, records):
        self._raise()
        self._q.put((rnd, list(records)))

    def busy(self):
        """A round is being trained or waiting to be."""
        return self._q.unfinished_tasks > 0

    def ready(self):
        """Number of weight sets finished so far (monotonic)."""
        self._raise()
        with self._lock:
            return len(self._done)

    def latest(self):
        with self._lock:
            return self._weights, list(self._done)

    def take(self, k):
        """Weight set number ``k`` (1 = after the first trained round); older sets are dropped.  In the
        data-parallel mode every rank loads the SAME set at the same sync index, whatever its own trainer has
        finished since."""
        with self._lock:
            w = self._sets[k]
            for old in [c for c in self._sets if c <= k]:
                del self._sets[old]
            return w

    def drain(self):
        """Block until every submitted round is trained."""
        self._q.join()
        self._raise()

    def close(self):
        self._q.put(None)
        self._thread.join()
        self._raise()

    def _raise(self):
        if self.error is not None:
            raise RuntimeError("the background trainer failed") from self.error

    def _loop(self):
        stream = None
        if self._train_fn is None:
            import torch
            torch.cuda.set_device(self.device)
            # The trainer's kernels are thousands of small dependent launches; beside a lockstep batch whose trunk
            # launches fill every CU (all of its LDS and registers) for 1.1 ms at a time they are dispatched a
            # few per trunk launch: 305 s for a round that takes 15 s alone (profiles/r04/
            # rolling_probe_train_rounds1024.json), 23 s instead of 5.9 s beside a batch whose host syncs every 8
            # steps, the same at stream priority -1 (tools/trainer_share_probe.py).  What gives it the device is
            # the self-play's pause after a move (--trainer-share).
            stream = torch.cuda.Stream(self.device, priority=TRAINER_STREAM_PRIORITY)
        while True:
            item = self._q.get()
            try:
                if item is None:
                    return
                rnd, records = item
                t0 = time.perf_counter()
                if self._train_fn is not None:
                    new, hist = self._train_fn(self._weights, records)
                else:
                    import torch
                    with torch.cuda.stream(stream):
                        new, hist = train_weights(self._weights, records, self.device, self.model_dir, group=self.group)
                    stream.synchronize()
                if self.model_path is not None:
                    if str(self.model_path).endswith((".h5", ".hdf5")):
                        from .keras_h5 import save_keras_h5
                        save_keras_h5(new, self.model_path)
                    else:
                        np.savez(self.model_path, **new)
                with self._lock:
                    self._weights = new
                    self._done.append((rnd, time.perf_counter() - t0, hist[-1] if hist else None))
                    if self.group is not None:
                        self._sets[len(self._done)] = new
                log.info("trainer: round %d trained on %d games in %.1fs", rnd, len(records), time.perf_counter() - t0)
            except BaseException as e:                    # surfaces at the next submit / ready / drain
                self.error = e
            finally:
                self._q.task_done()


def main(argv=None):
    parser = argparse.ArgumentParser(description="Plays self-play chess games on MI355X GPUs, "
                                     "stores the game records and trains the model on them.")
    parser.add_argument("model_dir", metavar="modeldir",
                        help="where to load the model from and store the records")
    parser.add_argument("--games", type=int, default=1)
    parser.add_argument("--threads", type=int, default=6,
                        help="accepted for compatibility (simulations are sequential per game)")
    parser.add_argument("--debug", action="store_true", default=False)
    parser.add_argument("--sims", type=int, default=900, help="MCTS iterations per move")
    parser.add_argument("--parallel", type=int, default=None, help="games in lockstep per GPU")
    parser.add_argument("--blocks", type=int, default=10)
    parser.add_argument("--filters", type=int, default=256)
    parser.add_argument("--seed", type=int, default=0)
    parser.add_argument("--no-noise", action="store_true")
    parser.add_argument("--rounds", type=int, default=1, help="play --games games, train, repeat")
    parser.add_argument("--no-train", action="store_true", help="only play and store the records")
    parser.add_argument("--rolling", action="store_true",
                        help="overlap the rounds: freed slots take the next round's games while a round's "
                             "long games finish; a round is trained on as soon as its last game ends -- on "
                             "rank 0, in the background, while every rank keeps playing -- and the games "
                             "under way continue on the new weights once they are there")
    parser.add_argument("--train-mode", choices=["rank0", "dp"], default="rank0",
                        help="--rolling on several GPUs: 'rank0' = rank 0 trains on every rank's games, one Adam step "
                             "per game in the reference's order (selfplay.py:98-108; one GPU trains ~25 k positions/s, "
                             "eight produce ~35 k/s at C3: the weights lag); 'dp' = every rank trains on ITS games, "
                             "gradients averaged over RCCL, one Adam step per <ranks> games -- not the reference's "
                             "arithmetic, but a trainer that scales with the GPUs.  EXPERIMENTAL: exercised on gloo ranks "
                             "only; on the nccl backend two communicators are driven from two unordered threads of one "
                             "process, which RCCL does not promise to survive")
    parser.add_argument("--trainer-share", type=float, default=0.2,
                        help="--rolling: fraction of rank 0's wall time its self-play leaves to the background trainer "
                             "WHILE a round is waiting to be trained (a pause after every move; the trainer's thousands "
                             "of small dependent kernels get almost nothing of the GPU beside a full lockstep batch: "
                             "305 s for a round that takes 15 s alone).  C3 on one GPU needs ~0.2 to keep pace; 0 = "
                             "self-play first, the trainer takes what is left")
    parser.add_argument("--max-plies", type=int, default=4096,
                        help="longest game record; a game still running there is handed over unfinished "
                             "(result None) and its slot refilled")
    parser.add_argument("--dist-timeout-min", type=float, default=360.0,
                        help="process-group timeout in minutes (the default RCCL watchdog of 10 minutes is "
                             "shorter than one training round of a large net)")
    parser.add_argument("--precision", choices=["auto", "f16", "f16x3", "hybrid"], default="auto",
                        help="arithmetic of the fused HIP tower: 'f16' = one fp16 MFMA per product (fastest; "
                             "1e-3 of fp32 only for soft nets), 'f16x3' = split operands, fp32-grade, ~3x the "
                             "tower time, 'hybrid' = f16x3 for priors and values, f16 with an f16x3 fall-back for "
                             "the reply choice, 'auto' = f16 where a probe shows it within the bar, else hybrid")
    parser.add_argument("--numpy-promotion", choices=["auto", "nep50", "legacy"], default="auto",
                        help="arithmetic of the PUCT term 10 * prior (mctree.py:79-87): 'legacy' = the float64 "
                             "product of the reference's pinned numpy 1.17.2, 'nep50' = the float32 product "
                             "of numpy >= 2, 'auto' = whichever the installed numpy computes")
    args = parser.parse_args(argv)
    logging.basicConfig(level=logging.DEBUG if args.debug else logging.INFO)

    import datetime
    import sys
    import torch
    import torch.distributed as dist
    from .model import ChessModel
    from .records import dumps, gather_records
    rank, world, local = 0, 1, 0
    if "RANK" in os.environ and int(os.environ.get("WORLD_SIZE", "1")) > 1:
        rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
        local = int(os.environ.get("LOCAL_RANK", rank))
        from . import multiprocess_env
        multiprocess_env()                        # dmabuf IPC for RCCL, before the first GPU call of this process
        # self-test hooks for a 1-GPU box: CRL_DEVICE pins every rank to one device and
        # CRL_DIST_BACKEND=gloo replaces RCCL, so the multi-rank control flow can be exercised there
        local = int(os.environ.get("CRL_DEVICE", local))
        torch.cuda.set_device(local)
        # an explicit timeout: in the stop-and-train mode the other ranks wait in the weight broadcast
        # for as long as rank 0 trains (20x256: ~1 400 s per 8 x 4096-game round; the watchdog default is 600 s)
        dist.init_process_group(os.environ.get("CRL_DIST_BACKEND", "nccl"),
                                timeout=datetime.timedelta(minutes=args.dist_timeout_min))
    os.makedirs(args.model_dir, exist_ok=True)
    path = get_model_path(args.model_dir)
    weights = path if os.path.exists(path) else None
    model = ChessModel(compile_model=not args.no_train, weights=weights, blocks=args.blocks,
                       filters=args.filters, device="cuda:%d" % local, seed=args.seed, precision=args.precision)
    per_rank = (args.games + world - 1) // world
    parallel = args.parallel or min(per_rank, 4096)
    allrecs = []
    max_plies = args.max_plies
    dp = args.train_mode == "dp" and world > 1 and args.rolling and not args.no_train
    # (data-parallel: the trainer THREADS of all ranks talk over a process group of their own -- a communicator
    # serves one thread at a time, and the main threads keep using the default group)
    train_group = dist.new_group(backend=dist.get_backend(), timeout=datetime.timedelta(minutes=args.dist_timeout_min)) if dp else None
    if dp and dist.get_backend() == "nccl" and rank == 0:
        print("selfplay: --train-mode dp is EXPERIMENTAL on the nccl backend (two communicators driven from two "
              "unordered threads per process; never run on RCCL ranks) -- 'rank0' is the reference's arithmetic",
              file=sys.stderr, flush=True)
    background = (BackgroundTrainer(model.weights, "cuda:%d" % local, path if rank == 0 else None, args.model_dir if rank == 0 else None,
                                    group=train_group)
                  if (args.rolling and not args.no_train and (rank == 0 or dp)) else None)

    def store(rnd, recs):
        """Gather the round's records (RCCL) and store them (rank 0)."""
        newrecs = gather_records(recs, max_plies)
        allrecs.extend(newrecs)
        if rank == 0:
            with open(os.path.join(args.model_dir, "gameplays.json"), "w") as f:
                f.write(dumps(allrecs))
            log.info("round %d: wrote %d game records", rnd, len(allrecs))
        return newrecs

    def after_round(rnd, recs):
        """Stop-and-train (the reference's order, selfplay.py:142-163): gather, store, train on rank 0,
        ship the weights; every rank waits for them."""
        newrecs = store(rnd, recs)
        if not args.no_train:
            # the reference trains in ONE process; here rank 0 trains on every rank's games and the
            # new weights go to the other ranks over RCCL (one broadcast per round).  The inference
            # tensors are rewritten in place: hipGraphs of running engines stay valid
            if rank == 0:
                t0 = time.perf_counter()
                hist = train_model_job(model, newrecs, path, args.model_dir)
                log.info("round %d: trained on %d games in %.1fs: %s", rnd, len(newrecs),
                         time.perf_counter() - t0, hist[-1] if hist else None)
            if world > 1:
                from .train import broadcast_weights
                model.load_dict(broadcast_weights(model.weights, "cuda:%d" % local, src=0))

    def after_round_rolling(rnd, recs):
        """Rolling rounds: gather, store, hand the round to rank 0's background trainer and play on."""
        newrecs = store(rnd, recs)
        if background is not None:
            background.submit(rnd, recs if dp else newrecs)        # dp: every rank trains on its own share

    def new_weights(k):
        """Every rank, at the same sync index: rank 0's newest trained weights into the inference tensors
        (one flat broadcast; in place, so the captured hipGraph stays valid)."""
        if dp:
            w = background.take(k)           # every rank's trainer produced the same set k
        else:
            w = background.latest()[0] if background is not None else model.weights
            if world > 1:
                from .train import broadcast_weights
                w = broadcast_weights(w, "cuda:%d" % local, src=0)
        model.load_dict(w)
        log.info("rank %d: weight set %d loaded", rank, k)

    if args.rolling:
        # rounds overlap: the batch keeps refilling from the next round's ids while a round's long
        # games finish (SelfPlayRunner.run_rolling); game ids run on across rounds
        runner = SelfPlayRunner(model, parallel, args.sims, seed=args.seed, noise=not args.no_noise,
                                rank=rank, world=world, device=local, max_plies=max_plies,
                                total_games=args.games * args.rounds, round_size=args.games,
                                numpy_promotion=args.numpy_promotion)
        t0 = time.perf_counter()
        poll = (lambda: background.ready() if background is not None else 0) if not args.no_train else None
        share = min(max(args.trainer_share, 0.0), 0.9)

        def idle(move_seconds):
            """rank 0, while its trainer has work: hand it ``share`` of the wall time (alone on the device it
            trains ~4x faster than between the self-play's launches)"""
            if background is not None and share > 0 and background.busy():
                time.sleep(move_seconds * share / (1.0 - share))

        runner.run_rolling(args.rounds, on_round=after_round_rolling, poll=poll,
                           on_news=new_weights if not args.no_train else None, idle=idle, news="min" if dp else "max")
        dt = time.perf_counter() - t0
        log.info("rank %d: %d rolling rounds of %d games, %d sims in %.1fs (%.0f sims/s)", rank, args.rounds,
                 args.games, runner.sims_run, dt, runner.sims_run / max(dt, 1e-9))
        if not args.no_train:
            if background is not None:
                try:
                    background.drain()                # the last rounds' training (the other ranks wait in sync_news)
                except RuntimeError as e:
                    runner.poll_failure = e           # ... and hear of a failure there
            runner.sync_news(poll, new_weights, news="min" if dp else "max")
            if background is not None:
                background.close()
        runner.close()
    else:
        for rnd in range(args.rounds):
            runner = SelfPlayRunner(model, parallel, args.sims, seed=args.seed + rnd * args.games,
                                    noise=not args.no_noise, rank=rank, world=world, device=local,
                                    max_plies=max_plies, total_games=args.games,
                                    numpy_promotion=args.numpy_promotion)
            t0 = time.perf_counter()
            recs = runner.run()
            dt = time.perf_counter() - t0
            log.info("round %d rank %d: %d games, %d sims in %.1fs (%.0f sims/s)", rnd, rank, len(recs),
                     runner.sims_run, dt, runner.sims_run / max(dt, 1e-9))
            runner.close()
            after_round(rnd, recs)
    import hashlib
    digest = hashlib.sha1(b"".join(np.ascontiguousarray(model.weights[k]).tobytes() for k in sorted(model.weights))).hexdigest()
    log.info("rank %d: weights at the end %s (%s); precision guard: %s", rank, digest[:16], model.precision,
             getattr(model, "guard", None))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
