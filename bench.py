#!/usr/bin/env python3
"""bench.py -- MCTS simulations/sec of the lockstep self-play loop on MI355X.

Contract (driver): ``python bench.py --gpus N --steps K --warmup W``.  For N > 1 the driver
launches it through torch.distributed.run, one rank per GPU; run by hand with ``--gpus N`` and
no WORLD_SIZE in the environment it starts the N rank processes itself (ordinary child
processes, started before this process touches a GPU) and exits with their status.  Either way
every rank checks ``--gpus`` against the process group's world size and the ranks exchange one
all_reduce before anything is timed: a line with ``n_gpus: N`` was produced by N ranks.

One *step* is one lockstep pass of the hot path: one MCTS simulation (select -> expand S1 ->
tower -> reply S2 -> tower -> backup) for EACH of the G games resident on the GPU, move
boundaries (fresh tree, root priors, host compute_policy, two pushes) included whenever a
game's budget of simulations completes.  ``value`` = simulations completed by all ranks / wall
time of the K timed steps (max over ranks), state resident in HBM throughout.  When K is
shorter than one move the window is placed mid-move (the trees are pre-grown un-timed), so a
short run sees trees of representative depth instead of 800 root expansions; such a window holds
no move boundary, so one boundary (end_move + begin_move of a full-length move) is timed right
after it and the line carries ``move_boundary`` and ``value_incl_boundaries`` = G x S / (S x
ms_per_step + boundary ms), the rate a run of whole moves sustains.

Workload (BASELINE.json metric "MCTS simulations/sec at 800 sims/move", config C3 -- fits one
GPU): 4096 games in lockstep per GPU, 800 sims/move, 10-block/128-filter random-init tower,
fp16 MFMA trunk, all games from the standard position, Dirichlet noise on.  Games are
independent: ranks share nothing on the hot path (weak scaling); for N > 1 the line also
carries ``record_gather``: the RCCL all_gather of every rank's game records (C4's only
collective), timed after the measured region.

Extra objects on the JSON line: ``roofline`` for the dominant kernel (the fused trunk,
MFMA-bound), ``roofline_tree`` for the hand-written HIP search kernels (HBM-bound),
``cpu_baseline`` for the reference-shaped CPU port (oracle/, config C1) timed on this box's
host cores on rank 0 at N=1.  ``traffic`` figures come from the tracked PMC table
profiles/pmc_traffic.json (rocprofv3 passes of exactly the named kernel and shape); they are
null when no pass exists for the kernel that ran.
"""
import argparse
import contextlib
import json
import os
import socket
import subprocess
import sys
import time

# The pool's host driver only supports dmabuf IPC: without this RCCL (and any CUDA-tensor sharing
# between processes) fails with "hipIpcGetMemHandle: invalid argument".  The image exports it; it
# is set here as well, before anything initialises HSA, so that the launcher path (torchrun) and the
# self-spawn path (--gpus N by hand) run their ranks in the same environment.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_PEAK_TFLOPS = {"f16": 2500.0, "bf16": 2500.0, "f32": 157.3}   # MI355X_MICROARCH.md, dense
HBM_PEAK_GBS = 8000.0
PMC_TABLE = os.path.join(ROOT, "profiles", "pmc_traffic.json")


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=1600)
    p.add_argument("--warmup", type=int, default=200)
    p.add_argument("--games", type=int, default=4096, help="games in lockstep per GPU")
    p.add_argument("--sims", type=int, default=800)
    p.add_argument("--blocks", type=int, default=10)
    p.add_argument("--filters", type=int, default=128)
    p.add_argument("--dtype", default="f16", choices=["f16", "bf16", "f32"])
    p.add_argument("--seed", type=int, default=0)
    p.add_argument("--no-graph", action="store_true")
    p.add_argument("--no-fused", action="store_true", help="PyTorch-ROCm trunk instead of the HIP kernel")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--cpu-seconds", type=float, default=15.0)
    p.add_argument("--precision", default="auto", choices=["auto", "f16", "f16x3", "hybrid"],
                   help="fused-trunk arithmetic of the timed region.  auto (default) = what the product picks for "
                        "these weights (ChessModel's probe), and the mode that was timed is then held to the 1e-3 "
                        "bar against the fp32 tower oracle on positions of this run's own games "
                        "(tower_error_vs_fp32); if it misses the bar the window is timed again in f16x3 and THAT "
                        "is `value`.  f16 = one fp16 MFMA per product, BASELINE's 'fp16 MFMA inference'; f16x3 = "
                        "hi/lo split operands, three MFMAs, fp32-grade; hybrid = f16x3 for every output under the "
                        "1e-3 bar (priors and value of S2), f16 with an f16x3 fall-back for the reply choice of S1 "
                        "(what auto picks when f16 misses its tolerance).  precision_modes carries all three rates")
    p.add_argument("--strict-steps", type=int, default=40,
                   help="steps of the other precision mode's leg timed after the main window (0 = skip)")
    p.add_argument("--parity-positions", type=int, default=4096,
                   help="positions of complete self-play games (played with the timed weights) on which the timed "
                        "tower mode is compared with the fp32 oracle after the timed region; 0 = skip "
                        "(tower_error_vs_fp32 null, no claim)")
    p.add_argument("--weights", default=None,
                   help="evaluate a weight file (.npz / Keras .h5; --blocks / --filters are taken from it) instead of a "
                        "random-init net: NOT the metric's configuration (BASELINE quotes random-init nets) -- for "
                        "measuring what the product runs once a net is trained; the line says so in config.workload")
    p.add_argument("--dist-timeout-min", type=float, default=30.0,
                   help="process-group timeout (N > 1): rank 0's post-window work -- parity gate against the fp32 "
                        "oracle, phase profile, the other precision modes' legs -- runs while the other ranks wait in "
                        "a collective; explicit so that no configuration depends on the backend's default watchdog")
    p.add_argument("--steps-per-graph", type=int, default=None,
                   help="lockstep steps captured into ONE hipGraph launch (default: LockstepEngine.STEPS_PER_GRAPH = 8).  "
                        "1 keeps a C5 hybrid step at ~130 kernel nodes per graph, which rocprofv3's kernel tracing "
                        "survives (it segfaults inside hipGraphLaunch at the ~1000 nodes of eight such steps)")
    p.add_argument("--graph-phase-steps", type=int, default=64,
                   help="steps of the stamped leg behind the timed window: the same step captured WITH one-thread stamp "
                        "kernels between its phases and around every trunk launch, replayed as a hipGraph -- phase and "
                        "kernel times of roofline.step_fit come from the replayed graph (0 = skip: eager HIP events only)")
    p.add_argument("--opening-moves", type=int, default=6,
                   help="shortened, noisy moves every game plays (un-timed) before the window, so that the games are on "
                        "lines of their own as in every later move of a real run (see play_opening)")
    p.add_argument("--opening-sims", type=int, default=32, help="simulations of each opening move")
    p.add_argument("--gph-seconds", type=float, default=14.0,
                   help="seconds of the games/hour leg behind the timed region (rank 0, N = 1): COMPLETE games at C2's size "
                        "(512 in lockstep, 100 sims/move, 6x64 random-init, refill), the first quarter un-counted while "
                        "the batch mixes; 0 = skip")
    p.add_argument("--numpy-promotion", default="auto", choices=["auto", "nep50", "legacy"],
                   help="arithmetic of the PUCT term 10 * prior (mctree.py:79-87); auto = the installed numpy's")
    return p.parse_args()


# ---- launching the ranks -------------------------------------------------------------------------
def visible_gpus():
    """Number of GPUs the ranks will see, WITHOUT initialising HIP/HSA in this process (the launcher
    parent must not touch the GPU: it only starts children): the KFD topology nodes that have SIMDs,
    narrowed by HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES when set.  None when /sys has no KFD."""
    base = "/sys/class/kfd/kfd/topology/nodes"
    try:
        nodes = sorted(os.listdir(base), key=int)
    except (OSError, ValueError):
        return None
    n, readable = 0, 0
    for node in nodes:
        try:
            props = dict(line.split()[:2] for line in open(os.path.join(base, node, "properties")) if line.strip())
        except (OSError, ValueError):
            continue
        readable += 1
        if int(props.get("simd_count", "0")) > 0:
            n += 1
    if readable == 0:
        return None                                          # topology not readable here: let the ranks find out
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([x for x in v.split(",") if x.strip() != ""]))
    return n


def spawn_ranks(n):
    """``--gpus n`` without a launcher: start n rank processes of this script (ordinary children;
    this parent makes no GPU call at all, before or after) and return the worst exit status."""
    if "CRL_BENCH_DEVICE" not in os.environ and os.environ.get("CRL_BENCH_DRYRUN") != "1":
        have = visible_gpus()
        if have is not None and have < n:
            print("bench.py: --gpus %d but only %d GPU(s) visible" % (n, have), file=sys.stderr)
            return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    try:
        live = list(procs)
        while live:
            for p in list(live):
                code = p.poll()
                if code is None:
                    continue
                live.remove(p)
                if code != 0:                              # one rank died: the others would hang in RCCL
                    rc = rc or code
                    for q in live:
                        q.terminate()
            time.sleep(0.2)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return rc


@contextlib.contextmanager
def stdout_to_stderr():
    """RCCL prints a version banner on STDOUT when a communicator is created; stdout carries the
    one JSON line, so the banner is sent to stderr (file-descriptor level: it comes from C)."""
    sys.stdout.flush()
    saved = os.dup(1)
    os.dup2(2, 1)
    try:
        yield
    finally:
        sys.stdout.flush()
        os.dup2(saved, 1)
        os.close(saved)


def init_ranks(a):
    """(rank, world, local device, dist or None): joins the process group the launcher described
    and proves that ``--gpus`` ranks are really there."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # self-test hooks: CRL_BENCH_DEVICE pins every rank to one device (1-GPU box) and
    # CRL_BENCH_BACKEND=gloo replaces RCCL, so the multi-rank control flow can be exercised anywhere
    if "CRL_BENCH_DEVICE" in os.environ:
        local = int(os.environ["CRL_BENCH_DEVICE"])
    if a.gpus != world:
        raise SystemExit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks" % (a.gpus, world))
    # CRL_BENCH_FORCE_GROUP=1 (self-test on a 1-GPU box): a world of one still creates the process
    # group, so the RCCL branch below, the reductions and the record gather execute on hardware
    if world == 1 and os.environ.get("CRL_BENCH_FORCE_GROUP") != "1":
        return rank, world, local, None
    import datetime
    import torch.distributed as dist
    backend = os.environ.get("CRL_BENCH_BACKEND", "nccl")
    timeout = datetime.timedelta(minutes=a.dist_timeout_min)
    with stdout_to_stderr():
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local), timeout=timeout)
            dev = torch.device("cuda", local)
        else:
            dist.init_process_group(backend, timeout=timeout)
            dev = torch.device("cpu")
        t = torch.tensor([float(rank + 1)], dtype=torch.float64, device=dev)
        dist.all_reduce(t)
        if dev.type == "cuda":
            torch.cuda.synchronize()
    if dist.get_world_size() != a.gpus or t.item() != a.gpus * (a.gpus + 1) / 2:
        raise SystemExit("bench.py: process group has %d ranks (all_reduce check %.1f), --gpus %d"
                         % (dist.get_world_size(), t.item(), a.gpus))
    return rank, world, local, dist


# ---- measurement helpers ---------------------------------------------------------------------------
def event_time_ms(fn, reps):
    """Average duration of fn() on torch's current stream -- the stream the HIP kernels are
    launched on (LockstepEngine binds the context to it) -- HIP events around `reps` calls."""
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    fn()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


class NoFenceEvent(object):
    """A HIP timing event created with hipEventDisableSystemFence, with torch.cuda.Event's record / elapsed_time.
    A default event performs a system-scope release when it is recorded -- the L2's dirty lines are written back -- which
    is nothing behind the fused trunk kernels (C3: the eager phases add up to 1.015 x the timed step) and ~0.3 ms behind a
    layer of the layer-wise tower that has just written 268 MB of activations (C5: 1.09 x).  These events only take
    timestamps.  Bound through ctypes to the HIP runtime the process has already loaded."""
    _hip = None
    DISABLE_SYSTEM_FENCE = 0x20000000

    @classmethod
    def hip(cls):
        if cls._hip is None:
            import ctypes
            path = next(line.split()[-1] for line in open("/proc/self/maps") if "libamdhip64" in line)
            cls._hip = ctypes.CDLL(path)
        return cls._hip

    def __init__(self, enable_timing=True):
        import ctypes
        self.ev = ctypes.c_void_p()
        if self.hip().hipEventCreateWithFlags(ctypes.byref(self.ev), self.DISABLE_SYSTEM_FENCE) != 0:
            raise RuntimeError("hipEventCreateWithFlags failed")

    def record(self):
        import ctypes
        if self.hip().hipEventRecord(self.ev, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)) != 0:
            raise RuntimeError("hipEventRecord failed")

    def elapsed_time(self, other):
        import ctypes
        ms = ctypes.c_float()
        self.hip().hipEventSynchronize(other.ev)
        if self.hip().hipEventElapsedTime(ctypes.byref(ms), self.ev, other.ev) != 0:
            raise RuntimeError("hipEventElapsedTime failed")
        return ms.value

    def __del__(self):
        try:
            self.hip().hipEventDestroy(self.ev)
        except Exception:
            pass


def timing_event_cls():
    """NoFenceEvent where the HIP runtime can be bound, else torch.cuda.Event."""
    try:
        a, b = NoFenceEvent(), NoFenceEvent()
        a.record()
        b.record()
        torch.cuda.synchronize()
        a.elapsed_time(b)
        return NoFenceEvent
    except Exception:
        return torch.cuda.Event


def profile_phases(run, n):
    """HIP-event time of each phase of a step, measured eagerly (no graph) on `n` real steps in
    the middle of a move, on the stream the kernels are launched on."""
    eng = run.engine
    if run._sims_in_move:                       # mid-move: finish it first
        run.end_move()
    if run._sims_in_move is None:
        run.begin_move()
    for _ in range(max(0, run.sims // 2 - n)):
        eng.step()
    Event = timing_event_cls()
    ev = [[Event(enable_timing=True) for _ in range(5)] for _ in range(n)]
    model = eng.evaluator
    hooked = hasattr(model, "trunk_events")
    if hooked:
        model.trunk_events = []                 # every trunk launch of these steps bracketed by HIP events
        model.trunk_event_cls = Event
    for i in range(n):
        ev[i][0].record()
        eng.phase_select_expand()
        ev[i][1].record()
        eng.phase_tower_s1()
        ev[i][2].record()
        eng.phase_reply()
        ev[i][3].record()
        eng.phase_tower_s2()
        ev[i][4].record()
    torch.cuda.synchronize()
    names = ["select_expand", "tower_s1", "reply", "tower_s2"]
    out = {nm: sum(ev[i][k].elapsed_time(ev[i][k + 1]) for i in range(n)) / n
           for k, nm in enumerate(names)}
    if hooked:
        # the trunk kernel's own time INSIDE real steps (between the search kernels and the heads, at the clock
        # the step holds), per arithmetic: {kind: (mean ms per launch, launches per step)}
        kinds = {}
        for kind, e0, e1 in model.trunk_events:
            kinds.setdefault(kind, []).append(e0.elapsed_time(e1))
        model.trunk_events = None
        out["trunk_in_step"] = {k: {"launch_ms": sum(v) / len(v), "launches_per_step": len(v) / n} for k, v in kinds.items()}
        trunk_ms = sum(sum(v) for v in kinds.values()) / n
        out["heads_and_margin_ms"] = out["tower_s1"] + out["tower_s2"] - trunk_ms
        out["trunk_ms_per_step"] = trunk_ms
        out["events"] = "hipEventDisableSystemFence" if Event is NoFenceEvent else "torch.cuda.Event"
    return out


def stamped_steps(run, ring, n):
    """``n`` (a multiple of the steps per graph launch) lockstep steps of the STAMPED build of the step (engine.set_stamps:
    one-thread crl_stamp kernels between the phases and around every trunk launch, captured into the hipGraph with them),
    continuing the move under way when it has room for them -- the stamped leg then sees the trees the timed window has
    just left -- else in the next move, grown un-stamped to its middle.  Leaves the engine on the plain graphs."""
    eng = run.engine
    if run._sims_in_move is None or run.sims - run._sims_in_move < n + 1:
        if run._sims_in_move:
            run.end_move()                                 # (the games go on: the next move is a real one)
        run.steps(min(run.sims // 2 + 8, run.sims - n - 1))
    eng.set_stamps(ring)
    try:
        eng.run_steps(n)
    finally:
        eng.set_stamps(None)
    run._sims_in_move += n


def graph_phases_begin(run, n):
    """Before the timed window: the ring, the stamped graphs (captured now: nothing is captured between the window and
    the stamped leg) and one warm replay of them.  Returns (ring, steps of the leg) or None without graphs."""
    from chessrl_amd.engine import StampRing
    eng = run.engine
    K = eng.STEPS_PER_GRAPH
    if not eng.use_graph or n <= 0 or run.sims < 4 * K:    # (a move too short to hold the warm replay and the leg)
        return None
    n = max(K, min(n, run.sims // 3) // K * K)
    ring = StampRing((n + 2 * K) * 16, eng.dev)
    stamped_steps(run, ring, K)                            # captures the stamped K-step graph and replays it once
    return ring, n


def graph_phases_end(run, ring, n):
    """Right behind the timed window: ``n`` stamped steps replayed from the hipGraph, the ring read back.  Phase and
    trunk-launch times come from the REPLAYED graph: consecutive stamps telescope to the step, so the parts add up to
    the stamped step exactly, and that step against the un-stamped timed one shows what the stamps cost."""
    from chessrl_amd.engine import summarise_stamps
    eng = run.engine
    torch.cuda.synchronize()
    ring.clear()
    t0 = time.perf_counter()
    stamped_steps(run, ring, n)
    stamps = ring.read()
    wall_ms = (time.perf_counter() - t0) * 1e3 / n
    out = summarise_stamps(stamps)
    out["host_wall_ms_per_step"] = wall_ms
    out["source"] = ("crl_stamp kernels captured into the step's hipGraph (%d steps per graph launch), %d steps replayed "
                     "right behind the timed window, device wall clock at %.0f kHz" % (eng.STEPS_PER_GRAPH, n, ring.ticks_per_ms))
    return out


def graph_phases(run, n):
    """Both halves in one go (the legs behind the main window: another precision mode's phases)."""
    got = graph_phases_begin(run, n)
    if got is None:
        return None
    try:
        return graph_phases_end(run, *got)
    finally:
        run.engine.drop_stamped_graphs()


def pmc_traffic(kernel, shape):
    """HBM-side bytes per launch of `kernel` at `shape` from the tracked PMC table (FETCH_SIZE x 2,
    the gfx950 wide-read correction of MI355X_MICROARCH.md, + WRITE_SIZE), or (None, why)."""
    try:
        table = json.load(open(PMC_TABLE))
    except (OSError, ValueError):
        return None, "no PMC table (%s)" % os.path.relpath(PMC_TABLE, ROOT)
    for e in table.get("passes", []):
        if e["kernel"] == kernel and e["shape"] == shape:
            return 2.0 * e["fetch_size_kb"] * 1e3 + e["write_size_kb"] * 1e3, e["source"]
    return None, "no PMC pass of %s at %s in %s" % (kernel, shape, os.path.relpath(PMC_TABLE, ROOT))


def trunk_kernel_name(F, G, bits):
    """The kernel crl_trunk_forward dispatches for this shape, asked of the library itself
    (crl_trunk_kernel_name), as rocprofv3 prints it."""
    import ctypes
    from chessrl_amd import _lib
    buf = ctypes.create_string_buffer(128)
    rc = _lib.lib().crl_trunk_kernel_name(F, G, int(bits), buf, len(buf))
    if rc != 0:
        raise RuntimeError("crl_trunk_kernel_name failed (%d)" % rc)
    return buf.value.decode()


def time_move_boundary(run):
    """One move boundary as a full-length move ends it: the last backprop, D2H of the root statistics,
    host compute_policy + argmax, the two pushes, harvest/refill (end_move) and the next move's fresh
    trees + root evaluation (begin_move).  The rest of the current move is played un-timed first.
    Host clock between two device synchronisations: the GPU is idle while the host works."""
    if run._sims_in_move is None:
        run.begin_move()
    half = max(1, run.sims // 2)
    while run._sims_in_move < run.sims:
        k = (half if run._sims_in_move < half else run.sims) - run._sims_in_move
        run.engine.run_steps(k)
        run._sims_in_move += k
        if run._sims_in_move == half:
            run._draw_noise_ahead()
    torch.cuda.synchronize()
    n0 = len(run.finished)
    t0 = time.perf_counter()
    run.end_move()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    run.begin_move()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    return {"end_move_ms": (t1 - t0) * 1e3, "begin_move_ms": (t2 - t1) * 1e3, "ms": (t2 - t0) * 1e3,
            # a boundary at which games finished also fetches their records, resets their slots and plays the
            # greedy opening of the new black games; the FIRST such boundary of a process additionally pays
            # one-off allocations (steady state, tools/boundary_probe.py: C3 4.2 ms, C2 1.4 ms)
            "games_finished_at_it": len(run.finished) - n0}


def cpu_baseline(seconds):
    """Reference-shaped CPU self-play (config C1: 1 game, 50 sims/move, tiny random-init net,
    sequential object tree, one tower call per request), bounded to ~`seconds` of CPU work."""
    from oracle import mcts_oracle, tower_oracle
    torch.set_num_threads(1)
    w = tower_oracle.init_weights(2, 32, seed=0)
    agent = mcts_oracle.OracleAgent(tower_oracle.TowerNet(w))
    from oracle.chess_oracle import OracleGame
    gam = OracleGame(player_color=True)
    sims, moves, t0 = 50, 0, time.perf_counter()
    while gam.get_result() is None and time.perf_counter() - t0 < seconds:
        bm, am = agent.best_move(gam, real_game=False, ai_move=True, max_iters=sims, noise=True)
        gam.move(bm)
        gam.move(am)
        moves += 1
    dt = time.perf_counter() - t0
    return {"value": moves * sims / dt, "unit": "simulations/s", "cores": 1, "kind": "port",
            "sample": "oracle/ (CPU restatement of selfplay.py->mctree.py + python-chess rules + fp32 "
                      "tower): 1 game, %d moves x %d sims, 2-block/32-filter net, %d tower calls, %.1f s; "
                      "reference's own figure: 0.4 s/iteration on an i5-7600K (DOCS.md:75)"
                      % (moves, sims, agent.n_evals, dt),
            "host_cores": os.cpu_count()}


def time_record_gather(run, dist, max_plies):
    """C4's one collective: every rank's game records to every rank (records.gather_blocks:
    RCCL all_gather over xGMI).  The records are the device's own record arrays of the G games
    resident on each rank -- what a finished wave of games hands over."""
    import numpy as np
    from chessrl_amd import records
    moves, plies, res = run.engine.ctx.records()
    t0 = time.perf_counter()
    block = records.pack_arrays(run.game_id, moves, plies, res, run.color, max_plies)
    pack_ms = (time.perf_counter() - t0) * 1e3
    dist.barrier()
    best = None
    for _ in range(3):
        st = {}
        rows, counts = records.gather_blocks(block, stats=st, force_collective=True)
        if best is None or st["ms"] < best["ms"]:
            best = st
        dist.barrier()
    assert rows.shape[0] == sum(counts) and len(np.unique(rows[:, 0].astype(np.int64) |
                                                           (rows[:, 1].astype(np.int64) << 31))) == rows.shape[0]
    return {"records": int(rows.shape[0]), "bytes_per_record": int(rows.shape[1] * 4),
            "bytes_gathered_per_rank": int(best["bytes_gathered"]), "ms": best["ms"],
            "GB_per_s_per_rank": best["bytes_gathered"] / best["ms"] / 1e6, "host_pack_ms": pack_ms,
            "backend": best["backend"],
            "what": "all_gather of counts + padded int32 record blocks incl. H2D/D2H staging, best of 3"}


def group_timeout_s(dist):
    """The timeout the process group really carries (read back from its backend options), in seconds."""
    if dist is None:
        return None
    try:
        pg = dist.distributed_c10d._get_default_group()
        dev = torch.device("cuda") if dist.get_backend() == "nccl" else torch.device("cpu")
        return pg._get_backend(dev).options._timeout.total_seconds()
    except Exception:                                        # private API: the line then says "unknown"
        return "unknown"


def agree_max(dist, x, rdev):
    """max over ranks of a small integer (identity without a process group)."""
    if dist is None:
        return int(x)
    t = torch.tensor([int(x)], dtype=torch.int64, device=rdev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return int(t.item())


def every_rank(dist, values, rdev):
    """[[values of rank 0], [values of rank 1], ...] on every rank (one all_gather of a few doubles)."""
    mine = torch.tensor([float(v) for v in values], dtype=torch.float64, device=rdev)
    if dist is None:
        return [mine.tolist()]
    world = dist.get_world_size()
    out = torch.zeros(world * len(values), dtype=torch.float64, device=rdev)
    dist.all_gather_into_tensor(out, mine)
    return out.cpu().view(world, len(values)).tolist()


def spread(xs):
    return {"min": min(xs), "mean": sum(xs) / len(xs), "max": max(xs), "ranks": list(xs)}


def dry_run(a, rank, world, dist):
    """CRL_BENCH_DRYRUN=1 (tests, no GPU needed): the launcher / process-group / record-gather
    control flow only.  Nothing is measured and the line says so."""
    import numpy as np
    from chessrl_amd import records
    rng = np.random.default_rng(rank)
    n = 5 + rank
    plies = rng.integers(0, 40, n)
    block = records.pack_arrays(rank + world * np.arange(n), rng.integers(0, 4000, (n, 64)), plies,
                                np.full(n, 2), np.zeros(n), 64)
    st = {}
    rows, counts = records.gather_blocks(block, stats=st)
    assert counts == [5 + r for r in range(world)]
    # the bench's own small collectives with synthetic per-rank numbers: the precision-mode agreement (any
    # rank on f16x3 -> every rank), the parity gate's decision, the per-rank step and trunk times
    rdev = torch.device("cpu")
    mode = agree_max(dist, 2 if rank == world - 1 else 0, rdev)
    per = every_rank(dist, [2.0 + 0.01 * rank, 1.0 + 0.001 * rank], rdev)
    if rank == 0:
        print(json.dumps({"metric": "MCTS simulations/sec at 800 sims/move", "value": None,
                          "dry_run": True, "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
                          "mode_agreed": ["f16", "hybrid", "f16x3"][mode],
                          "process_group_timeout_s": group_timeout_s(dist),
                          "per_rank": {"ms_per_step": spread([p[0] for p in per]),
                                       "trunk_launch_ms": spread([p[1] for p in per])},
                          "record_gather": {"records": int(rows.shape[0]), "backend": st.get("backend")}}),
              flush=True)


def play_opening(run, moves, sims):
    """``moves`` shortened moves of ``sims`` simulations each, Dirichlet noise on (un-timed).  Every game starts from the
    standard position and a search is deterministic, so without this all games of one colour hold the SAME tree for
    the whole first move -- and a first move cut to one simulation (rounds 1-5) has one root child, which the noisy
    policy must choose: the timed window then sat in move 2 of 4096 games in two distinct positions.  The tower's
    time does not care, but everything data-dependent does: the hybrid mode listed either no S1 board of a step or
    every board of a colour (C5, round 6 run 1: 0.1 ... 5.7 ms per indexed launch inside one window).  A few noisy moves
    put every game on its own line, as every later move of a real run is."""
    sims = max(1, min(sims, run.sims))
    for _ in range(moves):
        if run._sims_in_move is None:
            run.begin_move()
        run.engine.run_steps(sims - run._sims_in_move)
        run._sims_in_move = sims
        run.end_move()                                     # (a shortened move draws its noise at the boundary)


def timed_window(run, a, barrier, model=None, sync=None, stamped=0):
    """W un-timed warm-up steps, then EXACTLY K timed steps between barrier + synchronize pairs.  A window
    shorter than a move is centred on the middle of a move (trees pre-grown un-timed); a few shortened, noisy
    opening moves first (un-timed, ``play_opening``) put every game on a line of its own and load every
    move-boundary kernel and host path once, so that a boundary timed later is a steady-state one.
    ``stamped``: that many steps of the stamped build of the step are replayed right behind the window
    (``graph_phases_end``; its graphs are captured and warmed in front of the window).
    ``model``: its counter of S1 boards evaluated twice (hybrid) is read INSIDE the window, next to the simulation
    counters -- behind the un-timed steps (round 5 read it in front of them and divided ~21 windows' worth of
    fall-back boards by one window's simulations).  The runner's precision guard is switched off for the window: a
    mode change (and a graph re-capture) in mid-window would time two arithmetics under one name."""
    sync = sync or torch.cuda.synchronize
    guard_every, run.GUARD_EVERY = getattr(run, "GUARD_EVERY", 0), 0
    try:
        return _timed_window(run, a, barrier, model, sync, stamped)
    finally:
        run.GUARD_EVERY = guard_every


def _timed_window(run, a, barrier, model, sync, stamped):
    if run._sims_in_move:                         # a second window (another precision mode): finish the move
        run.end_move()
    # un-timed opening: the games leave their common line; it also loads every move-boundary kernel and host path
    play_opening(run, getattr(a, "opening_moves", 1) if run.moves_played == 0 else 1, getattr(a, "opening_sims", 1))
    pre = 0
    if a.steps < a.sims:
        # mid-move; when the window fits into the second half of the move it starts just behind the
        # point where the runner draws the move's Dirichlet noise ahead (sims // 2), so that no host
        # work of the runner falls into a window of a few tens of milliseconds
        second_half = a.sims // 2 + 8
        start = second_half if a.steps <= a.sims - second_half - 1 else (a.sims - a.steps) // 2
        pre = max(0, start - a.warmup)
    run.steps(pre + a.warmup)
    distinct = None
    get_positions = getattr(run.engine.ctx, "get_positions", None)
    if get_positions is not None:
        # how many different root positions the window's moves start from (evidence for the opening: rounds 1-5 timed 4096
        # games in TWO positions)
        import numpy as np
        distinct = int(len(np.unique(get_positions(), axis=0)))
    gp = graph_phases_begin(run, stamped) if stamped else None      # (may move on into the next move: room for its leg)
    run.engine.prepare_graphs(a.steps)        # (nothing is captured inside the timed region)
    fallback = getattr(model, "fallback_boards", None) if getattr(model, "fused", False) else None
    w = {"pre": pre, "window_start": run._sims_in_move or 0, "moves0": run.moves_played, "distinct_roots": distinct,
         "c0": run.engine.ctx.counters(), "fb0": fallback() if fallback else 0}
    barrier()
    t0 = time.perf_counter()
    run.steps(a.steps)                        # K lockstep steps (the engine replays several steps per hipGraph launch)
    sync()
    t1 = time.perf_counter()
    barrier()
    w["c1"] = run.engine.ctx.counters()
    w["fb1"] = fallback() if fallback else 0
    w["moves1"] = run.moves_played            # (the phase profile later crosses a move boundary of its own)
    w["dt"] = t1 - t0
    w["sims"] = w["c1"]["sims"] - w["c0"]["sims"]          # simulations completed (backed up) in the timed region
    # fraction of the window's simulations whose S1 board went through the f16x3 fall-back (this rank's games)
    w["twice"] = (w["fb1"] - w["fb0"]) / max(1, w["sims"])
    if gp is not None:
        # the stamped leg, right behind the window: same trees, same clocks, no capture in between
        try:
            w["graph_phases"] = graph_phases_end(run, *gp)
        finally:
            run.engine.drop_stamped_graphs()
    return w


def hybrid_entry(entry, model, win, run=None):
    """What a timed window in ``hybrid`` adds to its precision_modes entry.  With ``run``: the reply rule checked on the
    window's last tree leaves (every board whose f16 and f16x3 replies differ must be one the margin lists;
    ChessModel.reply_rule_check) -- un-timed, behind the window."""
    if getattr(model, "fused", False) and model.precision == "hybrid":
        entry["s1_boards_evaluated_twice"] = win["twice"]
        entry["reply_margin"] = model.reply_margin
        eng = getattr(run, "engine", None)
        if eng is not None and getattr(eng, "legal_priors", False) and hasattr(model, "reply_rule_check"):
            entry["reply_rule_on_the_last_leaves"] = model.reply_rule_check(eng.planes_s2, eng._lab_s2[0], eng._lab_s2[1])
    return entry


TOWER_BAR = 1e-3         # north_star: policy / value outputs within 1e-3 of the reference net on the same weights


def harvest_positions(model, n, seed, device):
    """``n`` positions of COMPLETE self-play games played with the timed weights in the timed arithmetic
    (128 games in lockstep, 16 simulations per move, Dirichlet noise on: openings, middle games, the long
    endgames random-init play drifts into, positions after promotions), drawn evenly over every game's
    length, as the tower's own input: plane bitboards int64 [n,128] written by the HIP encoder."""
    import numpy as np
    from chessrl_amd.engine import LockstepEngine
    from chessrl_amd.selfplay import SelfPlayRunner
    games = 128
    side = SelfPlayRunner(model, games, 16, seed=seed + 7919, noise=True, total_games=games, max_plies=1024,
                          device=device)
    recs = side.run()
    side.close()
    rng = np.random.default_rng(seed)
    moves = [np.asarray(r.moves, dtype=np.uint16) for r in recs if len(r.moves) >= 8]
    total = sum(len(m) for m in moves)
    prefixes = []
    for m in moves:
        k = max(1, int(round(n * len(m) / total)))
        for ply in np.unique(rng.integers(0, len(m) + 1, size=k)):
            prefixes.append(m[:int(ply)])
    while len(prefixes) < n:
        m = moves[int(rng.integers(len(moves)))]
        prefixes.append(m[:int(rng.integers(0, len(m) + 1))])
    prefixes = prefixes[:n]
    eng = LockstepEngine(model, n_games=n, max_sims=2, use_graph=False, max_plies=1024, device=device)
    eng.load_moves(prefixes)
    eng.ctx.encode(eng.planes_s1.data_ptr())
    eng.ctx.sync()
    bits = eng.planes_s1.clone()
    eng.close()
    plies = np.array([len(p) for p in prefixes])
    info = {"games": len(moves), "plies_mean": float(plies.mean()), "plies_max": int(plies.max()),
            "opening_lt_20": int((plies < 20).sum()), "late_ge_150": int((plies >= 150).sum()),
            "after_a_promotion": int(sum(bool(((p >> 12) & 7).any()) for p in prefixes))}
    return bits, info


def planes_of_bitboards(bits):
    """int64 [n,128] plane bitboards (bit sq of plane c = channel c on square sq; row 0 of the planes is
    rank 8: position p = sq ^ 56) -> fp32 NHWC [n,8,8,127], the input the reference net is given."""
    import numpy as np
    b = np.ascontiguousarray(bits).view(np.uint64)
    sq = (np.arange(64) ^ 56).astype(np.uint64)
    x = (b[:, None, :127] >> sq[None, :, None]) & np.uint64(1)
    return x.astype(np.float32).reshape(b.shape[0], 8, 8, 127)


def tower_error_vs_fp32(model, sets, precision):
    """The timed tower mode against the fp32 tower oracle (oracle/tower_oracle.py: the CPU restatement of
    model.py:31-63 the 1e-3 bar is defined on) on the SAME weights -- the checker of the headline, run
    after the timed region beside cpu_baseline.  ``sets``: [(name, plane bitboards int64 [n,128] on the
    device, info)]."""
    import numpy as np
    from oracle import tower_oracle
    t0 = time.perf_counter()
    out = {"mode": precision, "bar": TOWER_BAR, "positions": 0, "sets": []}
    dp_all, dv_all = [], []
    for name, bits, info in sets:
        pol, val = model._forward_fused(bits, precision=precision)
        torch.cuda.synchronize()
        pol, val = pol.cpu(), val.cpu()
        host = bits.cpu().numpy()
        dp, dv = [], []
        for i in range(0, host.shape[0], 512):
            epol, evalue = tower_oracle.forward(model.weights, planes_of_bitboards(host[i:i + 512]))
            dp.append((pol[i:i + 512] - epol).abs().max(dim=1).values.numpy())
            dv.append((val[i:i + 512] - evalue).abs().numpy())
        dp, dv = np.concatenate(dp), np.concatenate(dv)
        dp_all.append(dp)
        dv_all.append(dv)
        e = {"set": name, "positions": int(len(dv)), "dpolicy_max": float(dp.max()), "dvalue_max": float(dv.max())}
        e.update(info or {})
        out["sets"].append(e)
    dp, dv = np.concatenate(dp_all), np.concatenate(dv_all)
    out.update(positions=int(len(dv)), dpolicy_max=float(dp.max()), dvalue_max=float(dv.max()),
               dvalue_p999=float(np.quantile(dv, 0.999)), dvalue_mean=float(dv.mean()),
               positions_beyond_bar=int(((dp > TOWER_BAR) | (dv > TOWER_BAR)).sum()),
               within_bar=bool(max(dp.max(), dv.max()) <= TOWER_BAR),
               oracle="oracle/tower_oracle.forward (fp32 PyTorch-CPU, Keras semantics of model.py:31-63), "
                      "%d threads" % torch.get_num_threads(),
               seconds=time.perf_counter() - t0)
    return out


def issued_flops(F, B, G):
    """MFMA FLOPs the split-precision trunk ISSUES per launch: three products per residual-conv MAC (hi.Whi,
    lo.Whi, hi.Wlo), two for the stem (its 0/1 planes have no lo part); the 1x1 head convs run on the VALU."""
    return 2.0 * (2 * 73152 * F + 3 * 1152 * F * F * B) * G


def compliant_roofline(model, eng, ph, F, B, G, peak, shape, gp=None):
    """``roofline_compliant``: the kernel every evaluation under the 1e-3 bar runs when a net needs the compliant
    mode (hybrid / f16x3: S2's priors and value) -- the split-precision trunk -- timed inside eager steps of that
    mode on the same games and weights: algorithmic fraction (the three MFMAs of a product count once), issued
    fraction, PMC traffic of that kernel at this shape."""
    kern = trunk_kernel_name(F, G, int(eng.bitplanes) | 2)
    ins = (gp or {}).get("trunk", {}).get("f16x3")
    src = "stamp kernels around the launch inside the replayed hipGraph of the step"
    if ins is None:
        ins = ph.get("trunk_in_step", {}).get("f16x3")
        src = "HIP events around the launch inside eager steps mid-move"
    if ins is None:
        return None
    ms = ins["launch_ms"]
    alg = 2.0 * (73152 * F + 1152 * F * F * B + 192 * F) * G
    traffic, src = pmc_traffic(kern, shape)
    plane_in = 1024 if eng.bitplanes else 64 * 128 * 2
    return {"bound": "mfma", "kernel": "crl_tower::%s (split precision: hi/lo fp16 operands, three MFMAs per product)" % kern,
            "launch_ms": ms, "launch_ms_source": src,
            "flops_per_launch": alg, "achieved": alg / ms / 1e9, "peak": peak, "unit": "TFLOP/s",
            "frac": alg / ms / 1e9 / peak,
            "issued_flops_per_launch": issued_flops(F, B, G), "issued_frac": issued_flops(F, B, G) / ms / 1e9 / peak,
            "traffic": traffic, "traffic_source": src,
            "algorithmic_bytes": G * plane_in + 2 * (9 * 128 * F + 2 * B * 9 * F * F) * 2 + G * 192 * 4,
            "trunk_in_step": (gp or {}).get("trunk") or ph["trunk_in_step"],
            "phase_ms": (gp or {}).get("parts") or {k: ph[k] for k in ("select_expand", "tower_s1", "reply", "tower_s2")}}


def tracked_whole_run(config=None, mode=None, root=None):
    """Whole-game figures of a tracked rolling-rounds run (tools/rolling_probe.py; NOT measured by this bench run) of
    the SAME configuration -- (games in lockstep, sims per move, "BxF") -- in the SAME tower precision mode, without
    training in the loop: moves per game for the games/hour estimate and the measured games/hour.  The newest match
    wins (by round directory, then by file time).  None when the tree holds no such run: round 5's line carried the
    games/hour of a ``hybrid`` run of another net beside an ``f16`` rate."""
    import glob
    import re

    def age(path):
        m = re.search(r"r(\d+)", os.path.basename(os.path.dirname(path)))
        return (int(m.group(1)) if m else -1, os.path.getmtime(path))

    for path in sorted(glob.glob(os.path.join(root or ROOT, "profiles", "r*", "rolling_probe*.json")), key=age, reverse=True):
        try:
            d = json.load(open(path))
            tower = d["tower"].split()
            cfg = (d["games_in_lockstep"], d["sims_per_move"], tower[0])
            ran = d.get("tower_precision", tower[1] if len(tower) > 1 else None)
            if d.get("training_in_the_loop") or d.get("tower_precision_at_start", ran) != ran:
                continue                                     # another workload / a run that changed its arithmetic
            if (config is not None and tuple(config) != cfg) or (mode is not None and ran != mode):
                continue
            plies = [r["plies_mean"] for r in d["rounds"]]
            return {"source": os.path.relpath(path, root or ROOT) + " (tools/rolling_probe.py: start-up and final tail "
                              "included; NOT measured by this bench run)",
                    "config": cfg, "tower_precision": ran,
                    "moves_per_game": sum(plies) / len(plies) / 2.0,
                    "games_per_hour": d["games_per_hour_overall"], "seconds": d["seconds_total"],
                    "games": d["games_total"], "training_in_the_loop": False}
        except (OSError, ValueError, KeyError, ZeroDivisionError, IndexError):
            continue
    return None


def measure_games_per_hour(seconds, seed, device, steps_per_graph=None):
    """``metric`` names self-play games/hour: this leg MEASURES one, in the bench run itself -- complete games at C2's
    size (512 in lockstep, 100 sims/move, 6x64 random-init net in the mode ``auto`` picks, Dirichlet noise, finished
    slots refilled at once), the only BASELINE configuration whose games (~0.15 s per move round, ~2 s per game) turn
    over several times in a leg of seconds.  Every game is young at the start, so the first quarter of the leg only
    mixes the batch; the rate is taken over the rest."""
    from chessrl_amd.model import ChessModel
    from chessrl_amd.selfplay import SelfPlayRunner
    G, S = 512, 100
    model = ChessModel(blocks=6, filters=64, device="cuda:%d" % device, seed=seed, precision="auto")
    run = SelfPlayRunner(model, G, S, seed=seed + 104729, noise=True, device=device, max_plies=2048,
                         steps_per_graph=steps_per_graph)
    run.play_move()                                        # captures, first boundary
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    base = len(run.finished)
    marks = []
    while True:
        run.play_move()                                    # (ends on a synchronising boundary)
        t = time.perf_counter() - t0
        marks.append((t, len(run.finished) - base, run.sims_run))
        if t >= seconds:
            break
    i0 = next(i for i, m in enumerate(marks) if m[0] >= seconds / 4.0)
    if i0 == len(marks) - 1:
        i0 = 0
    games, secs = marks[-1][1] - marks[i0][1], marks[-1][0] - marks[i0][0]
    recs = run.finished[base + marks[i0][1]:]
    out = {"config": "C2: %d self-play games in lockstep, %d sims/move, 6x64 random-init tower (%s), Dirichlet noise, "
                     "finished slots refilled" % (G, S, model.precision),
           "games": int(games), "seconds": secs, "games_per_hour": games / secs * 3600.0 if secs > 0 else None,
           "moves_per_game": (float(sum(len(r.moves) for r in recs)) / len(recs) / 2.0) if recs else None,
           "simulations_per_s": (marks[-1][2] - marks[i0][2]) / secs if secs > 0 else None,
           "whole_leg": {"games": int(marks[-1][1]), "seconds": marks[-1][0], "mixing_seconds_not_counted": marks[i0][0]},
           "note": "measured in this run, complete games; a young batch finishes its short games first, hence the "
                   "un-counted first quarter"}
    run.close()
    return out


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(a.gpus))
    rank, world, local, dist = init_ranks(a)
    if os.environ.get("CRL_BENCH_DRYRUN") == "1":
        dry_run(a, rank, world, dist)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    # collectives of the bench itself travel on the process group's own device type
    rdev = dev if (dist is None or dist.get_backend() == "nccl") else torch.device("cpu")

    from chessrl_amd.model import ChessModel
    from chessrl_amd.selfplay import SelfPlayRunner
    tdt = {"f16": torch.float16, "bf16": torch.bfloat16, "f32": torch.float32}[a.dtype]
    model = ChessModel(blocks=a.blocks, filters=a.filters, device="cuda:%d" % local, dtype=tdt,
                       seed=a.seed, fused=not a.no_fused, precision=a.precision, weights=a.weights)
    a.blocks, a.filters = model.blocks, model.filters          # (a weight file brings its own architecture)
    # every rank times the same arithmetic: "auto" decides per rank (same weights, same probe -- but a
    # decision at the edge of the tolerance must not leave one rank in f16x3 beside seven in f16)
    order = ["f16", "hybrid", "f16x3"]
    if model.fused and dist is not None:
        strictest = order[agree_max(dist, order.index(model.precision), rdev)]
        if strictest != model.precision:
            model.set_precision(strictest)
    probe_at_load = dict(model.precision_probe) if getattr(model, "precision_probe", None) else None
    max_plies = 2048
    run = SelfPlayRunner(model, a.games, a.sims, seed=a.seed, noise=True, rank=rank, world=world,
                         device=local, use_graph=not a.no_graph, max_plies=max_plies,
                         numpy_promotion=a.numpy_promotion, steps_per_graph=a.steps_per_graph)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def reduce_window(w):
        """(simulations of all ranks, max wall time over ranks, per-rank [ms_per_step])"""
        per = every_rank(dist, [w["sims"], w["dt"]], rdev)
        return sum(p[0] for p in per), max(p[1] for p in per), [p[1] / a.steps * 1e3 for p in per]

    n_stamped = a.graph_phase_steps if (rank == 0 and model.fused) else 0
    win = timed_window(run, a, barrier, model, stamped=n_stamped)
    total_sims, max_dt, rank_ms = reduce_window(win)
    timed = {model.precision if model.fused else a.dtype: hybrid_entry({"simulations_per_s": total_sims / max_dt,
                                                                        "ms_per_step": max_dt / a.steps * 1e3}, model, win, run)}

    # ---- the headline carries its own parity evidence: the mode that was timed against the fp32 oracle on
    # the same weights, on positions of complete games played with those weights and on the positions the
    # timed window itself handed to the tower last (rank 0; the oracle is the checker, never the product)
    parity, parity_sets = None, None
    if model.fused and a.parity_positions > 0 and rank == 0:
        leaves = torch.cat([run.engine.planes_s1[:512], run.engine.planes_s2[:512]]).clone()
        bits, info = harvest_positions(model, a.parity_positions, a.seed, local)
        parity_sets = [("complete self-play games with the timed weights (128 games x 16 sims/move)", bits, info),
                       ("the timed window's last tower inputs (512 x S1 + 512 x S2 tree leaves)", leaves, None)]
        parity = tower_error_vs_fp32(model, parity_sets, model.precision)
    retime = agree_max(dist, parity is not None and not parity["within_bar"] and model.precision == "f16", rdev)
    if retime:
        # the timed mode misses the bar on these weights: the headline is the rate of the mode the product
        # falls back to (every output under the bar evaluated in f16x3)
        model.set_precision(model.AUTO_STRICT)
        win = timed_window(run, a, barrier, model, stamped=n_stamped)
        total_sims, max_dt, rank_ms = reduce_window(win)
        timed[model.precision] = hybrid_entry({"simulations_per_s": total_sims / max_dt,
                                               "ms_per_step": max_dt / a.steps * 1e3}, model, win, run)
        if rank == 0:
            failed = parity
            parity = tower_error_vs_fp32(model, parity_sets, model.precision)
            parity["first_timed_mode"] = {k: failed[k] for k in ("mode", "dpolicy_max", "dvalue_max", "dvalue_p999",
                                                                  "positions_beyond_bar", "within_bar")}
    c0, c1, dt = win["c0"], win["c1"], win["dt"]
    pre, window_start, moves0, moves1 = win["pre"], win["window_start"], win["moves0"], win["moves1"]
    gather = time_record_gather(run, dist, max_plies) if dist is not None else None
    # every rank's trunk launch time (clock / straggler visibility when the scaling curve is run)
    eng = run.engine
    k_ms = event_time_ms(lambda: model._run_fused(eng.planes_s2), 50) if model.fused else 0.0
    rank_k_ms = [p[0] for p in every_rank(dist, [k_ms], rdev)]

    if rank == 0:
        G, F, B = a.games, a.filters, a.blocks
        d = {k: c1[k] - c0[k] for k in c1}
        nodes = max(1, d["nodes"])
        depth = d["depth_sum"] / max(1, d["sims"])
        branch = d["branch_sum"] / nodes
        shape = "%d boards, %d blocks x %d filters" % (G, B, F)
        # ---- dominant kernel (MFMA-bound) ----------------------------------------------------
        tower_ms = event_time_ms(eng.phase_tower_s2, 20)     # trunk + both heads, as the step runs them
        tower_flops = 2.0 * model.macs_per_eval() * G
        peak = MFMA_PEAK_TFLOPS[a.dtype]
        ms_step = max_dt / a.steps * 1e3
        boundary = time_move_boundary(run)               # rank 0 only; the other ranks wait at the last barrier
        ph = profile_phases(run, 16)                     # eager steps mid-move: every phase and every trunk launch
        gp = win.get("graph_phases")                     # the same from the replayed hipGraph, right behind the timed window
        main_kind = model._trunk_mode() if model.fused else None     # the arithmetic of the roofline's kernel
        k_ms_b2b, in_step = k_ms, False
        if gp is not None and main_kind in gp["trunk"]:
            # the kernel's time INSIDE the replayed graph (stamp kernels around the launch): round 5's C5 line added
            # eager phase times up to 1.03-1.10 x its own graph-replayed step -- legs of one long process hold
            # different clocks; the stamps ride in the graph that is being timed
            in_step = "graph"
            k_ms = gp["trunk"][main_kind]["launch_ms"]
        elif model.fused and main_kind in ph.get("trunk_in_step", {}):
            in_step = "eager"
            # the kernel's time INSIDE the steps (HIP events around the launch, on the launch stream): 50 launches
            # back to back after the run hold a lower clock than a launch between search kernels and heads does --
            # round 4's line carried a kernel time that did not fit twice into its own step
            k_ms = ph["trunk_in_step"][main_kind]["launch_ms"]
        if model.fused:
            # the hand-written fused trunk: ONE launch per tower forward.  Algorithmic FLOPs per
            # launch = 2 x (stem 73152 F + blocks 1152 F^2 B + head convs 192 F) MACs per board
            #  x G boards.
            k_flops = 2.0 * (73152 * F + 1152 * F * F * B + 192 * F) * G
            kern = trunk_kernel_name(F, G, int(eng.bitplanes) | (2 if model._trunk_mode() == "f16x3" else 0))
            k_name = "crl_tower::%s (fused stem + %d residual blocks + head convs, %d boards)" % (kern, B, G)
            traffic, traffic_src = pmc_traffic(kern, shape)
            # what one launch must move: the encoder's planes as handed over (1 KiB of plane
            # bitboards per board, or 16 KiB of fp16 planes), the folded fp16 weights once, the
            # head activations (192 floats per board)
            plane_in = 1024 if eng.bitplanes else 64 * 128 * 2
            alg_bytes = G * plane_in + (9 * 128 * F + 2 * B * 9 * F * F) * 2 + G * 192 * 4
        else:
            # PyTorch-ROCm tower: the FxF 3x3 residual-block convolution.  bias=None: exactly ONE
            # kernel per call (MIOpen igemm_fwd_gtcx35_nhwc_*), comparable with rocprofv3 --stats
            conv = model.net.conv1[0]
            x = torch.randn((G, F, 8, 8), device=dev, dtype=tdt).contiguous(memory_format=torch.channels_last)
            k_ms = event_time_ms(lambda: torch.nn.functional.conv2d(x, conv.weight, None, padding=1), 50)
            k_flops = 2.0 * 9 * F * F * 64 * G
            k_name = "3x3 conv %d->%d, batch %d x 8x8 (PyTorch-ROCm / MIOpen igemm)" % (F, F, G)
            traffic, traffic_src, alg_bytes = None, "not profiled", None
        roof = {"bound": "mfma", "kernel": k_name,
                "achieved": k_flops / k_ms / 1e9, "peak": peak, "unit": "TFLOP/s",
                "frac": k_flops / k_ms / 1e9 / peak, "traffic": traffic, "traffic_source": traffic_src,
                "algorithmic_bytes": alg_bytes,
                "launch_ms": k_ms, "launch_ms_source": {"graph": "stamp kernels around the launch inside the replayed "
                                                                 "hipGraph of the step (graph_phases)",
                                                        "eager": "HIP events around the launch inside 16 eager steps "
                                                                 "mid-move (profile_phases)",
                                                        False: "HIP events around 50 back-to-back launches"}[in_step],
                "launch_ms_back_to_back": k_ms_b2b, "flops_per_launch": k_flops,
                "tower_forward_ms": tower_ms, "tower_tflops": tower_flops / tower_ms / 1e9,
                "tower_frac": tower_flops / tower_ms / 1e9 / peak}
        eager_fit = None
        if model.fused and "trunk_ms_per_step" in ph:
            parts = {"trunk_launches": ph["trunk_ms_per_step"], "heads": ph["heads_and_margin_ms"],
                     "select_expand": ph["select_expand"], "reply": ph["reply"]}
            total = sum(parts.values())
            eager_fit = dict(parts, sum_ms=total, ratio=total / ms_step, trunk_in_step=ph["trunk_in_step"],
                             events=ph.get("events"), note="eager phases (HIP events) against the timed step")
        if gp is not None:
            # self-consistency: the step's parts as stamped inside the replayed graph must fit the timed step.  They
            # add up to the stamped step by construction; the ratio is the stamped step (parts + one launch gap per
            # stamp) against the un-stamped K timed steps
            total = gp["ms_per_step"]
            roof["step_fit"] = dict(parts=gp["parts"], sum_ms=total, ms_per_step=ms_step, ratio=total / ms_step,
                                    fits=bool(total <= 1.02 * ms_step), trunk_in_step=gp["trunk"],
                                    stamps_per_step=gp["stamps_per_step"], steps=gp["steps"], source=gp["source"],
                                    eager=eager_fit,
                                    note="parts stamped inside the replayed hipGraph against the un-stamped timed step")
        elif eager_fit is not None:
            roof["step_fit"] = dict(eager_fit, ms_per_step=ms_step, fits=bool(eager_fit["sum_ms"] <= 1.02 * ms_step))
        if model.fused and main_kind == "f16x3":
            roof["issued_frac"] = issued_flops(F, B, G) / k_ms / 1e9 / peak
        # ---- hand-written HIP search kernels (HBM-bound): select+expand and reply -----------
        # algorithmic bytes per simulation, SURVEY.md section 8d with the measured d and b;
        # with the fused trunk the encoders hand over 1-KiB plane bitboards (expanded on chip); any
        # other evaluator gets 16-KiB fp16 planes.  With the HIP heads only the legal moves' priors
        # exist (2 x b x (2 B label + 4 B prior written + 4 B read)); else full policy vectors
        plane_bytes = 1024.0 if eng.bitplanes else 16384.0      # per evaluated position (S1 and S2)
        # select reads one 24-byte edge record per legal move per level (value sum f64, visits i32,
        # prior f32, move u16, child u16: csrc/state.hpp; SURVEY's 20 B of fields + the move/child ids
        # the descent needs); the fixed part is SURVEY section 8d's 2.4 KB, of which the policy term
        # (2 x b x 2 B of logits there) is replaced by what this build really exchanges with the heads
        policy_bytes = (2 * branch * (2 + 4 + 4) if eng.legal_priors          # label u16 + prior f32 written + read
                        else 2 * (2 * 1968 * 4 + branch * 4))                  # full fp32 vectors written + read, gathered
        tree_bytes = (2400.0 - 2 * branch * 2) + 24.0 * depth * branch + policy_bytes + 2 * plane_bytes
        tree_ms = ph["select_expand"] + ph["reply"]
        t_traffic, t_src = pmc_traffic("k_select_expand + k_reply",
                                       "%d games, %s planes, %s" % (G, "bit" if eng.bitplanes else "fp16",
                                                                    "legal priors" if eng.legal_priors else "full policies"))
        tree = {"bound": "hbm", "kernel": "k_select_expand + k_reply (one simulation x %d games)" % G,
                "achieved": tree_bytes * G / tree_ms / 1e6, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": tree_bytes * G / tree_ms / 1e6 / HBM_PEAK_GBS,
                "traffic": t_traffic, "traffic_source": t_src,
                "launch_ms": tree_ms, "phase_ms": ph, "bytes_per_sim": tree_bytes,
                "bytes_model": "2.4 KB fixed (SURVEY 8d) + 24 B x depth x branch (edge records) + "
                               "%s + 2 x %d B planes" % ("2 x branch x 10 B legal labels/priors" if eng.legal_priors
                                                         else "2 x (2 x 7872 B policy vectors + 4 B x branch)", int(plane_bytes)),
                "mean_depth": depth, "mean_branch": branch}
        inside = (moves1 - moves0) // max(1, G)
        if boundary is not None:
            # whole moves at this step rate with the measured boundary: what a run of many moves sustains
            incl = G * a.sims / ((a.sims * ms_step + boundary["ms"]) * 1e-3) if inside == 0 else total_sims / max_dt
            boundary["note"] = ("timed separately (the window holds no boundary); value_incl_boundaries = G x S / "
                                "(S x ms_per_step + boundary ms)" if inside == 0 else
                                "the window already holds %d boundaries: value_incl_boundaries = value" % inside)
        else:
            incl = None
        # ---- both precision modes on the same games and weights: the timed one(s) over the K-step window,
        # the other one over a short window mid-move -----------------------------------------------------
        modes = None
        # the timed mode already is the compliant one -> its own in-step profile; else filled in by the hybrid leg
        compliant = (compliant_roofline(model, eng, ph, F, B, G, peak, shape, gp)
                     if model.fused and main_kind == "f16x3" else None)
        if model.fused:
            modes = {k: dict(v, window="the K timed steps") for k, v in timed.items()}
            modes[model.precision].update(trunk_kernel=kern, trunk_launch_ms=k_ms, trunk_launch_ms_back_to_back=k_ms_b2b)
            probe_dist = model.probe_error()
            keep = model.precision
            for other in ("f16", "hybrid", "f16x3"):
                if other in modes or a.strict_steps <= 0 or world != 1:
                    continue
                model.set_precision(other)
                run.begin_move()                                # fresh trees (the profiled move is abandoned)
                grow = min(max(8, run.sims // 4), run.sims // 2)
                n3 = max(1, min(a.strict_steps, run.sims - grow - 1))
                eng.run_steps(grow)                             # (re-captures the graphs) to mid-move
                eng.prepare_graphs(n3)
                torch.cuda.synchronize()
                cs0, fb0 = eng.ctx.counters()["sims"], model.fallback_boards()
                ts = time.perf_counter()
                eng.run_steps(n3)
                torch.cuda.synchronize()
                dts = time.perf_counter() - ts
                run._sims_in_move = grow + n3
                nsim = eng.ctx.counters()["sims"] - cs0
                k3 = trunk_kernel_name(F, G, int(eng.bitplanes) | (2 if model._trunk_mode() == "f16x3" else 0))
                modes[other] = {"simulations_per_s": nsim / dts,
                                "ms_per_step": dts / n3 * 1e3, "window": "%d steps mid-move" % n3,
                                "trunk_kernel": k3,
                                "trunk_launch_ms": event_time_ms(lambda: model._run_fused(eng.planes_s2), 10),
                                "note": {"f16x3": "same games, same weights, three MFMAs per product everywhere",
                                         "hybrid": "same games, same weights: S2 (priors, value) in f16x3, the reply "
                                                   "choice of S1 in f16 with an f16x3 fall-back for close calls -- the "
                                                   "mode precision='auto' picks when f16 is not within its tolerance",
                                         "f16": "same games, same weights, one fp16 MFMA per product: NOT the headline -- "
                                                "on these weights it is outside the 1e-3 bar or the probe's tolerance"}[other]}
                if other == "hybrid":
                    modes[other]["s1_boards_evaluated_twice"] = (model.fallback_boards() - fb0) / max(1, nsim)
                    modes[other]["reply_margin"] = model.reply_margin
                    compliant = compliant_roofline(model, eng, profile_phases(run, 8), F, B, G, peak, shape,
                                                   graph_phases(run, min(32, a.graph_phase_steps)))
            if model.precision != keep:
                model.set_precision(keep)
            modes["f16_vs_f16x3_on_probe"] = probe_dist
        cfg_name = {(512, 100, 6, 64): "C2", (4096, 800, 10, 128): "C3 (= C4 per-GPU shard)",
                    (4096, 800, 20, 256): "C5 per-GPU shard"}.get((G, a.sims, B, F), "custom")
        timed_mode = getattr(model, "precision", a.dtype)
        whole = tracked_whole_run((G, a.sims, "%dx%d" % (B, F)), timed_mode) if a.weights is None else None
        same_cfg = whole is not None
        rate = incl or total_sims / max_dt
        gph = None
        if world == 1 and a.gph_seconds > 0 and model.fused:
            gph = measure_games_per_hour(a.gph_seconds, a.seed, local, a.steps_per_graph)
        out = {
            "metric": "MCTS simulations/sec at 800 sims/move", "value": total_sims / max_dt,
            "unit": "simulations/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": max_dt / a.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
            "config": {"workload": "%s: %d self-play games in lockstep per GPU, %d sims/move, "
                                   "%d-block/%d-filter %s tower, standard start position, "
                                   "Dirichlet noise on" % (cfg_name if a.weights is None else "custom (trained weights)", G, a.sims, B, F,
                                                           "random-init" if a.weights is None else
                                                           "TRAINED (%s)" % os.path.basename(a.weights)),
                       "games_per_gpu": G, "sims_per_move": a.sims, "tower": "%dx%d" % (B, F),
                       "hipgraph": not a.no_graph, "steps_per_hipgraph_launch": eng.STEPS_PER_GRAPH if not a.no_graph else None,
                       "fused_trunk_kernel": bool(model.fused),
                       "policy_format": "legal priors [G,256]" if eng.legal_priors else "full [G,1968]",
                       "tower_precision": getattr(model, "precision", a.dtype),
                       "tower_precision_requested": a.precision, "tower_precision_probe": probe_at_load,
                       "tower_precision_why": ("the mode first timed missed the 1e-3 bar against the fp32 oracle on these "
                                               "weights (tower_error_vs_fp32.first_timed_mode): timed again in " + model.precision if retime
                                               else ("asked for by --precision" if a.precision != "auto" else
                                                     "ChessModel(precision='auto'): f16 kept only if within %g of f16x3 on "
                                                     "%d probe positions, else %s" % (model.PROBE_TOL, model.PROBE_POSITIONS,
                                                                                     model.AUTO_STRICT))),
                       "tower_precision_guard": getattr(model, "guard", None),
                       "trunk_kernel": kern if model.fused else None,
                       "numpy_promotion": eng.numpy_promotion,
                       "parallelism": "games sharded, no collective on the hot path"},
            "window": {"untimed_steps_before": pre + a.warmup, "first_sim_of_move": window_start,
                       "opening": "%d shortened moves of %d simulations, Dirichlet noise on, un-timed (play_opening)"
                                  % (a.opening_moves, a.opening_sims),
                       "distinct_root_positions": win.get("distinct_roots"), "games": G,
                       "move_boundaries_inside": inside,
                       "note": "a window shorter than one move is centred mid-move"},
            "per_rank": {"ms_per_step": spread(rank_ms), "trunk_launch_ms": spread(rank_k_ms)},
            "move_boundary": boundary, "value_incl_boundaries": incl,
            "moves_per_sec": rate / a.sims,
            # games/hour: steady state with refill = moves/s / moves per game, the latter read from the
            # tracked whole-run measurement of the SAME configuration (else no estimate is made)
            # games/hour.  MEASURED in this run: complete games at C2's size (measure_games_per_hour).  For the timed
            # configuration two figures READ from a tracked whole run of the same configuration in the same tower
            # precision mode (else none is stated): the steady-state estimate = this run's moves/s / that run's moves per
            # game, and that run's own games/hour
            "self_play_games_per_hour_measured": gph,
            "self_play_games_per_hour_est": (rate / a.sims / whole["moves_per_game"] * 3600.0 if same_cfg else None),
            "self_play_games_per_hour_whole_run": ({k: whole[k] for k in ("games_per_hour", "seconds", "games", "tower_precision",
                                                                         "training_in_the_loop", "source")}
                                                   if same_cfg else None),
            "tower_evals_per_sim": d["evals"] / max(1, d["sims"]),
            "gflop_per_sim": 2 * 2 * model.macs_per_eval() / 1e9,
            "roofline": roof, "roofline_tree": tree, "roofline_compliant": compliant, "precision_modes": modes,
            "tower_error_vs_fp32": parity,
        }
        if gather is not None:
            out["record_gather"] = gather
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(a.cpu_seconds)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()            # rank 0 is still profiling its phases: nobody tears RCCL down early
    run.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
