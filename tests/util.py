"""Shared helpers for the parity tests (oracle <-> C-ABI conversions)."""
from oracle.chess_oracle import OcBoard, OracleGame, board_from_fen, board_to_array  # noqa: F401

PERFT_FENS = {
    "kiwipete": "r3k2r/p1ppqpb1/bn2pnp1/3PN3/1p2P3/2N2Q1p/PPPBBPPP/R3K2R w KQkq - 0 1",
    "pos3": "8/2p5/3p4/KP5r/1R3p1k/8/4P1P1/8 w - - 0 1",
    "pos4": "r3k2r/Pppp1ppp/1b3nbN/nP6/BBP1P3/q4N2/Pp1P2PP/R2Q1RK1 w kq - 0 1",
    "pos4m": "r2q1rk1/pP1p2pp/Q4n2/bbp1p3/Np6/1B3NBn/pPPP1PPP/R3K2R b KQ - 0 1",
    "pos5": "rnbq1k1r/pp1Pbppp/2p5/8/2B5/8/PPP1NnPP/RNBQK2R w KQ - 1 8",
    "pos6": "r4rk1/1pp1qppp/p1np1n2/2b1p1B1/2B1P1b1/P1NP1N2/1PP1QPPP/R4RK1 w - - 0 10",
    "ep_pin": "8/8/8/KPp4r/8/8/8/4k3 w - c6 0 2",
    "ep_check": "8/8/8/2k5/3Pp3/8/8/4K3 b - d3 0 1",
    "promo": "n1n5/PPPk4/8/8/8/8/4Kppp/5N1N b - - 0 1",
    "dblcheck": "4k3/8/8/8/8/5n2/4r3/4K3 w - - 0 1",
}
# the known position with the most legal moves of any reachable chess position (218): the capacity
# of every per-node move / edge array (CRL_MAX_MOVES)
MAX_MOVES_FEN = "R6R/3Q4/1Q4Q1/4Q3/2Q4Q/Q4Q2/pp1Q4/kBNN1KB1 w - - 0 1"


def array_to_board(a):
    b = OcBoard()
    for i in range(6):
        b.bb[i] = int(a[i])
    b.white = int(a[6])
    b.state = int(a[7]) & 0xFFFFFFFF
    return b


def oracle_row(game):
    """np.uint64[8] of the oracle game's current board (state incl. derived ep bit)."""
    return board_to_array(game.board_at(0))


def oracle_games_parallel(net_kw, seed, sims, gid_colors, workers=None):
    """{game id: history} of complete oracle self-play games (noise on, per-game streams as in
    SelfPlayRunner), played on the host cores in parallel by child interpreters
    (``python -m tests.oracle_worker``: CPU only, started as ordinary child processes)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    workers = workers or max(1, min(16, (os.cpu_count() or 2) - 1, len(gid_colors)))
    procs = []
    for w in range(workers):
        jobs = [dict(net=net_kw, seed=seed, sims=sims, gid=int(g), color=bool(c))
                for g, c in gid_colors[w::workers]]
        p = subprocess.Popen([sys.executable, "-m", "tests.oracle_worker"], cwd=root, stdin=subprocess.PIPE,
                             stdout=subprocess.PIPE, text=True, env=dict(os.environ, OMP_NUM_THREADS="1"))
        p.stdin.write(json.dumps(jobs))
        p.stdin.close()
        procs.append(p)
    out = {}
    for p in procs:
        data = p.stdout.read()
        if p.wait() != 0:
            raise RuntimeError("oracle worker failed")
        out.update({int(k): v for k, v in json.loads(data).items()})
    return out


def selfplay_position_prefixes(n, seed=5, games=96, sims=12):
    """``n`` positions harvested from REAL self-play, as move-id prefixes: complete games played by
    the lockstep runner (6x64 random-init tower, Dirichlet noise), then (game, ply) pairs drawn
    evenly over each game's length -- openings, middle games, the long endgames random-init play
    drifts into, positions after promotions, captures and castling.  Returns (prefixes, info)."""
    import numpy as np
    from chessrl_amd.model import ChessModel
    from chessrl_amd.selfplay import SelfPlayRunner
    run = SelfPlayRunner(ChessModel(blocks=6, filters=64, seed=seed), games, sims, seed=seed, noise=True,
                         total_games=games, max_plies=1024)
    recs = run.run()
    run.close()
    rng = np.random.default_rng(seed)
    moves = [np.asarray(r.moves, dtype=np.uint16) for r in recs if len(r.moves) >= 8]
    total = sum(len(m) for m in moves)
    prefixes = []
    for m in moves:                                         # evenly spaced plies of every game
        k = max(1, int(round(n * len(m) / total)))
        for ply in np.unique(rng.integers(0, len(m) + 1, size=k)):
            prefixes.append(m[:int(ply)])
    while len(prefixes) < n:                                # top up (rounding, duplicates)
        m = moves[int(rng.integers(len(moves)))]
        prefixes.append(m[:int(rng.integers(0, len(m) + 1))])
    prefixes = prefixes[:n]
    plies = np.array([len(p) for p in prefixes])
    info = {"games": len(moves), "plies_min": int(plies.min()), "plies_max": int(plies.max()),
            "plies_mean": float(plies.mean()),
            "after_a_promotion": int(sum(bool(((p >> 12) & 7).any()) for p in prefixes)),
            "opening_lt_20": int((plies < 20).sum()), "late_ge_150": int((plies >= 150).sum())}
    return prefixes, info


def encode_prefixes(model, prefixes):
    """(model input as the search hands it over -- plane bitboards for the fused trunk --, the same
    positions as fp32 NHWC planes [n,8,8,127] for the fp32 oracle), both from the HIP encoder."""
    import numpy as np
    from chessrl_amd.engine import LockstepEngine
    n = len(prefixes)
    a = LockstepEngine(model, n_games=n, max_sims=2, use_graph=False, max_plies=1024)
    a.load_moves(prefixes)
    a.ctx.encode(a.planes_s1.data_ptr())
    x = a.planes_s1.clone()
    a.close()
    b = LockstepEngine(model, n_games=n, max_sims=2, use_graph=False, max_plies=1024, bitplanes=False,
                       legal_priors=False)
    b.load_moves(prefixes)
    b.ctx.encode(b.planes_s1.data_ptr())
    planes = b.planes_s1[..., :127].float().cpu().numpy()
    b.close()
    return x, planes
