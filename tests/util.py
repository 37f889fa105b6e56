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
