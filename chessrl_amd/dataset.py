"""``DatasetGame`` -- host mirror of the reference's game-record container.

Same surface as /root/reference/src/chessrl/dataset.py:6-97:
a list of ``Game`` objects, JSON (de)serialisation of ``Game.get_history()`` dicts
(``{moves, result, player_color, date}``), ``augment_game`` (one training sample per move).
Games are the HIP-backed ``chessrl_amd.game.Game``; ``loads`` replays the recorded moves
through the rules kernels exactly like the reference replays them through python-chess.
``records.GameRecord`` is the slot-free form the lockstep runner produces;
``from_records`` converts.
"""
import json

from . import game


class DatasetGame(object):
    def __init__(self, games=None):
        self.games = [] if games is None else games

    def augment_game(self, game_base):
        """dataset.py:21-43: for the N moves of a game, N (state, next_move, result) samples."""
        hist = game_base.get_history()
        augmented = []
        g = game.Game(date=hist["date"], player_color=hist["player_color"])
        for m in hist["moves"]:
            augmented.append({"game": g, "next_move": m, "result": hist["result"]})
            g = g.get_copy()
            g.move(m)
        return augmented

    def load(self, path, slot_free=False):
        with open(path, "r") as f:
            self.loads(f.read(), slot_free=slot_free)

    def loads(self, string, slot_free=False):
        """dataset.py:50-57.  ``slot_free=True`` keeps the games as ``GameRecord``s (no device
        slot per game: training sets larger than the 4096-slot ``Game`` arena); the moves are then
        replayed -- and checked -- on the device when a batch is built."""
        if slot_free:
            from . import records
            self.games.extend(r for r in records.loads(string) if len(r) > 0)
            return
        for item in json.loads(string):
            g = game.Game(date=item["date"], player_color=item["player_color"])
            if len(item["moves"]) > 0:
                for m in item["moves"]:
                    g.move(m)
                self.games.append(g)

    def from_records(self, records):
        """Finished ``GameRecord``s of the lockstep runner -> ``Game`` objects."""
        for r in records:
            g = game.Game(date=r.date, player_color=r.player_color)
            for m in r.get_history()["moves"]:
                g.move(m)
            self.games.append(g)
        return self

    def save(self, path):
        existing = DatasetGame()
        try:
            existing.load(path)
        except FileNotFoundError:
            pass
        with open(path, "w") as f:
            json.dump([x.get_history() for x in existing.games + self.games], f)

    def append(self, other):
        if isinstance(other, game.Game):
            self.games.append(other)
        elif isinstance(other, DatasetGame):
            self.games.extend(other.games)

    def __str__(self):
        return json.dumps([x.get_history() for x in self.games])

    def __add__(self, other):
        self.append(other)
        return self

    def __len__(self):
        return len(self.games)

    def __getitem__(self, key):
        return self.games[key]
