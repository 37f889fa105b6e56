"""``DatasetGame`` -- the reference's container of recorded games.

Drop-in for /root/reference/src/chessrl/dataset.py:6-97: holds games, reads and writes the JSON
list of ``Game.get_history()`` dicts (``{moves, result, player_color, date}``) and expands a game
into one training sample per move (``augment_game``).  Two kinds of entries are accepted
wherever a game is expected, because both answer ``get_history()``: the HIP-backed
``chessrl_amd.game.Game`` (one device slot each) and the slot-free ``records.GameRecord`` the
lockstep runner emits.  Moves read from JSON are replayed through the rules kernels, i.e.
checked, like the reference replays them through python-chess.
"""
import json

from . import game as game_mod


def _replayed(moves, date, player_color):
    g = game_mod.Game(date=date, player_color=player_color)
    for uci in moves:
        g.move(uci)
    return g


def _histories(entries):
    return [e.get_history() for e in entries]


class DatasetGame(object):

    def __init__(self, games=None):
        self.games = games if games is not None else []

    # ---- samples -------------------------------------------------------------------------------
    def augment_game(self, game_base):
        """One ``{game, next_move, result}`` sample per recorded move: the position before the
        move, the move played there and the game's final result (dataset.py:21-43)."""
        h = game_base.get_history()
        samples = []
        position = game_mod.Game(date=h["date"], player_color=h["player_color"])
        for uci in h["moves"]:
            samples.append(dict(game=position, next_move=uci, result=h["result"]))
            position = position.get_copy()
            position.move(uci)
        return samples

    # ---- JSON ------------------------------------------------------------------------------------
    def loads(self, string, slot_free=False):
        """Append the games of a JSON text; games without moves are dropped (dataset.py:50-57).
        ``slot_free=True`` keeps them as ``GameRecord``s -- no device slot per game, for sets
        larger than the ``Game`` arena; their moves are checked when a training batch is built."""
        if slot_free:
            from . import records
            self.games += [r for r in records.loads(string) if len(r)]
            return
        for item in json.loads(string):
            if item["moves"]:
                self.games.append(_replayed(item["moves"], item["date"], item["player_color"]))

    def load(self, path, slot_free=False):
        with open(path) as f:
            self.loads(f.read(), slot_free=slot_free)

    def save(self, path):
        """Write the file's previous games followed by this dataset's (dataset.py:59-70)."""
        before = DatasetGame()
        try:
            before.load(path)
        except FileNotFoundError:
            pass
        with open(path, "w") as f:
            json.dump(_histories(before.games) + _histories(self.games), f)

    def __str__(self):
        return json.dumps(_histories(self.games))

    # ---- container ----------------------------------------------------------------------------------
    def from_records(self, recs):
        """``GameRecord``s -> ``Game`` objects (each takes a device slot)."""
        self.games += [_replayed(r.get_history()["moves"], r.date, r.player_color) for r in recs]
        return self

    def append(self, other):
        """A single game or every game of another dataset."""
        if isinstance(other, DatasetGame):
            self.games.extend(other.games)
        elif isinstance(other, game_mod.Game) or callable(getattr(other, "get_history", None)):
            self.games.append(other)                       # a Game, or a slot-free GameRecord
        else:
            raise TypeError("DatasetGame.append: expected a DatasetGame, a Game or a GameRecord, got %s"
                            % type(other).__name__)

    def __add__(self, other):
        self.append(other)
        return self

    __iadd__ = __add__

    def __len__(self):
        return len(self.games)

    def __getitem__(self, key):
        return self.games[key]
