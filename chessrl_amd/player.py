"""``Player`` -- base class of everything that can be asked for a move.

Mirrors the role of the reference's abstract player
(/root/reference/src/chessrl/player.py:1-14): it only records the side the player has
(``color``: True = white) and refuses to be used directly; ``Agent`` is the one concrete
player of this package (the Stockfish player of the reference is out of scope).
"""


class Player(object):
    def __init__(self, color):
        if type(self) is Player:
            raise TypeError("Player is abstract: instantiate Agent (or another subclass)")
        self.color = color

    def best_move(self, game):
        """UCI string of the move this player makes in ``game`` (subclasses implement it)."""
        raise NotImplementedError("%s does not implement best_move" % type(self).__name__)
