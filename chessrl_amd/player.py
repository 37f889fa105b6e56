"""``Player`` -- the reference's abstract player (/root/reference/src/chessrl/player.py:1-14)."""


class Player(object):
    def __init__(self, color):
        if type(self) is Player:
            raise Exception("Cannot create Player Abstract class.")
        self.color = color

    def best_move(self, game):
        raise Exception("Abstract class.")
