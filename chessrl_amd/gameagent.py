"""``GameAgent`` -- interactive single game against the network.

Drop-in for the reference's class of the same name (/root/reference/src/chessrl/gameagent.py:7-50):
a ``Game`` in which every accepted move of the human side is answered at once by the agent's
greedy move -- ``best_move(real_game=True)``, the arg-max of the policy over the legal moves
(agentdistributed.py:57-58).  When the agent holds white, the very first ``move`` call plays
the agent's opening move and ignores its argument (gameagent.py:35-38).  Batch-1 path: one
encoder launch and one tower forward per answer, no tree search.
"""
from .agent import Agent
from .game import Game


def _resolve_agent(agent, human_color):
    """An ``Agent`` as given, or one built from a weights path playing the other colour."""
    if isinstance(agent, Agent):
        return agent
    if isinstance(agent, str):
        return Agent(not human_color, weights=agent)
    raise ValueError("GameAgent needs an Agent instance or the path of a weights file (.npz / .h5)")


class GameAgent(Game):

    def __init__(self, agent, player_color=Game.WHITE, board=None, date=None):
        super().__init__(board=board, player_color=player_color, date=date)
        self.agent = _resolve_agent(agent, player_color)

    def _agent_replies(self):
        return Game.move(self, self.agent.best_move(self, real_game=True))

    def move(self, movement):
        """Play ``movement`` for the human side and let the agent answer; returns whether a move
        was made.  An illegal move changes nothing and returns False (game.py:38-41)."""
        if self.agent.color and len(self) == 0:
            self._agent_replies()                 # white agent, empty board: it opens instead
            return True
        if not Game.move(self, movement):
            return False
        if self.get_result() is None:
            self._agent_replies()
        return True

    def get_copy(self):
        return GameAgent(self.agent, player_color=self.player_color, board=self)

    def tearup(self):
        """Drop the agent and give the device slot back."""
        self.agent = None
        self.free()
