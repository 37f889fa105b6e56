"""``GameAgent`` -- a game against the neural agent.

Same surface as /root/reference/src/chessrl/gameagent.py:7-50: a ``Game`` that answers every
legal move of the human side with the agent's greedy move (``best_move(real_game=True)``:
arg-max of the policy over the legal moves, agentdistributed.py:57-58); if the agent has white
the first call of ``move`` plays the agent's opening move instead of the argument
(gameagent.py:35-38).  This is the batch-1 interactive path: one encoder launch and one tower
forward per move, no search.
"""
from .agent import Agent
from .game import Game


class GameAgent(Game):

    def __init__(self, agent, player_color=Game.WHITE, board=None, date=None):
        super().__init__(board=board, player_color=player_color, date=date)
        if isinstance(agent, Agent):
            self.agent = agent
        elif type(agent) == str:
            self.agent = Agent(not player_color, weights=agent)
        else:
            raise ValueError("An agent or path to the agents weights (.npz) is needed")

    def move(self, movement):
        """Makes a move; the agent answers.  Illegal moves are ignored (returns False)."""
        made_movement = False
        if self.agent.color and len(self) == 0:              # agent has white: it opens
            super().move(self.agent.best_move(self, real_game=True))
            made_movement = True
        else:
            made_movement = super().move(movement)
            if made_movement and self.get_result() is None:
                super().move(self.agent.best_move(self, real_game=True))
        return made_movement

    def get_copy(self):
        return GameAgent(board=self, agent=self.agent, player_color=self.player_color)

    def tearup(self):
        """Free resources."""
        del self.agent
        self.free()
