"""MI355X-native AlphaZero-Gomoku self-play engine (HIP kernels behind a C ABI).

The module names mirror the reference's: Board / Game (game.py), MCTSPlayer
(mcts_alphaZero.py), MCTS_Pure (mcts_pure.py's MCTSPlayer, under the name train.py imports it
by), PolicyValueNet (policy_value_net_pytorch.py).  Importing the
package needs the built library alphazero_gomoku_amd/libazgomoku.so; there is no
CPU fallback."""
from . import _lib
from .board import Board
from .engine import PRIOR_HASH, PRIOR_NET, PRIOR_PEAKED, PRIOR_UNIFORM32, SelfPlayEngine
from .game import Game
from .mcts import MCTSPlayer
from .mcts_pure import MCTSPlayer as MCTS_Pure

__all__ = ["Board", "Game", "MCTSPlayer", "MCTS_Pure", "PolicyValueNet", "SelfPlayEngine", "PRIOR_NET",
           "PRIOR_UNIFORM32", "PRIOR_HASH", "PRIOR_PEAKED"]


def __getattr__(name):
    # PolicyValueNet pulls in torch; import it lazily
    if name == "PolicyValueNet":
        from .net import PolicyValueNet
        return PolicyValueNet
    raise AttributeError(name)
