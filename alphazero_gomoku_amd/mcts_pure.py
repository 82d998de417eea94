"""The rollout ("pure") MCTS opponent with the reference's surface (mcts_pure.py:249-273),
searching on the GPU.  The reference imports it as `from mcts_pure import MCTSPlayer as MCTS_Pure`
(train.py:249) and plays it against the network-guided player to monitor training."""
import logging

import numpy as np

from . import _lib
from .engine import SelfPlayEngine

_logger = logging.getLogger(__name__)


def numpy_stream_to_words(state=None):
    """np.random.get_state() (the legacy global stream the reference's rollouts draw from) as
    the uint32 [625] block azg_pure_search takes: 624 key words, then the position."""
    st = np.random.get_state() if state is None else state
    if st[0] != "MT19937":
        raise ValueError("the rollout player needs numpy's legacy MT19937 stream")
    words = np.empty(625, dtype=np.uint32)
    words[:624] = st[1]
    words[624] = st[2]
    return words


def words_to_numpy_stream(words, template=None):
    """Inverse of numpy_stream_to_words: put the stream back (np.random.set_state).  The cached
    Gaussian of the template state is kept: the rollouts draw uniforms only."""
    st = np.random.get_state() if template is None else template
    np.random.set_state((st[0], np.asarray(words[:624], dtype=np.uint32), int(words[624]), st[3], st[4]))


class MCTSPlayer(object):
    """AI player based on MCTS with random rollouts instead of a network."""

    def __init__(self, c_puct=5, n_playout=2000, device=0):
        self._c_puct, self._n_playout, self._device = float(c_puct), int(n_playout), device
        self._engine, self._engine_key = None, None
        self.player = None

    def _get_engine(self, board):
        key = (board.width, board.height, board.n_in_row)
        if self._engine is None or self._engine_key != key:
            if board.width != board.height:
                raise ValueError("only square boards are supported by the HIP engine")
            # the arena holds one expanded node per playout; no network is attached
            self._engine = SelfPlayEngine(board.width, board.n_in_row, self._n_playout, 1, self._c_puct,
                                          device=self._device, prior_mode=_lib.PRIOR_UNIFORM32,
                                          arena_nodes=self._n_playout + 8)
            self._engine_key = key
            self._engine.reset_games()
        return self._engine

    def set_player_ind(self, p):
        self.player = p

    def reset_player(self):
        """mcts_pure.py:258-259: mcts.update_with_move(-1); every search starts from a fresh root anyway."""
        if self._engine is not None:
            self._engine.update_with_move([-1])

    def get_action(self, board):
        """mcts_pure.py:261-270: n_playout rollout playouts from a fresh tree, most visited move.
        Draws from (and advances) numpy's global legacy stream exactly like the reference."""
        if len(board.availables) == 0:
            _logger.info("WARNING: the board is full")
            return None
        eng = self._get_engine(board)
        eng.set_position(0, board.states, board.last_move, board.current_player)
        template = np.random.get_state()
        words = numpy_stream_to_words(template).reshape(1, 625)
        move = int(eng.pure_search(self._n_playout, words)[0])
        words_to_numpy_stream(words[0], template)
        if move < 0:
            raise ValueError("the rollout player found no candidate move (Board.get_move_target is empty)")
        return move

    def __str__(self):
        return "MCTS_Pure {}".format(self.player)
