"""MCTSPlayer with the reference's surface (mcts_alphaZero.py:198-265), searching on the GPU."""
import logging
import random

import numpy as np

from . import _lib
from .engine import SelfPlayEngine

_logger = logging.getLogger(__name__)


def softmax(x):
    """mcts_alphaZero.py:16-19"""
    probs = np.exp(x - np.max(x))
    probs /= np.sum(probs)
    return probs


class MCTSPlayer(object):
    """AI player based on MCTS.

    `policy_value_function` is what the reference passes (train.py:102-105, human_play.py).
    When it is the bound method PolicyValueNet.policy_value_fn of this package (or the net
    itself), the whole search of a move runs in one GPU launch with the net's weights; the
    strings 'uniform32' / 'hash' select the exact synthetic providers of the parity tests.  Any
    OTHER callable -- the reference's contract policy_value_fn(board) -> (iterable of (action,
    prob), value), mcts_alphaZero.py:98-103 -- is called once per playout: the GPU selects down
    to a leaf, the host evaluates it, the GPU expands and backs up (priors are rounded to float32
    and widened once, the convention of the whole engine)."""

    def __init__(self, policy_value_function, c_puct=5, n_playout=2000, is_selfplay=0, device=0):
        self._c_puct, self._n_playout = float(c_puct), int(n_playout)
        self._is_selfplay = is_selfplay
        self._device = device
        self._net = None
        self._callback = None
        self._mode = _lib.PRIOR_NET
        owner = getattr(policy_value_function, "__self__", policy_value_function)
        if isinstance(policy_value_function, str):
            self._mode = {"uniform32": _lib.PRIOR_UNIFORM32, "hash": _lib.PRIOR_HASH}[policy_value_function]
        elif hasattr(owner, "sync_engine") and hasattr(owner, "param_tensors"):
            self._net = owner
        elif callable(policy_value_function):
            self._callback = policy_value_function
            self._mode = _lib.PRIOR_UNIFORM32                 # placeholder: the fused search is not used
        else:
            raise TypeError("policy_value_function must be callable (or 'uniform32' / 'hash')")
        self._engine = None
        self._engine_key = None
        self._weights_version = -1
        self.player = None

    def _get_engine(self, board):
        key = (board.width, board.height, board.n_in_row)
        if self._engine is None or self._engine_key != key:
            if board.width != board.height:
                raise ValueError("only square boards are supported by the HIP engine")
            self._engine = SelfPlayEngine(board.width, board.n_in_row, self._n_playout, 1, self._c_puct,
                                          device=self._device, prior_mode=self._mode)
            self._engine_key = key
            self._weights_version = -1
            self._engine.reset_games()
        if self._net is not None:
            self._weights_version = self._net.sync_engine(self._engine, self._weights_version)
        return self._engine

    def set_player_ind(self, p):
        self.player = p

    def reset_player(self):
        """mcts_alphaZero.py:208-209: mcts.update_with_move(-1)"""
        if self._engine is not None:
            self._engine.update_with_move([-1])

    def get_move_probs(self, board, temp=1e-3):
        """MCTS.get_move_probs (mcts_alphaZero.py:154-181): (acts, probs)."""
        eng = self._get_engine(board)
        eng.set_position(0, board.states, board.last_move, board.current_player)
        if self._callback is not None:
            self._callback_playouts(eng, board)
        else:
            eng.search()
        visits, _, _, _ = eng.root_stats()
        acts = list(board.availables)
        probs = softmax(1.0 / temp * np.log(np.array(visits[0][acts], dtype=np.int64) + 1e-10))
        return acts, probs

    def _callback_playouts(self, eng, board):
        """n_playout times MCTS._playout (mcts_alphaZero.py:113-148) with the policy evaluated here."""
        from .board import Board
        S = board.width * board.height
        for _ in range(self._n_playout):
            s1, s2, _, last, cur, term = eng.playout_begin(0)
            if term:
                eng.playout_end(None, 0.0)
                continue
            leaf = Board.from_stones(board.width, board.height, board.n_in_row, s1, s2, last, cur)
            action_probs, value = self._callback(leaf)
            priors = np.zeros(S, dtype=np.float32)
            for a, p in action_probs:
                priors[a] = p
            eng.playout_end(priors, float(np.float32(value)))

    def get_action(self, board, temp=1e-3, return_prob=0):
        """mcts_alphaZero.py:212-262"""
        sensible_moves = board.availables
        move_probs = np.zeros(board.width * board.height)
        if len(sensible_moves) == 0:
            _logger.info("WARNING: the board is full")
            return None
        acts, probs = self.get_move_probs(board, temp)
        move_probs[list(acts)] = probs
        if self._is_selfplay:
            # Dirichlet noise for exploration; the draw order (dirichlet, then choice's
            # uniform) is the reference's, on the global numpy stream
            move = np.random.choice(acts, p=0.75 * probs + 0.25 * np.random.dirichlet(0.1 * np.ones(len(probs))))
            self._engine.update_with_move([int(move)])            # tree reuse
        else:
            move = np.random.choice(acts, p=probs)
            # the reference opens in the 3x3 centre of a 15x15 board (mcts_alphaZero.py:251-253,
            # hard-coded indices that are illegal on smaller boards); kept for 15x15 only
            if len(board.states) == 0 and board.width == 15 and board.height == 15:
                move = random.choice([96, 97, 98, 111, 112, 113, 126, 127, 128])
            self._engine.update_with_move([-1])
        if return_prob:
            return move, move_probs
        return move

    def __str__(self):
        return "MCTS {}".format(self.player)
