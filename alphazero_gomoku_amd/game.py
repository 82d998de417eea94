"""Game server with the reference's surface (game.py:175-416) plus the batched entry point."""
import logging

import numpy as np

from .board import Board  # noqa: F401  (re-exported: the reference keeps Board and Game together)
from .engine import SelfPlayEngine

_logger = logging.getLogger(__name__)


class Game(object):
    def __init__(self, board, **kwargs):
        self.board = board

    def graphic(self, board, player1, player2):
        """ASCII board, row 0 at the bottom (game.py:181-209)."""
        print("Player", player1, "with X".rjust(3))
        print("Player", player2, "with O".rjust(3))
        print()
        for i in range(board.height - 1, -1, -1):
            cells = []
            for j in range(board.width):
                p = board.states.get(i * board.width + j, -1)
                cells.append('X' if p == player1 else 'O' if p == player2 else '-')
            print("%4d " % i + "  ".join(cells))
        print("     " + " ".join("%2d" % x for x in range(board.width)))

    def start_play(self, player1, player2, start_player=0, is_shown=1):
        """game.py:211-290: a game between two players; returns the winner (1, 2 or -1)."""
        if start_player not in (0, 1):
            raise Exception('start_player should be either 0 (player1 first) or 1 (player2 first)')
        self.board.init_board(start_player)
        p1, p2 = self.board.players
        player1.set_player_ind(p1)
        player2.set_player_ind(p2)
        players = {p1: player1, p2: player2}
        if is_shown:
            self.graphic(self.board, player1.player, player2.player)
        while True:
            player_in_turn = players[self.board.get_current_player()]
            move = player_in_turn.get_action(self.board)
            self.board.do_move(move)
            if is_shown:
                self.graphic(self.board, player1.player, player2.player)
            end, winner = self.board.game_end()
            if end:
                return winner

    def start_self_play(self, player, is_shown=0, temp=1e-3):
        """game.py:360-416: one self-play game; returns (winner, zip(states, mcts_probs, winners_z))."""
        self.board.init_board()
        p1, p2 = self.board.players
        states, mcts_probs, current_players = [], [], []
        while True:
            move, move_probs = player.get_action(self.board, temp=temp, return_prob=1)
            states.append(self.board.current_state())
            mcts_probs.append(move_probs)
            current_players.append(self.board.current_player)
            self.board.do_move(move)
            if is_shown:
                self.graphic(self.board, p1, p2)
            end, winner = self.board.game_end()
            if end:
                winners_z = np.zeros(len(current_players))
                if winner != -1:
                    winners_z[np.array(current_players) == winner] = 1.0
                    winners_z[np.array(current_players) != winner] = -1.0
                player.reset_player()
                if is_shown:
                    print("Game end. Winner is player:" if winner != -1 else "Game end. Tie",
                          winner if winner != -1 else "")
                return winner, zip(states, mcts_probs, winners_z)

    def start_self_play_batch(self, net, n_games, n_playout=400, c_puct=5, temp=1.0, seed0=0,
                              n_parallel=1024, device=0, engine=None):
        """n_games self-play games, n_parallel of them in flight on one GPU.  Game k draws its
        exploration noise from numpy's legacy RandomState seeded with seed0 + k, i.e. it
        is the game `np.random.seed(seed0 + k); start_self_play(...)` would produce.
        Returns a list of (game_id, winner-aware tuples) = (id, states, mcts_probs, winners_z)."""
        b = self.board
        eng = engine or SelfPlayEngine(b.width, b.n_in_row, n_playout, min(n_parallel, n_games), c_puct, device=device)
        if net is not None:
            net.sync_engine(eng, -1)
        res = eng.play_games(n_games, temp=temp, seed0=seed0)
        s = res["samples"]
        out = []
        if s is not None:
            order = np.lexsort((s["ply"], s["game_id"]))
            ids = s["game_id"][order]
            for gid in np.unique(ids):
                sel = order[ids == gid]
                out.append((int(gid), s["states"][sel].astype(np.float64), s["pi"][sel], s["z"][sel].astype(np.float64)))
        return out
