"""Host-side Board with the reference's interface (game.py:11-172).

The position is kept as two Python-int stone sets (bit m = cell m); `states` and
`availables` are maintained next to them because callers of the reference read those
attributes directly (mcts_alphaZero.py:213-216, policy_value_net_pytorch.py:102).
This object is what a caller hands to MCTSPlayer.get_action; the playouts themselves
run on the GPU from a packed copy of it."""
import numpy as np


class Board(object):
    def __init__(self, **kwargs):
        self.width = int(kwargs.get('width', 8))
        self.height = int(kwargs.get('height', 8))
        self.n_in_row = int(kwargs.get('n_in_row', 5))
        self.players = [1, 2]
        self.states = {}
        self._stones = [0, 0]
        self.move_target = []

    # game.py:25-58
    def move_in_rect(self, move, rect):
        row, col = move // self.width, move % self.width
        return rect[1] <= row <= rect[0] and rect[2] <= col <= rect[3]

    def get_move_target(self):
        """The cells the rollout opponent considers: empty cells inside the stones' bounding box
        grown by two (rows and columns 5..9 on an empty board)."""
        if not self.states:
            top, bottom, left, right = 9, 5, 5, 9
        else:
            rows = [m // self.width for m in self.states]
            cols = [m % self.width for m in self.states]
            top, bottom, left, right = max(rows), min(rows), min(cols), max(cols)
            step = 2
            top = top + step if top <= self.height - step else self.height
            bottom = bottom - step if bottom >= step else 0
            left = left - step if left >= step else 0
            right = right + step if right <= self.width - step else self.width
        self.move_target = [x for x in self.availables if self.move_in_rect(x, [top, bottom, left, right])]

    # game.py:60-68
    def init_board(self, start_player=0):
        if self.width < self.n_in_row or self.height < self.n_in_row:
            raise Exception('board width and height can not be less than {}'.format(self.n_in_row))
        self.current_player = self.players[start_player]
        self.availables = list(range(self.width * self.height))
        self.states = {}
        self.last_move = -1
        self._stones = [0, 0]

    @classmethod
    def from_stones(cls, width, height, n_in_row, stones1, stones2, last_move, current_player):
        """The board object a policy_value_fn sees for a position given as two stone sets
        (Python ints, bit m = cell m): states, availables, last_move and current_player as
        Board.do_move would have left them."""
        b = cls(width=width, height=height, n_in_row=n_in_row)
        b._stones = [int(stones1), int(stones2)]
        b.states = {}
        for m in range(width * height):
            if (stones1 >> m) & 1:
                b.states[m] = 1
            elif (stones2 >> m) & 1:
                b.states[m] = 2
        b.availables = [m for m in range(width * height) if m not in b.states]
        b.last_move = last_move
        b.current_player = current_player
        return b

    # game.py:70-90
    def move_to_location(self, move):
        return [move // self.width, move % self.width]

    def location_to_move(self, location):
        if len(location) != 2:
            return -1
        h, w = location
        move = h * self.width + w
        return move if 0 <= move < self.width * self.height else -1

    # game.py:119-127
    def do_move(self, move):
        self.availables.remove(move)          # ValueError on an occupied cell, as the reference
        self.states[move] = self.current_player
        self._stones[self.current_player - 1] |= 1 << int(move)
        self.current_player = 3 - self.current_player
        self.last_move = move

    def _anchors(self, stones):
        """Bit set of cells that start an n-long ray of `stones` in one of the four scan
        directions of game.py:144-158 (right, up, up-right, up-left)."""
        W, H, n = self.width, self.height, self.n_in_row
        out = 0
        for stride, need_right, need_left in ((1, True, False), (W, False, False),
                                              (W + 1, True, False), (W - 1, False, True)):
            valid = 0
            for h in range(H):
                if stride != 1 and h > H - n:
                    continue
                for w in range(W):
                    if need_right and w > W - n:
                        continue
                    if need_left and w < n - 1:
                        continue
                    valid |= 1 << (h * W + w)
            run = stones & valid
            for j in range(1, n):
                run &= stones >> (j * stride)
            out |= run
        return out

    # game.py:129-160
    def has_a_winner(self):
        if len(self.states) < self.n_in_row + 2:
            return False, -1
        a1, a2 = self._anchors(self._stones[0]), self._anchors(self._stones[1])
        if not (a1 | a2):
            return False, -1
        low1 = (a1 & -a1) if a1 else 1 << 4096
        low2 = (a2 & -a2) if a2 else 1 << 4096
        return True, (1 if low1 < low2 else 2)

    # game.py:162-169
    def game_end(self):
        win, winner = self.has_a_winner()
        if win:
            return True, winner
        if not len(self.availables):
            return True, -1
        return False, -1

    # game.py:92-116
    def current_state(self):
        W, H = self.width, self.height
        sq = np.zeros((4, W, H))
        if self.states:
            for m, p in self.states.items():
                sq[0 if p == self.current_player else 1][m // W, m % H] = 1.0
            sq[2][self.last_move // W, self.last_move % H] = 1.0
        if len(self.states) % 2 == 0:
            sq[3][:, :] = 1.0
        return sq[:, ::-1, :]

    # game.py:171-172
    def get_current_player(self):
        return self.current_player
