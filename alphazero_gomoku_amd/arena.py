"""Evaluation arena: the network-guided player against the rollout player, many games at a
time on one GPU.

The reference plays these games one after another on the global numpy stream
(TrainPipeline.policy_evaluate, train.py:216-276, through Game.start_play, game.py:211-290).
Here game i owns its own legacy streams, seeded seed0 + i, and is exactly the game

    np.random.seed(seed0 + i); random.seed(seed0 + i)
    Game(board).start_play(MCTSPlayer(net.policy_value_fn, c_puct, n_playout),
                           MCTS_Pure(5, pure_n_playout), start_player=i % 2)

would be.  Games with the same first mover advance in lock step in one engine: when the
network player is to move every slot runs the CNN-guided search, when the rollout player is to
move every slot runs azg_pure_search."""
import random

import numpy as np

from .engine import SelfPlayEngine
from .mcts import softmax
from .mcts_pure import numpy_stream_to_words

CENTRE_15 = [96, 97, 98, 111, 112, 113, 126, 127, 128]      # mcts_alphaZero.py:252


def _choice_index(probs, u):
    """np.random.choice(acts, p=probs) for one draw whose uniform is u."""
    cdf = np.cumsum(probs)
    cdf /= cdf[-1]
    return int(min(np.searchsorted(cdf, u, side="right"), len(cdf) - 1))


def _play_group(net, ids, start_player, n, n_in_row, n_playout, pure_n_playout, c_puct, seed0, device):
    B = len(ids)
    eng = SelfPlayEngine(n, n_in_row, n_playout, B, c_puct, device=device,
                         arena_nodes=max(4 * n_playout + 64, pure_n_playout + 8))
    net.sync_engine(eng, -1)
    eng.reset_games()
    rs = [np.random.RandomState((seed0 + i) & 0xFFFFFFFF) for i in ids]
    py = [random.Random(seed0 + i) for i in ids]
    moves_log = [[] for _ in ids]
    S = n * n
    # player 1 (stone 1) is the network player; Board.init_board(start_player) gives the first move
    # to players[start_player], and the engine's slots start with stone 1 to move
    if start_player == 1:
        for g in range(B):
            eng.set_position(g, {}, -1, 2)
    occupied = np.zeros((B, S), dtype=bool)
    ply = 0
    while True:
        ended, winner, _ = eng.status()
        live = ~ended
        if not live.any():
            break
        net_to_move = (ply % 2 == 0) == (start_player == 0)
        moves = np.full(B, -1, dtype=np.int32)
        if net_to_move:
            # MCTSPlayer.get_action, evaluation mode (mcts_alphaZero.py:212-255): fresh tree, search,
            # pi at temp 1e-3, one np.random.choice draw, no exploration noise
            eng.update_with_move(np.where(live, -1, -2).astype(np.int32))
            eng.search()
            visits, _, _, _ = eng.root_stats()
            for g in np.nonzero(live)[0]:
                acts = np.nonzero(~occupied[g])[0]
                probs = softmax(1.0 / 1e-3 * np.log(visits[g][acts].astype(np.int64) + 1e-10))
                move = int(acts[_choice_index(probs, rs[g].random_sample())])
                if ply == 0 and n == 15:
                    move = py[g].choice(CENTRE_15)
                moves[g] = move
        else:
            words = np.zeros((B, 625), dtype=np.uint32)
            states = [r.get_state() for r in rs]
            for g in range(B):
                words[g] = numpy_stream_to_words(states[g])
            moves = eng.pure_search(pure_n_playout, words)
            for g in np.nonzero(live)[0]:
                st = states[g]
                rs[g].set_state((st[0], words[g, :624].copy(), int(words[g, 624]), st[3], st[4]))
        for g in np.nonzero(live)[0]:
            moves_log[g].append(int(moves[g]))
            occupied[g, moves[g]] = True
        eng.advance(np.where(live, moves, -1).astype(np.int32))
        ply += 1
    _, winner, _ = eng.status()
    eng.close()
    return [dict(game=i, start_player=start_player, winner=int(winner[g]), moves=moves_log[g]) for g, i in enumerate(ids)]


def play_arena_games(net, n_games, board_n=8, n_in_row=5, n_playout=400, pure_n_playout=1000, c_puct=5.0, seed0=0,
                     device=0):
    """n_games evaluation games, sides alternating (game i: start_player = i % 2, train.py:264-266).
    Returns one record per game, in game order: dict(game, start_player, winner (1 = the network
    player, 2 = the rollout player, -1 = tie), moves)."""
    out = []
    for sp in (0, 1):
        ids = list(range(sp, n_games, 2))
        if ids:
            out += _play_group(net, ids, sp, board_n, n_in_row, n_playout, pure_n_playout, float(c_puct), int(seed0), device)
    return sorted(out, key=lambda r: r["game"])


def win_ratio(records):
    """train.py:272: (wins + half the ties) / games, from the network player's side."""
    if not records:
        return 0.0
    wins = sum(r["winner"] == 1 for r in records)
    ties = sum(r["winner"] == -1 for r in records)
    return 1.0 * (wins + 0.5 * ties) / len(records)
