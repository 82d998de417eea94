"""TrainPipeline: the self-play / policy-update loop of the reference's train.py
(train.py:60-214, 278-330) on top of the HIP self-play engine.

What is kept: the constructor's hyper-parameters and their meaning, collect_selfplay_data
(self-play + 8-fold augmentation into the replay buffer), policy_update (mini-batch from the
buffer, `epochs` train steps with the KL early stop, the adaptive learning-rate multiplier and the
explained-variance report) and run().  What differs: games are played `play_batch_size` at a time
on the GPU instead of one by one; hyper-parameters come as keyword arguments instead of the YAML
file; policy_evaluate plays the rollout opponent only (the external Yixin engine of
train.py:231-245 is not supported).
With torch.distributed initialised every rank plays its share of the games, all ranks gather all
tuples, and a policy update is one data-parallel step (each rank its slice of the mini-batch)."""
import random

import numpy as np

from .board import Board
from .engine import SelfPlayEngine
from .game import Game
from .train_data import ReplayBuffer, equi_data_arrays


class TrainPipeline:
    def __init__(self, board_width=8, board_height=8, n_in_row=5, init_model=None, learn_rate=2e-3,
                 lr_multiplier=1.0, temp=1.0, n_playout=400, c_puct=5, buffer_size=10000, batch_size=512,
                 play_batch_size=1, epochs=5, kl_targ=0.02, check_freq=50, game_batch_num=1500,
                 n_parallel=1024, device=0, seed0=0, policy_value_net=None, pure_mcts_playout_num=1000):
        from .net import PolicyValueNet
        self.board_width, self.board_height, self.n_in_row = board_width, board_height, n_in_row
        self.board = Board(width=board_width, height=board_height, n_in_row=n_in_row)
        self.game = Game(self.board)
        self.learn_rate, self.lr_multiplier, self.temp = learn_rate, lr_multiplier, temp
        self.n_playout, self.c_puct = n_playout, c_puct
        self.buffer_size, self.batch_size = buffer_size, batch_size
        self.data_buffer = ReplayBuffer(buffer_size, board_width)            # deque(maxlen) of train.py:82
        self.play_batch_size, self.epochs, self.kl_targ = play_batch_size, epochs, kl_targ
        self.check_freq, self.game_batch_num = check_freq, game_batch_num
        self.best_win_ratio = 0.2                                            # train.py:88
        self.pure_mcts_playout_num = pure_mcts_playout_num
        self.policy_value_net = policy_value_net or PolicyValueNet(board_width, board_height, model_file=init_model,
                                                                   device=device)
        self.device, self.seed0, self.games_played = device, seed0, 0
        self.engine = SelfPlayEngine(board_width, n_in_row, n_playout, min(n_parallel, max(1, play_batch_size)),
                                     c_puct, device=device)
        self._engine_version = -1
        self.episode_len = 0
        self.rank, self.world = 0, 1
        try:
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized():
                self.rank, self.world = dist.get_rank(), dist.get_world_size()
                self.policy_value_net.data_parallel = self.world > 1
        except ImportError:
            pass

    # ---- train.py:137-159 --------------------------------------------------------------------
    def collect_selfplay_data(self, n_games=1):
        """Play n_games self-play games (ids continue from the previous call; game id g uses the
        noise stream seeded seed0 + g), augment every tuple 8-fold and extend the replay buffer."""
        self._engine_version = self.policy_value_net.sync_engine(self.engine, self._engine_version)
        first = self.games_played + self.rank
        mine = len(range(first, self.games_played + n_games, self.world))
        res = self.engine.play_games(mine, temp=self.temp, seed0=self.seed0, first_game_id=first,
                                     game_id_stride=self.world) if mine else dict(samples=None)
        self.games_played += n_games
        s = res["samples"]
        if s is None:
            s = dict(states=np.zeros((0, 4, self.board_height, self.board_width), dtype=np.float32),
                     pi=np.zeros((0, self.board_width * self.board_height)), z=np.zeros(0, dtype=np.float32),
                     game_id=np.zeros(0, dtype=np.int64), ply=np.zeros(0, dtype=np.int32))
        if self.world > 1:
            import torch
            from . import dist as azdist
            dev = torch.device("cuda", self.device)
            allt = azdist.gather_samples({k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in s.items()})
            s = {k: v.cpu().numpy() for k, v in allt.items()}
        order = np.lexsort((s["ply"], s["game_id"]))                          # game after game, ply after ply
        states, pi, z = s["states"][order], s["pi"][order], s["z"][order]
        self.episode_len = int(len(z))
        if len(z):
            self.data_buffer.extend(*equi_data_arrays(self.engine, states, pi, z))

    # ---- train.py:161-214 --------------------------------------------------------------------
    def policy_update(self):
        state_batch, mcts_probs_batch, winner_batch = self.data_buffer.sample(self.batch_size)
        net = self.policy_value_net
        old_probs, old_v = net.policy_value(state_batch)
        sl = slice(self.rank, None, self.world)                               # this rank's share of the mini-batch
        kl, loss, entropy = 0.0, float("nan"), float("nan")
        for _ in range(self.epochs):
            loss, entropy = net.train_step(state_batch[sl], mcts_probs_batch[sl], winner_batch[sl],
                                           self.learn_rate * self.lr_multiplier)
            new_probs, new_v = net.policy_value(state_batch)
            kl = float(np.mean(np.sum(old_probs * (np.log(old_probs + 1e-10) - np.log(new_probs + 1e-10)), axis=1)))
            if kl > self.kl_targ * 4:                                         # early stopping if D_KL diverges badly
                break
        if kl > self.kl_targ * 2 and self.lr_multiplier > 0.05:
            self.lr_multiplier /= 1.5
        elif kl < self.kl_targ / 2 and self.lr_multiplier < 10:
            self.lr_multiplier *= 1.5
        var_z = np.var(np.array(winner_batch))
        self.explained_var_old = 1 - np.var(np.array(winner_batch) - old_v.flatten()) / var_z if var_z > 0 else float("nan")
        self.explained_var_new = 1 - np.var(np.array(winner_batch) - new_v.flatten()) / var_z if var_z > 0 else float("nan")
        self.last_kl = kl
        return loss, entropy

    # ---- train.py:216-276 --------------------------------------------------------------------
    def policy_evaluate(self, n_games=10, batched=False, seed0=0):
        """Play the current policy (evaluation mode: no exploration noise, fresh tree per move)
        against the rollout MCTS player, sides alternating, and return the win ratio
        (wins + half the ties) / n_games.  Only for monitoring the progress of training.
        batched=False plays the games one after another on numpy's global stream, like the
        reference; batched=True plays them side by side on the GPU, game i on its own streams
        seeded seed0 + i (arena.play_arena_games)."""
        from collections import defaultdict
        if batched:
            from .arena import play_arena_games, win_ratio
            recs = play_arena_games(self.policy_value_net, n_games, self.board_width, self.n_in_row, self.n_playout,
                                    self.pure_mcts_playout_num, self.c_puct, seed0, self.device)
            self.last_win_cnt = {w: sum(r["winner"] == w for r in recs) for w in (1, 2, -1)}
            return win_ratio(recs)
        from .mcts import MCTSPlayer
        from .mcts_pure import MCTSPlayer as MCTS_Pure
        current_mcts_player = MCTSPlayer(self.policy_value_net.policy_value_fn, c_puct=self.c_puct,
                                         n_playout=self.n_playout, device=self.device)
        pure_mcts_player = MCTS_Pure(c_puct=5, n_playout=self.pure_mcts_playout_num, device=self.device)
        win_cnt = defaultdict(int)
        for i in range(n_games):
            winner = self.game.start_play(current_mcts_player, pure_mcts_player, start_player=i % 2, is_shown=0)
            win_cnt[winner] += 1
        self.last_win_cnt = dict(win_cnt)
        return 1.0 * (win_cnt[1] + 0.5 * win_cnt[-1]) / n_games

    # ---- train.py:278-340 --------------------------------------------------------------------
    def run(self, on_batch=None, best_model=None, evaluate_games=10):
        """Self-play -> policy update, and every check_freq batches an evaluation against the rollout
        player: a new best win ratio saves the model to best_model (when given); a perfect score
        strengthens the opponent by 1000 playouts and resets the mark (train.py:324-338)."""
        for i in range(self.game_batch_num):
            self.collect_selfplay_data(self.play_batch_size)
            loss = entropy = None
            if len(self.data_buffer) > self.batch_size:
                loss, entropy = self.policy_update()
            if on_batch is not None:
                on_batch(i, self.episode_len, loss, entropy)
            if self.check_freq and (i + 1) % self.check_freq == 0 and self.rank == 0 and evaluate_games > 0:
                win_ratio = self.policy_evaluate(evaluate_games)
                if win_ratio > self.best_win_ratio:
                    self.best_win_ratio = win_ratio
                    if best_model:
                        self.policy_value_net.save_model(best_model)
                    if self.best_win_ratio == 1.0 and self.pure_mcts_playout_num < 5000:
                        self.pure_mcts_playout_num += 1000
                        self.best_win_ratio = 0.0
        return self.policy_value_net


def main(argv=None):
    """The training loop as a program: `python -m alphazero_gomoku_amd.train --games 3000` on one
    GPU, or under `python -m torch.distributed.run --nproc-per-node N` with one rank per GPU
    (games shard by id over the ranks, tuples are gathered over RCCL, the policy update is
    data-parallel).  Rank 0 prints one JSON line with what was played and learned."""
    import argparse
    import json
    import os
    import time
    ap = argparse.ArgumentParser(description=main.__doc__)
    ap.add_argument("--board", type=int, default=8)
    ap.add_argument("--n-in-row", type=int, default=5)
    ap.add_argument("--playouts", type=int, default=400)
    ap.add_argument("--games", type=int, default=3000, help="self-play games in total")
    ap.add_argument("--play-batch", type=int, default=1024, help="games per self-play batch (all ranks together)")
    ap.add_argument("--batch-size", type=int, default=512)
    ap.add_argument("--buffer-size", type=int, default=200000)
    ap.add_argument("--epochs", type=int, default=5)
    ap.add_argument("--lr", type=float, default=2e-3)
    ap.add_argument("--check-freq", type=int, default=0, help="evaluate against the rollout player every this many batches (0 = never)")
    ap.add_argument("--pure-playouts", type=int, default=1000)
    ap.add_argument("--save", default=None, help="write the final state_dict here")
    args = ap.parse_args(argv)
    os.environ.setdefault("NCCL_DEBUG", "WARN")            # keeps most of RCCL's start-up banner off stdout
    import torch
    import torch.distributed as dist
    rank, world, local = 0, 1, int(os.environ.get("LOCAL_RANK", "0"))
    if "RANK" in os.environ:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        rank, world = dist.get_rank(), dist.get_world_size()
    torch.manual_seed(0)                                   # every rank starts from the same weights
    random.seed(1234)                                      # and samples the same mini-batches
    per_rank = max(1, (args.play_batch + world - 1) // world)
    tp = TrainPipeline(board_width=args.board, board_height=args.board, n_in_row=args.n_in_row, learn_rate=args.lr,
                       n_playout=args.playouts, buffer_size=args.buffer_size, batch_size=args.batch_size,
                       play_batch_size=args.play_batch, epochs=args.epochs, check_freq=args.check_freq,
                       game_batch_num=(args.games + args.play_batch - 1) // args.play_batch, n_parallel=per_rank,
                       device=local, pure_mcts_playout_num=args.pure_playouts)
    if world > 1:
        from . import dist as azdist
        azdist.broadcast_weights(tp.policy_value_net)
    log = []
    t0 = time.perf_counter()
    tp.run(on_batch=lambda i, n, loss, ent: log.append((i, n, loss, ent)))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if rank == 0:
        if args.save:
            tp.policy_value_net.save_model(args.save)
        losses = [l for _, _, l, _ in log if l is not None]
        print(json.dumps({"games": tp.games_played, "ranks": world, "seconds": dt, "games_per_s_incl_training": tp.games_played / dt,
                          "samples": int(sum(n for _, n, _, _ in log)), "policy_updates": len(losses),
                          "first_loss": losses[0] if losses else None, "last_loss": losses[-1] if losses else None,
                          "lr_multiplier": tp.lr_multiplier, "best_win_ratio": tp.best_win_ratio}))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
