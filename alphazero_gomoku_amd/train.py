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
import os
import random
import sys
import time

import numpy as np

from .board import Board
from .engine import SelfPlayEngine, SelfPlayRunner
from .game import Game
from .train_data import DeviceReplayBuffer


class TrainPipeline:
    """Keyword arguments beyond the reference's hyper-parameters:
    n_parallel        game slots of the engine (per rank)
    updates_per_batch policy_update calls after every collect_selfplay_data(play_batch_size).  The
                      reference's loop (train.py:306-320) makes ONE call per collected batch and its
                      configuration collects one game per batch (play_batch_size: 1), i.e. one update
                      per game; here a batch is play_batch_size games played side by side, so
                      updates_per_batch = play_batch_size reproduces that ratio and 1 (the default)
                      reproduces the reference's code path for the given play_batch_size.
    continuous        False: every batch is a cohort of play_batch_size fresh games played to the end
                      with the weights of that moment (what the reference does, one game at a time).
                      True: the slots never idle -- a finished game is replaced at once, a batch is the
                      next play_batch_size games to finish, games in flight go on across policy updates
                      (they see new weights from their next move on), and the policy updates of batch k
                      are issued between the move steps of batch k+1 (one per move step on a single
                      GPU; with several ranks all of them after the first move step, see _between_steps).
    rng               source of the mini-batch draw (random.sample): a private random.Random by
                      default; pass the `random` module for the reference's global stream.
    max_lr_multiplier the multiplier is raised (x1.5) only while it is below this; 10 is the reference's
                      rule (train.py:199), under which it reaches 11.39.  With one game per update and a
                      3000-tuple buffer (the reference's configuration) the KL it measures grows before
                      that matters; with mini-batches drawn from hundreds of thousands of fresh tuples KL
                      stays below kl_targ/2 until learn_rate x 11.39 = 0.023 wrecks the net (measured: every
                      ReLU dead after seven updates, loss stuck at 5.15; a ceiling of 4, i.e. up to 5.06,
                      still ends that way; with 2, i.e. up to 2.25, the net of three runs in six was dead
                      within its first dozen updates and stayed so: KL = 0 and loss 5.16 for good; with 1
                      only a net that STARTED with a dead value head failed, see zero_head_biases).
                      Default (None): 10 with play_batch_size 1, else 1 (the multiplier starts at 1 and
                      can then only fall: the learning rate never exceeds learn_rate).
    zero_head_biases  a net created here (no init_model, no policy_value_net) gets the biases of its two
                      1x1 head convolutions zeroed (PolicyValueNet.zero_head_biases says why).  Default
                      (None): only when games are played in batches, where dozens of updates at the full
                      learning rate follow each other on a net that has not moved yet; with this and the
                      ceiling of 1, twenty of twenty differently seeded runs learned."""

    @staticmethod
    def controller_defaults(play_batch_size, max_lr_multiplier=None, zero_head_biases=None):
        """(ceiling of the learning-rate multiplier, whether a fresh net gets its head-convolution
        biases zeroed) for the given arguments.  With one game per batch both are the reference's
        (ceiling 10 as in train.py:199, initialisation untouched); batched play deliberately
        deviates (ceiling 1, zero head biases) unless told otherwise -- no reference fixture covers
        that regime."""
        batched = play_batch_size > 1
        ceiling = float(max_lr_multiplier) if max_lr_multiplier is not None else (1.0 if batched else 10.0)
        return ceiling, (batched if zero_head_biases is None else bool(zero_head_biases))

    def __init__(self, board_width=8, board_height=8, n_in_row=5, init_model=None, learn_rate=2e-3,
                 lr_multiplier=1.0, temp=1.0, n_playout=400, c_puct=5, buffer_size=10000, batch_size=512,
                 play_batch_size=1, epochs=5, kl_targ=0.02, check_freq=50, game_batch_num=1500,
                 n_parallel=1024, device=0, seed0=0, policy_value_net=None, pure_mcts_playout_num=1000,
                 updates_per_batch=1, continuous=False, rng=None, max_lr_multiplier=None, zero_head_biases=None):
        import torch
        from .net import PolicyValueNet
        self.board_width, self.board_height, self.n_in_row = board_width, board_height, n_in_row
        self.board = Board(width=board_width, height=board_height, n_in_row=n_in_row)
        self.game = Game(self.board)
        self.learn_rate, self.lr_multiplier, self.temp = learn_rate, lr_multiplier, temp
        self.n_playout, self.c_puct = n_playout, c_puct
        self.buffer_size, self.batch_size = buffer_size, batch_size
        self._rng = rng if rng is not None else random.Random(0x5EED + seed0)
        self.data_buffer = DeviceReplayBuffer(buffer_size, board_width, torch.device("cuda", device), self._rng)
        self.play_batch_size, self.epochs, self.kl_targ = play_batch_size, epochs, kl_targ
        self.check_freq, self.game_batch_num = check_freq, game_batch_num
        self.updates_per_batch, self.continuous = int(updates_per_batch), bool(continuous)
        self.max_lr_multiplier, want_zero = self.controller_defaults(play_batch_size, max_lr_multiplier, zero_head_biases)
        self.best_win_ratio = 0.2                                            # train.py:88
        self.pure_mcts_playout_num = pure_mcts_playout_num
        fresh = policy_value_net is None and not init_model
        self.policy_value_net = policy_value_net or PolicyValueNet(board_width, board_height, model_file=init_model,
                                                                   device=device)
        self.zeroed_head_biases = bool(fresh and want_zero)      # reported by the training program
        if self.zeroed_head_biases:
            self.policy_value_net.zero_head_biases()
        self.device, self.seed0, self.games_played = device, seed0, 0
        self.rank, self.world = 0, 1
        try:
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized():
                self.rank, self.world = dist.get_rank(), dist.get_world_size()
                self.policy_value_net.data_parallel = self.world > 1
        except ImportError:
            pass
        per_rank = (max(1, play_batch_size) + self.world - 1) // self.world
        self.engine = SelfPlayEngine(board_width, n_in_row, n_playout, min(n_parallel, per_rank), c_puct, device=device)
        if continuous and "AZG_SEARCH_PARTS" not in os.environ and board_width <= 8:
            # policy updates run between move steps, beside the search in flight: their kernels reach the
            # GPU between two launches of a search, so more, shorter launches (same trees) let them through
            # sooner: 2 731 / 2 788 / 2 906 games/s including training with 4 / 8 / 16 launches per move
            self.engine.set_search_parts(16)
        self._engine_version = -1
        self.episode_len = 0
        self.policy_updates = 0
        self.loss_history = []                   # (loss, entropy) of every policy update, in the order they ran
        self._runner = None                      # continuous mode: the self-play run that never stops
        self._chunks, self._games_in_chunks = [], 0
        self._todo = []                          # continuous mode: policy updates still to issue

    # ---- train.py:137-159 --------------------------------------------------------------------
    def _new_runner(self, first, total):
        return SelfPlayRunner(self.engine, temp=self.temp, seed0=self.seed0, first_game_id=first,
                              game_id_stride=self.world, total=total, on_samples=self._on_samples,
                              device_samples=True, pipelined=True, packed_samples=self.world > 1)

    def _on_samples(self, chunk):
        self._chunks.append(chunk)

    def _between_steps(self):
        """Continuous mode: one pending policy update per move step, issued while the next search
        (already enqueued by the step) runs; after the last one the engine gets the new weights."""
        if self._todo:
            # (AZG_UPDATES_BESIDE_SEARCH=1: diagnostic, the single-GPU schedule with several ranks as well)
            if self.world > 1 and os.environ.get("AZG_UPDATES_BESIDE_SEARCH") != "1":
                # Data-parallel updates hold collectives, and a collective's kernel gets no CU while a
                # search launch is resident (two workgroups per CU fill the LDS and the registers): every
                # one of the ~11 collectives of an update would wait for the end of a search.  So with
                # several ranks the search in flight is allowed to finish once, and all pending updates
                # run back to back before the next move step.
                import torch
                torch.cuda.synchronize(torch.device("cuda", self.device))
                while self._todo:
                    self._todo.pop()()
            else:
                self._todo.pop()()
            if not self._todo:
                self._engine_version = self.policy_value_net.sync_engine(self.engine, self._engine_version)

    def collect_selfplay_data(self, n_games=1):
        """Play n_games self-play games over all ranks (game id g draws its noise from the stream
        seeded seed0 + g), augment every tuple 8-fold (get_equi_data) and extend the replay buffer.
        Tuples stay in GPU memory from the engine to the buffer."""
        import torch
        mine = len(range(self.rank, n_games, self.world))        # this rank's share of the batch
        if self.continuous:
            if self._runner is None:
                self._engine_version = self.policy_value_net.sync_engine(self.engine, self._engine_version)
                self._runner = self._new_runner(self.rank, None)
                self._runner.start()
            run, target = self._runner, self._runner.games_delivered + mine
            while run.games_delivered < target:
                run.step()
                self._between_steps()
            while self.world > 1 and self._todo:                 # collectives below: every rank arrives with the same history
                self._between_steps()
        elif mine:
            self._engine_version = self.policy_value_net.sync_engine(self.engine, self._engine_version)
            run = self._new_runner(self.games_played + self.rank, mine)
            run.start()
            try:
                while run.games_done < mine:
                    run.step()
            finally:
                run.finish()
        self.games_played += n_games
        chunks, self._chunks = self._chunks, []
        dev = torch.device("cuda", self.device)
        keys = ("stones", "last", "pi", "z", "game_id", "ply") if self.world > 1 else ("states", "pi", "z", "game_id", "ply")
        if chunks:
            s = {k: torch.cat([c[k] for c in chunks]) for k in keys}
        else:
            empty = self.engine.selfplay_fetch(0, device=True, packed=self.world > 1)
            s = {k: empty[k] for k in keys}
        if self.world > 1:
            from . import dist as azdist
            if self.continuous:
                # The move step above has already enqueued the next search.  A collective's kernel gets no CU
                # beside a resident search launch, and every rank's would then wait for every other rank's
                # search: let the one in flight finish first (once per batch of games).
                torch.cuda.synchronize(dev)
            s = azdist.gather_packed(s, self.board_width * self.board_height)
            s["states"] = self.engine.unpack_states(s["stones"], s["last"], s["ply"])
        order = torch.argsort(s["game_id"] * 1024 + s["ply"].to(torch.int64))   # game after game, ply after ply
        states, pi, z = s["states"][order], s["pi"][order], s["z"][order]
        self.episode_len = int(z.shape[0])
        if self.episode_len:
            from .train_data import equi_data_arrays
            self.data_buffer.extend(*equi_data_arrays(self.engine, states, pi, z))

    # ---- train.py:161-214 --------------------------------------------------------------------
    def _draw(self):
        """The mini-batch positions: drawn once (rank 0) and shared, so that every rank trains on
        slices of the SAME mini-batch whatever else has consumed random numbers on it."""
        idx = self.data_buffer.draw(self.batch_size)
        if self.world > 1:
            import torch
            from . import dist as azdist
            idx = azdist.broadcast_indices(idx, torch.device("cuda", self.device))
        return idx

    def policy_update(self):
        import time
        import torch
        t_begin = time.perf_counter()
        state_batch, mcts_probs_batch, winner_batch = self.data_buffer.take(self._draw())
        net = self.policy_value_net
        on_device = torch.is_tensor(state_batch)
        # (device batches: the KL monitor's forward pass, and its sum, run in the library: policy_value_in_training)
        with_kl = on_device and hasattr(net, "policy_value_in_training")
        policy_value = net.policy_value_in_training if with_kl else net.policy_value
        old_probs, old_v = policy_value(state_batch)
        sl = slice(self.rank, None, self.world)                               # this rank's share of the mini-batch
        kl, loss, entropy = 0.0, float("nan"), float("nan")
        new_v = old_v
        fused = with_kl and self.world == 1 and getattr(net, "native_train", False) and hasattr(net, "train_epoch")
        for _ in range(self.epochs):
            if fused:                                                         # step + KL forward: one library call, one wait
                loss, entropy, new_probs, new_v, kl = net.train_epoch(state_batch, mcts_probs_batch, winner_batch,
                                                                      self.learn_rate * self.lr_multiplier, state_batch, old_probs)
                if kl > self.kl_targ * 4:
                    break
                continue
            loss, entropy = net.train_step(state_batch[sl], mcts_probs_batch[sl], winner_batch[sl],
                                           self.learn_rate * self.lr_multiplier)
            if with_kl:
                new_probs, new_v, kl = policy_value(state_batch, old_probs)
            else:
                new_probs, new_v = policy_value(state_batch)
                if on_device:
                    kl = float(torch.mean(torch.sum(old_probs.double() * (torch.log(old_probs.double() + 1e-10)
                                                                          - torch.log(new_probs.double() + 1e-10)), dim=1)))
                else:
                    kl = float(np.mean(np.sum(old_probs * (np.log(old_probs + 1e-10) - np.log(new_probs + 1e-10)), axis=1)))
            kl = self._agree(kl)
            if kl > self.kl_targ * 4:                                         # early stopping if D_KL diverges badly
                break
        if kl > self.kl_targ * 2 and self.lr_multiplier > 0.05:
            self.lr_multiplier /= 1.5
        elif kl < self.kl_targ / 2 and self.lr_multiplier < getattr(self, "max_lr_multiplier", 10.0):
            self.lr_multiplier *= 1.5
        # (explained variances, train.py:203-208: printed by the reference, read by nothing; computed when asked for)
        self._explained = (winner_batch, old_v, new_v)
        self.last_kl = kl
        self.policy_updates += 1
        # (host clock around the whole update; the KL value read above has waited for the device)
        self.update_seconds = getattr(self, "update_seconds", 0.0) + time.perf_counter() - t_begin
        self.__dict__.setdefault("loss_history", []).append((loss, entropy))
        return loss, entropy

    def _explained_var(self, which):
        winner_batch, old_v, new_v = self._explained
        v = (old_v, new_v)[which]
        if hasattr(winner_batch, "double"):
            wz, pv = winner_batch.double().cpu().numpy(), v.double().cpu().numpy().flatten()
        else:
            wz, pv = np.array(winner_batch), np.asarray(v).flatten()
        var_z = np.var(wz)
        return 1 - np.var(wz - pv) / var_z if var_z > 0 else float("nan")

    explained_var_old = property(lambda self: self._explained_var(0))
    explained_var_new = property(lambda self: self._explained_var(1))

    def _agree(self, kl):
        """Every rank must take the same early-stop and multiplier decisions (the train steps hold
        collectives): the largest KL any rank measured counts for all."""
        if self.world == 1:
            return kl
        import torch
        from . import dist as azdist
        return azdist.max_over_ranks(kl, torch.device("cuda", self.device))

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
    def run(self, on_batch=None, best_model=None, evaluate_games=10, evaluate_batched=False):
        """Self-play -> policy update, and every check_freq batches an evaluation against the rollout
        player: a new best win ratio saves the model to best_model (when given); a perfect score
        strengthens the opponent by 1000 playouts and resets the mark (train.py:324-338).
        evaluate_batched plays the evaluation games side by side (policy_evaluate(batched=True));
        every evaluation is appended to self.eval_history as (batch, rollout playouts, win ratio)."""
        self.eval_history = []
        for i in range(self.game_batch_num):
            self.collect_selfplay_data(self.play_batch_size)
            loss = entropy = None
            if len(self.data_buffer) > self.batch_size:
                if self.continuous:
                    # the updates of an earlier batch that self-play has not got round to come first
                    while self._todo:
                        self._between_steps()
                    self._last = (None, None)

                    def one():
                        self._last = self.policy_update()
                    self._todo = [one] * self.updates_per_batch
                    if i + 1 == self.game_batch_num:
                        while self._todo:
                            self._between_steps()
                    loss, entropy = self._last
                else:
                    for _ in range(self.updates_per_batch):
                        loss, entropy = self.policy_update()
            if on_batch is not None:
                on_batch(i, self.episode_len, loss, entropy)
            if self.check_freq and (i + 1) % self.check_freq == 0 and evaluate_games > 0:
                if self.rank == 0:
                    win_ratio = self.policy_evaluate(evaluate_games, batched=evaluate_batched, seed0=1000 * (i + 1))
                    self.eval_history.append((i + 1, self.pure_mcts_playout_num, win_ratio))
                    if win_ratio > self.best_win_ratio:
                        self.best_win_ratio = win_ratio
                        if best_model:
                            self.policy_value_net.save_model(best_model)
                        if self.best_win_ratio == 1.0 and self.pure_mcts_playout_num < 5000:
                            self.pure_mcts_playout_num += 1000
                            self.best_win_ratio = 0.0
                if self.world > 1:
                    import torch.distributed as dist
                    dist.barrier()                                # the other ranks wait here, not inside a later collective
        if self._runner is not None:
            self._runner.finish()
            self._runner = None
        return self.policy_value_net


def _hang_watchdog(seconds, rank, device):
    """Diagnostic (AZG_HANG_WATCHDOG=seconds in the environment): if the run is still going after that
    long, every thread's Python stack goes to stderr, then a helper thread waits for the device: if ALL
    work this process has enqueued on the GPU completes, the run is stuck on the host side (it says how
    long the wait took); if that wait does not come back within 20 s it is stuck on the device.  The
    process then exits with status 3."""
    import faulthandler
    import threading

    def watch():
        time.sleep(seconds)
        print("[watchdog rank %d] still running after %.0f s; stacks:" % (rank, seconds), file=sys.stderr, flush=True)
        faulthandler.dump_traceback(file=sys.stderr, all_threads=True)
        done = threading.Event()

        def wait_for_device():
            import torch
            t = time.perf_counter()
            torch.cuda.synchronize(device)
            print("[watchdog rank %d] all device work of this process completed (waited %.3f s): stuck on the host side"
                  % (rank, time.perf_counter() - t), file=sys.stderr, flush=True)
            done.set()
        threading.Thread(target=wait_for_device, daemon=True).start()
        if not done.wait(20.0):
            print("[watchdog rank %d] device work still pending after 20 s: stuck on the device" % rank, file=sys.stderr, flush=True)
        os._exit(3)
    threading.Thread(target=watch, daemon=True).start()


def main(argv=None):
    """The training loop as a program: `python -m alphazero_gomoku_amd.train --games 3000` on one
    GPU, or under `python -m torch.distributed.run --nproc-per-node N` with one rank per GPU
    (games shard by id over the ranks, tuples are gathered over RCCL, the policy update is
    data-parallel).  Rank 0 prints one JSON line with what was played and learned."""
    import argparse
    import json
    import os
    import sys
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
    ap.add_argument("--updates-per-batch", default="1",
                    help="policy updates after every self-play batch: a number, or 'games' for one per game of the batch "
                         "(the ratio of the reference's configuration, play_batch_size: 1)")
    ap.add_argument("--continuous", action="store_true",
                    help="never idle a slot: finished games are replaced at once, policy updates run between move steps")
    ap.add_argument("--slots", type=int, default=1024, help="game slots per GPU")
    ap.add_argument("--final-eval", type=int, default=0,
                    help="after the loop, play this many games against the rollout player (batched arena) and report the win ratio")
    ap.add_argument("--eval-games", type=int, default=10, help="games per evaluation of --check-freq")
    ap.add_argument("--eval-batched", action="store_true", help="play the evaluation games side by side on the GPU")
    ap.add_argument("--lr-multiplier", type=float, default=1.0, help="start value of the adaptive learning-rate multiplier")
    ap.add_argument("--max-lr-multiplier", type=float, default=None,
                    help="the KL controller raises the multiplier only below this (the reference's rule is 10, which "
                         "lets the learning rate reach 0.023: too much for mini-batches of fresh tuples from 1024 games; "
                         "default 1, or 10 with --play-batch 1)")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="every rank uses GPU 0 and the collectives go over gloo: runs the multi-rank code path on a "
                         "one-GPU machine (RCCL refuses two ranks on one device); not a measurement")
    args = ap.parse_args(argv)
    os.environ.setdefault("NCCL_DEBUG", "WARN")            # keeps most of RCCL's start-up banner off stdout
    import torch
    import torch.distributed as dist
    rank, world, local = 0, 1, 0 if args.rehearse_on_one_gpu else int(os.environ.get("LOCAL_RANK", "0"))
    if "RANK" in os.environ:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        torch.cuda.set_device(local)
        if args.rehearse_on_one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        rank, world = dist.get_rank(), dist.get_world_size()
    torch.manual_seed(0)                                   # every rank starts from the same weights
    random.seed(1234)                                      # and samples the same mini-batches
    per_rank = args.slots
    upb = args.play_batch if args.updates_per_batch == "games" else int(args.updates_per_batch)
    tp = TrainPipeline(board_width=args.board, board_height=args.board, n_in_row=args.n_in_row, learn_rate=args.lr,
                       n_playout=args.playouts, buffer_size=args.buffer_size, batch_size=args.batch_size,
                       play_batch_size=args.play_batch, epochs=args.epochs, check_freq=args.check_freq,
                       game_batch_num=(args.games + args.play_batch - 1) // args.play_batch, n_parallel=per_rank,
                       device=local, pure_mcts_playout_num=args.pure_playouts, updates_per_batch=upb,
                       continuous=args.continuous, lr_multiplier=args.lr_multiplier,
                       max_lr_multiplier=args.max_lr_multiplier)
    if world > 1:
        from . import dist as azdist
        azdist.broadcast_weights(tp.policy_value_net)
    # One-time library initialisation stays outside the timer: the first optimiser step of a process
    # makes MIOpen pick (and on a fresh machine compile) its convolution kernels, seconds that say
    # nothing about the loop.  A scratch net of the same shape takes one step on zeros; the
    # pipeline's own net, optimizer state and random streams are untouched.
    from .net import PolicyValueNet
    scratch = PolicyValueNet(args.board, args.board, device=local)
    dev = torch.device("cuda", local)
    nb = max(1, args.batch_size // world)
    scratch.train_step(torch.zeros((nb, 4, args.board, args.board), device=dev), torch.full((nb, args.board * args.board), 1.0 / (args.board * args.board), device=dev),
                       torch.zeros(nb, device=dev), 1e-3)
    scratch.policy_value(torch.zeros((args.batch_size, 4, args.board, args.board), device=dev))
    del scratch
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    if os.environ.get("AZG_HANG_WATCHDOG"):
        _hang_watchdog(float(os.environ["AZG_HANG_WATCHDOG"]), rank, local)
    log = []
    t0 = time.perf_counter()
    def on_batch(i, n_samples, loss, ent):
        log.append((i, n_samples, loss, ent))
        if rank == 0 and (i + 1) % 8 == 0:                 # a sign of life for whoever watches a long run
            print("batch %d: %d games, loss %s, %.0f s" % (i + 1, tp.games_played, "%.3f" % loss if loss is not None else "-",
                                                         time.perf_counter() - t0), file=sys.stderr, flush=True)
    tp.run(on_batch=on_batch, evaluate_games=args.eval_games, evaluate_batched=args.eval_batched)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    final = None
    if rank == 0 and args.final_eval > 0:
        final = tp.policy_evaluate(args.final_eval, batched=True)
    if world > 1:
        dist.barrier()
    if rank == 0:
        if args.save:
            tp.policy_value_net.save_model(args.save)
        losses = [l for l, _ in tp.loss_history]
        print(json.dumps({"games": tp.games_played, "ranks": world, "seconds": dt, "games_per_s_incl_training": tp.games_played / dt,
                          "samples": int(sum(n for _, n, _, _ in log)), "policy_updates": tp.policy_updates,
                          "policy_update_ms_mean": (1e3 * getattr(tp, "update_seconds", 0.0) / tp.policy_updates
                                                    if tp.policy_updates else None),
                          "updates_per_batch": upb, "play_batch": args.play_batch, "continuous": bool(args.continuous),
                          "slots_per_gpu": tp.engine.B, "rehearsal_on_one_gpu": bool(args.rehearse_on_one_gpu),
                          "trees_moved_to_large_arena": tp.engine.arena_stats()[0], "trees_dropped_for_lack_of_one": tp.engine.arena_stats()[1],
                          "first_loss": losses[0] if losses else None, "last_loss": losses[-1] if losses else None,
                          "loss_curve": [round(losses[i], 3) for i in range(0, len(losses), max(1, len(losses) // 16))],
                          "entropy_first_last": [tp.loss_history[0][1], tp.loss_history[-1][1]] if tp.loss_history else None,
                          "win_ratio_vs_rollout_player": final, "rollout_playouts": tp.pure_mcts_playout_num,
                          "lr_multiplier": tp.lr_multiplier, "max_lr_multiplier": tp.max_lr_multiplier,
                          "zero_head_biases": tp.zeroed_head_biases,
                          "reference_controller": tp.max_lr_multiplier == 10.0 and not tp.zeroed_head_biases,
                          "best_win_ratio": tp.best_win_ratio,
                          "evaluations_batch_playouts_win_ratio": tp.eval_history,
                          "timer": "starts after one optimiser step on a scratch net (library initialisation), ends after the last update"}))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
