"""Where the host time of TrainPipeline.policy_update goes: a replay buffer of synthetic tuples, 200 updates
under cProfile.  Usage (on a GPU box): python tests/time_update.py"""
import cProfile
import pstats
import sys
import time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from alphazero_gomoku_amd.train import TrainPipeline

tp = TrainPipeline(board_width=8, board_height=8, n_in_row=5, n_playout=400, play_batch_size=1024, buffer_size=200000, device=0)
g = torch.Generator(device="cuda").manual_seed(1)
n = 50000
states = (torch.rand((n, 4, 8, 8), device="cuda", generator=g) < 0.2).float()
pi = torch.softmax(3 * torch.randn((n, 64), device="cuda", generator=g), dim=1)
z = torch.sign(torch.randn(n, device="cuda", generator=g))
tp.data_buffer.extend(states, pi, z)
for _ in range(10):
    tp.policy_update()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200):
    tp.policy_update()
torch.cuda.synchronize()
print("policy_update: %.3f ms" % ((time.perf_counter() - t0) / 200 * 1e3))
pr = cProfile.Profile()
pr.enable()
for _ in range(200):
    tp.policy_update()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
