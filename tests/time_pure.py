#!/usr/bin/env python3
"""Times azg_pure_search (the rollout opponent) on the GPU and, beside it, the CPU oracle's
restatement of the same search on one host core.

    python tests/time_pure.py [--games 1024] [--playouts 1000] [--board 8]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--games", type=int, default=1024)
    ap.add_argument("--playouts", type=int, default=1000)
    ap.add_argument("--board", type=int, default=8)
    ap.add_argument("--n-in-row", type=int, default=5)
    ap.add_argument("--reps", type=int, default=3)
    args = ap.parse_args()
    import alphazero_gomoku_amd as az
    n, k, B, npo = args.board, args.n_in_row, args.games, args.playouts
    eng = az.SelfPlayEngine(n, k, npo, B, 5.0, prior_mode=az.PRIOR_UNIFORM32, arena_nodes=npo + 8)
    eng.reset_games()
    eng.set_position(0, {n * (n // 2) + n // 2: 1}, n * (n // 2) + n // 2, 2)
    words = np.zeros((B, 625), dtype=np.uint32)
    for g in range(B):
        st = np.random.RandomState(g).get_state()
        words[g, :624], words[g, 624] = st[1], st[2]
    eng.pure_search(npo, words)                              # warm-up
    t0 = time.perf_counter()
    for _ in range(args.reps):
        eng.pure_search(npo, words)
    dt = (time.perf_counter() - t0) / args.reps
    out = {"board": n, "games": B, "playouts_per_search": npo, "ms_per_launch": dt * 1e3,
           "rollout_playouts_per_s": B * npo / dt}
    from oracle import az_oracle
    az_oracle.build()
    b = az_oracle.OracleBoard(n, n, k)
    b.init_board(0)
    rng = az_oracle.OracleRng(0)
    t0 = time.perf_counter()
    az_oracle.pure_get_move(b, 5.0, npo, rng)
    dt1 = time.perf_counter() - t0
    out["cpu_port_one_core_playouts_per_s"] = npo / dt1
    print(json.dumps(out))
    eng.close()


if __name__ == "__main__":
    main()
