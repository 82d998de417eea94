"""Soak run: python profiles/soak.py BOARD PLAYOUTS GAMES STEPS [MODEL] -- pipelined self-play for STEPS move steps,
samples fetched and dropped, one JSON progress line every 2000 steps (so that a long run is seen to be alive).
MODEL: a weight file for PolicyValueNet (a trained net: deep trees, most of the tree kept from move to move)
instead of the random-init net."""
import sys, time, json
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch, bench
import alphazero_gomoku_amd as az
from alphazero_gomoku_amd.engine import SelfPlayRunner
n, k, playouts, games, steps = int(sys.argv[1]), 5, int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
dev = torch.device("cuda", 0)
params = bench.random_init_params(n, dev)
eng = az.SelfPlayEngine(n, k, playouts, games, 5.0, device=0)
if len(sys.argv) > 5 and sys.argv[5].endswith(".npz"):          # the fixture of the reference's trained nets
    eng.set_weights(bench.trained_params(sys.argv[5], n, dev))
elif len(sys.argv) > 5:
    from alphazero_gomoku_amd.net import PolicyValueNet
    PolicyValueNet(n, n, model_file=sys.argv[5]).sync_engine(eng, -1)
else:
    eng.set_weights(params)
count = [0, 0]
def on(chunk):
    count[0] += int(chunk["z"].shape[0]); count[1] += 1
run = SelfPlayRunner(eng, temp=1.0, seed0=0, total=None, on_samples=on, device_samples=True, pipelined=True)
run.start()
t0 = time.time()
for s in range(steps):
    run.step()
    if (s + 1) % 2000 == 0:
        print(json.dumps({"steps": s + 1, "games": run.games_done, "tuples": count[0], "seconds": time.time() - t0, "arena_stats": eng.arena_stats(), "reroot_stats": eng.reroot_stats()}), flush=True)
run.finish()
print(json.dumps({"done": True, "steps": steps, "games": run.games_done, "plies": run.plies_done, "tuples": count[0], "seconds": time.time() - t0,
                  "games_per_s": run.games_done / (time.time() - t0), "arena_stats(moved_to_large, trees_dropped)": eng.arena_stats(),
                  "reroot_stats(in_place, compacted)": eng.reroot_stats()}), flush=True)
