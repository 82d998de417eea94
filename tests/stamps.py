"""Diagnostic: phase shares of the search kernel (needs a build with -DAZG_STAMPS)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import helpers
import alphazero_gomoku_amd as az
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8
eng = az.SelfPlayEngine(n, 5, 400 if n == 8 else 200, B)
eng.set_weights(helpers.make_params(n, seed=100 + n))
eng.reset_games(seeds=np.arange(B, dtype=np.uint32))
for _ in range(6):
    eng.search(); eng.sample_and_advance(1.0, want_pi=False)
out = (C.c_longlong * 32)()
eng.L.azg_debug_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_longlong)]
eng.L.azg_debug_stamps(eng.h, out)
eng.profile()
for _ in range(4):
    eng.search(); eng.sample_and_advance(1.0, want_pi=False)
eng.L.azg_debug_stamps(eng.h, out)
l, ms, ev = eng.profile()
names = ["select+end", "sync->cnn", "fill", "conv1", "conv2", "conv3+heads", "sync", "headsum", "fc", "softmax", "expand+backup"]
tot = sum(out[:11])
print("launches", l, "ms/launch", ms / l, "cycles/playout", tot / (4 * eng.n_playout))
for i, nm in enumerate(names):
    print("%-14s %8.0f cyc/playout  %5.1f%%" % (nm, out[i] / (4 * eng.n_playout), 100.0 * out[i] / tot))
