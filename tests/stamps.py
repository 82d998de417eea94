"""Diagnostic: phase shares of the search kernel (needs a build with -DAZG_STAMPS)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import helpers
import alphazero_gomoku_amd as az
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8
# A stamp build carries a branch of one thread behind the matrix loops; whether THIS build still
# computes the exact searches is checked first and printed with the figures (tests/hazard_check.py:
# identical searches against the evaluation kernel and the oracle's float64 net).
import hazard_check
_ex = hazard_check.check(n, 256, 8)
print("exactness of this build: root priors differ from the evaluation kernel in %d of %d slots, trees differ in %d, "
      "evaluation kernel vs oracle net %.1e, search vs oracle net %.1e -> %s"
      % (_ex["root_priors_differ_from_eval_kernel"], _ex["slots"], _ex["trees_differ_from_the_most_common"],
         _ex["eval_kernel_max_abs_err_vs_oracle_net"], _ex["search_root_priors_max_abs_err_vs_oracle_net"],
         "EXACT" if _ex["root_priors_differ_from_eval_kernel"] == 0 and _ex["trees_differ_from_the_most_common"] == 0
         and _ex["eval_kernel_max_abs_err_vs_oracle_net"] < 1e-5 else "NOT EXACT: the figures below are of searches that differ from the shipped ones"))
eng = az.SelfPlayEngine(n, 5, int(os.environ.get("PLAYOUTS", 400 if n == 8 else 200)), B, cnn_mode=int(os.environ.get("CNN_MODE", "0")))
if os.environ.get("MODEL", "").endswith(".npz"):     # the reference's trained nets as the fixture holds them (Theano layout)
    with np.load(os.environ["MODEL"]) as _d:
        eng.set_weights([_d["params_%d_%d" % (n, i)] for i in range(16)], layout=az._lib.LAYOUT_THEANO)
elif os.environ.get("MODEL"):              # a trained net instead of random weights: deep, narrow trees
    from alphazero_gomoku_amd.net import PolicyValueNet
    PolicyValueNet(n, n, model_file=os.environ["MODEL"]).sync_engine(eng, -1)
else:
    eng.set_weights(helpers.make_params(n, seed=100 + n))
from alphazero_gomoku_amd.engine import SelfPlayRunner
run = SelfPlayRunner(eng, temp=1.0, seed0=0, total=None, on_samples=None, device_samples=False)
run.start()
WARM = int(os.environ.get("WARM", "40"))        # steps before the measured ones (finished games are restarted)
for _ in range(WARM):
    run.step()
NW = 32 + 2 * 2048
out = (C.c_longlong * NW)()
eng.L.azg_debug_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_longlong)]
eng.L.azg_debug_stamps(eng.h, out)
eng.profile()
for _ in range(4):
    run.step()
eng.L.azg_debug_stamps(eng.h, out)
l, ms, ev = eng.profile()
names = ["select+end", "sync->cnn", "fill", "conv1", "conv2", "conv3+heads", "sync", "headsum", "fc", "softmax", "expand+backup", "conv3 k-loop", "conv1 setup", "conv1 k-loop", "conv2 setup", "conv2 k-loop", "conv1 epilogue", "conv2 epilogue"]
ngroups = (B + 1) // 2 if (n <= 8 or os.environ.get("DUAL")) else B          # workgroups per launch
per_group = 4 if n <= 8 else (2 if os.environ.get("DUAL") else 1)             # playouts per workgroup iteration
raw = list(out)
out = [v / ngroups for v in out]
depth_sum = out[18]; out[18] = 0.0                     # slot 18 counts path entries (root .. leaf) of wave 0's descents
tot = sum(out[:28])
print("launches", l, "ms/launch", ms / l, "cycles/playout", tot / (4 * eng.n_playout),
      "clock GHz", out[30] / max(1, out[31]) * 0.1, "path entries per descent (root .. leaf)", depth_sum / (4 * eng.n_playout))
if os.environ.get("DUAL"):
    # two-wave-group kernel: thread 0 stamps the net waves, thread 256 the tree waves; per leaf evaluation
    names = {1: "net: wait for a leaf", 14: "net: conv2 setup", 15: "net: conv2 k-loop", 17: "net: conv2 epilogue", 4: "net: sync + a1gen",
             11: "net: conv3 k-loop", 5: "net: conv3 epilogue + heads", 10: "net: preload + sync + ack",
             0: "tree: wait for the net", 18: "tree: head sums -> lists", 19: "tree: policy FC", 3: "tree: value FC", 8: "tree: sync after FC", 9: "tree: softmax + sync", 2: "tree: expand + backup", 6: "tree: select + end test",
             7: "tree: input planes", 12: "tree: wait for A1", 13: "tree: conv1", 16: "tree: terminal backup"}
    ev_per_group = ev / l / ngroups if l else 1
    print("evaluations per workgroup and launch %.0f" % ev_per_group)
    for grp in ("net", "tree"):
        tot_g = sum(raw[i] for i, nm in names.items() if nm.startswith(grp)) / ngroups / l / ev_per_group
        print("%s waves: %.0f cycles per evaluation" % (grp, tot_g))
        for i, nm in names.items():
            if nm.startswith(grp):
                print("  %-30s %8.0f" % (nm, raw[i] / ngroups / l / ev_per_group))
    names = []
for i, nm in enumerate(names):
    print("%-14s %8.0f cyc/playout  %5.1f%%" % (nm, out[i] / (4 * eng.n_playout), 100.0 * out[i] / tot))
if not os.environ.get("DUAL") and sum(out[20:24]) > 0:
    # inside the descent (wave 0's, per level below the root): slots 20-23 of wave_select
    levels = depth_sum - 4 * eng.n_playout
    print("inside the descent: %.2f levels per descent; cycles per level:" % (levels / (4 * eng.n_playout)))
    for i, nm in zip(range(20, 24), ("path record of the level above + loop head", "loads issued -> float32 screen evaluated",
                                      "double-precision pass (near-ties only) + argmax", "hand-over of the chosen child (readlanes)")):
        print("  %-52s %7.0f   (%6.0f per playout)" % (nm, out[i] / levels, out[i] / (4 * eng.n_playout)))
    print("  %-52s           (%6.0f per playout)" % ("rest of select + end test (slot 0)", out[0] / (4 * eng.n_playout)))
wg = np.array(raw[32:32 + 2 * ngroups], dtype=np.int64).reshape(-1, 2)
ms_wg = wg[:, 0] * 1e-5
xcc = (wg[:, 1] >> 32) & 0xF
hw = wg[:, 1] & 0xFFFFFFFF
cu = (hw >> 8) & 0xF
se = (hw >> 13) & 0x7
print("loop time per workgroup of the last launch, ms: mean %.3f  min %.3f  max %.3f" % (ms_wg.mean(), ms_wg.min(), ms_wg.max()))
for x in range(8):
    m = xcc == x
    if m.any():
        print("  xcc %d: %3d workgroups  mean %.3f  max %.3f" % (x, m.sum(), ms_wg[m].mean(), ms_wg[m].max()))
key = xcc * 1000 + se * 32 + ((hw >> 12) & 1) * 16 + cu
cnt = np.bincount(np.unique(key, return_counts=True)[1])
print("workgroups per (xcc, se, cu):", {i: int(c) for i, c in enumerate(cnt) if c})
order = np.argsort(key, kind="stable")
pairs = ms_wg[order].reshape(-1, 2)
pairs.sort(axis=1)
print("per CU (two workgroups): faster mean %.3f, slower mean %.3f; slower - faster: mean %.3f max %.3f" % (
    pairs[:, 0].mean(), pairs[:, 1].mean(), (pairs[:, 1] - pairs[:, 0]).mean(), (pairs[:, 1] - pairs[:, 0]).max()))
print("histogram of workgroup loop times (ms):", np.histogram(ms_wg, bins=10)[0].tolist(), np.round(np.histogram(ms_wg, bins=10)[1], 2).tolist())
print("histogram of the slower workgroup per CU:", np.histogram(pairs[:, 1], bins=8)[0].tolist(), np.round(np.histogram(pairs[:, 1], bins=8)[1], 2).tolist())
