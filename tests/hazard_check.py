"""Diagnostic (run by hand on a GPU box, not collected by pytest): B identical 15x15 searches of one
library build against the evaluation kernel of the SAME build.

    AZG_LIBRARY=abx/libazg_NAME.so [BOARD=15] python tests/hazard_check.py [B] [playouts]

Every slot starts the same game (same seed, empty board), so all B trees must be identical, and the
root's priors must be the evaluation kernel's probabilities of the empty board bit for bit.  Prints
one JSON line: how many slots differ, and for the first wrong slot which cells are off and by how
much (log ratio of the prior to the evaluation kernel's), which tells a missing weight row from a
missing partial sum."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def check(n, B=256, npo=8):
    """-> the record described above for board size n."""
    import helpers
    import alphazero_gomoku_amd as az
    from oracle import az_oracle
    k = 5
    params = helpers.make_params(n, seed=100 + n)
    eng = az.SelfPlayEngine(n, k, npo, B)
    eng.set_weights(params)
    ev = az.SelfPlayEngine(n, k, 1, 1)
    ev.set_weights(params)
    eng.reset_games(seeds=np.full(B, 7, dtype=np.uint32))
    eng.search()
    visits, q, p, rn = eng.root_stats()
    empty = np.zeros((1, 4, n, n), dtype=np.float32)
    empty[0, 3] = 1.0
    pr, v = ev.eval_batch(empty)
    want = pr[0].astype(np.float64)
    # the evaluation kernel of a build under test may be wrong in the same way: the oracle's float64 net is the outside reference
    po, vo = az_oracle.OracleNet(n, n, params, precise=True).forward(empty[0].astype(np.float64))
    eval_err = float(np.abs(want - po).max())
    search_err = float(max(np.abs(p[g] - po).max() for g in range(B)))
    wrong_prior = [g for g in range(B) if not np.array_equal(p[g], want)]
    # the most common tree is taken as the right one for the visit comparison
    keys = [visits[g].tobytes() + q[g].tobytes() for g in range(B)]
    mode = max(set(keys), key=keys.count)
    wrong_tree = [g for g in range(B) if keys[g] != mode]
    out = {"library": os.path.basename(os.environ.get("AZG_LIBRARY", "default")), "board": n, "slots": B, "playouts": npo,
           "eval_kernel_max_abs_err_vs_oracle_net": eval_err, "search_root_priors_max_abs_err_vs_oracle_net": search_err,
           "root_priors_differ_from_eval_kernel": len(wrong_prior), "trees_differ_from_the_most_common": len(wrong_tree),
           "first_wrong_slots": wrong_prior[:16]}
    if wrong_prior:
        g = wrong_prior[0]
        cells = np.flatnonzero(p[g] != want)
        out["first_wrong_slot_cells"] = cells.tolist()[:80]
        out["log_ratio"] = [float(np.log(p[g][c] / want[c])) for c in cells[:80]]
        out["cells_wrong_by_slot"] = {int(s): np.flatnonzero(p[s] != want).tolist()[:12] for s in wrong_prior[:8]}
    eng.close()
    ev.close()
    return out


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    npo = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    print(json.dumps(check(int(os.environ.get("BOARD", "15")), B, npo)))


if __name__ == "__main__":
    main()
