"""azg_board_step_batch_static (the entry point that allocates nothing: its staging and its
illegal-move flag live in a fixed per-GPU array) against azg_board_step_batch, through the C
ABI, bit for bit, on two calls in a row with the second one smaller; and the property it exists
for: the GPU's free memory does not move across its calls."""
import numpy as np
import pytest

import helpers

pytestmark = pytest.mark.gpu


def _boards(n, k, count, seed, min_depth=0):
    """count positions of random legal play (not ended), with one more random legal move each."""
    rs = np.random.RandomState(seed)
    S = n * n
    st = np.zeros((count, 2, 4), dtype=np.uint64)
    ply = np.zeros(count, dtype=np.int32)
    last = np.full(count, -1, dtype=np.int32)
    moves = np.zeros(count, dtype=np.int32)
    player = np.zeros(count, dtype=np.int32)
    for i in range(count):
        cells = rs.permutation(S)
        depth = int(rs.randint(min_depth, S - 1))
        for t in range(depth):
            c = int(cells[t])
            st[i, t % 2, c >> 6] |= np.uint64(1) << np.uint64(c & 63)
        ply[i] = depth
        last[i] = int(cells[depth - 1]) if depth else -1
        moves[i] = int(cells[depth])
        player[i] = 1 + depth % 2
    return st, ply, last, moves, player


def _step(eng, args, static):
    st, ply, last, moves, player = [a.copy() for a in args]
    ended, winner, planes = eng.board_step_batch(st, ply, last, moves, player, static=static)
    return st, ply, last, ended, winner, planes


@pytest.mark.parametrize("n,k", [(8, 5), (15, 5)])
def test_static_board_step_equals_allocating_one(n, k):
    import alphazero_gomoku_amd as az
    eng = az.SelfPlayEngine(n, k, 1, 1, prior_mode=az.PRIOR_HASH)
    import torch
    _step(eng, _boards(n, k, 1024, 9), static=True)          # warm-up: whatever the runtime sets up on a first launch
    torch.cuda.synchronize()
    free_before = torch.cuda.mem_get_info()[0]               # hipMemGetInfo: the whole device, not torch's pool
    for count, seed in ((1000, 1), (37, 2)):                 # the second call is smaller than the first
        args = _boards(n, k, count, seed)
        got = _step(eng, args, static=True)
        assert torch.cuda.mem_get_info()[0] == free_before, "the static entry point changed the GPU's free memory"
        want = _step(eng, args, static=False)
        free_before = torch.cuda.mem_get_info()[0]           # the allocating entry point may have grown its buffers
        for a, b in zip(want, got):
            assert a.dtype == b.dtype and np.array_equal(a, b)
        assert want[3].any() or count < 100                      # some of the random boards end
    # an illegal move is reported, and the flag does not stick to the next call
    st, ply, last, moves, player = _boards(n, k, 5, 3, min_depth=1)     # every board holds at least one stone
    bad = moves.copy()
    bad[2] = last[2]                                                    # the cell of the stone just played: occupied
    with pytest.raises(az._lib.AzgError):
        eng.board_step_batch(st.copy(), ply.copy(), last.copy(), bad, player, static=True)
    assert torch.cuda.mem_get_info()[0] == free_before
    want = _step(eng, (st, ply, last, moves, player), static=False)
    got = _step(eng, (st, ply, last, moves, player), static=True)
    for a, b in zip(want, got):
        assert np.array_equal(a, b)
    # more boards than the fixed array holds: a clear error, nothing launched
    big = _boards(n, k, 1025, 4)
    with pytest.raises(az._lib.AzgError, match="at most 1024"):
        _step(eng, big, static=True)
    eng.close()
