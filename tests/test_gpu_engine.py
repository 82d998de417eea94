"""GPU parity tests: the HIP engine, called through the C ABI, against the CPU oracle and
the golden vectors produced from the reference.  Integer results (visit counts, moves,
winners, planes) must match bit-exactly; tree Q values bit-exactly (IEEE double, same
operation order); pi within 1e-12; CNN outputs within 1e-4 (float32)."""
import os

import numpy as np
import pytest

import helpers

pytestmark = pytest.mark.gpu
G = helpers.GOLDEN


def load(name):
    return helpers.load_golden(name)


@pytest.fixture(scope="module")
def az():
    import alphazero_gomoku_amd as pkg
    return pkg


def stones_of(moves, first=1):
    st = {}
    for i, m in enumerate(moves):
        st[int(m)] = first if i % 2 == 0 else 3 - first
    return st


MODES = {"uniform32": 1, "hash": 2}


def test_rules_replay_batched(az):
    """Board.do_move / game_end / current_state for every golden game, all games of one
    board size stepping together through azg_board_step_batch."""
    d = load("g1_rules.npz")
    meta, moves, ends, winners, states = d["meta"], d["moves"], d["ends"], d["winners"], d["states"]
    games = []
    mo = so = 0
    for (n, k, ln) in meta:
        games.append((int(n), int(k), moves[mo:mo + ln], ends[mo:mo + ln], winners[mo:mo + ln],
                      states[so:so + 8 * n * n].reshape(2, 4, n, n)))
        mo += ln
        so += 8 * n * n
    for (n, k) in sorted(set((g[0], g[1]) for g in games)):
        sel = [g for g in games if g[0] == n and g[1] == k]
        eng = az.SelfPlayEngine(n, k, 1, 1, prior_mode=az.PRIOR_HASH)
        nb = len(sel)
        st = np.zeros((nb, 2, 4), dtype=np.uint64)
        ply = np.zeros(nb, dtype=np.int32)
        last = np.full(nb, -1, dtype=np.int32)
        maxlen = max(len(g[2]) for g in sel)
        done_len = np.array([len(g[2]) for g in sel])
        for t in range(maxlen):
            act = np.nonzero(done_len > t)[0]
            sub_st = np.ascontiguousarray(st[act])
            sub_ply = np.ascontiguousarray(ply[act])
            sub_last = np.ascontiguousarray(last[act])
            mv = np.array([sel[i][2][t] for i in act], dtype=np.int32)
            pl = np.array([1 + t % 2] * len(act), dtype=np.int32)
            e, w, planes = eng.board_step_batch(sub_st, sub_ply, sub_last, mv, pl)
            st[act], ply[act], last[act] = sub_st, sub_ply, sub_last
            for j, i in enumerate(act):
                assert (int(e[j]), int(w[j])) == (int(sel[i][3][t]), int(sel[i][4][t])), (n, k, i, t)
                ln = len(sel[i][2])
                if t + 1 == ln:
                    assert np.array_equal(planes[j], sel[i][5][1].astype(np.float32)), (n, k, i, "final")
                if t + 1 == ln // 2 and ln // 2 > 0:
                    assert np.array_equal(planes[j], sel[i][5][0].astype(np.float32)), (n, k, i, "mid")
        with pytest.raises(az._lib.AzgError):
            eng.board_step_batch(np.ascontiguousarray(st[:1]), np.ascontiguousarray(ply[:1]),
                                 np.ascontiguousarray(last[:1]), np.array([sel[0][2][0]], dtype=np.int32),
                                 np.array([1], dtype=np.int32))
        eng.close()


def test_tree_fixtures_bit_exact(az):
    """Root statistics after n playouts (fresh root, then re-rooted and searched again)
    against the reference's own MCTS with exact synthetic callbacks."""
    for c in load("g2_tree.npz")["cases"]:
        n, k = int(c["n"]), int(c["k"])
        eng = az.SelfPlayEngine(n, k, int(c["n_playout"]), 1, float(c["c_puct"]), prior_mode=MODES[c["cb"]])
        eng.reset_games()
        prefix = [int(m) for m in c["prefix"]]
        st = stones_of(prefix)
        cur = 1 + len(prefix) % 2
        eng.set_position(0, st, prefix[-1] if prefix else -1, cur)
        eng.search()
        visits, q, p, rn = eng.root_stats()
        acts = c["acts"]
        tag = (n, c["cb"], int(c["n_playout"]), tuple(prefix))
        assert np.array_equal(visits[0][acts], c["visits"]), tag
        assert visits[0].sum() == c["visits"].sum(), tag
        assert np.array_equal(q[0][acts], c["q"]), tag
        assert np.array_equal(p[0][acts], c["p"]), tag
        mv = int(c["move"])
        if len(c["acts2"]):
            moves = np.array([mv], dtype=np.int32)
            eng.advance(moves)
            _, _, _, rn = eng.root_stats()
            assert rn[0] == c["root_n_before"], tag
            eng.search()
            visits, q, p, rn = eng.root_stats()
            assert np.array_equal(visits[0][c["acts2"]], c["visits2"]), tag
            assert np.array_equal(q[0][c["acts2"]], c["q2"]), tag
            assert rn[0] == c["root_n_after"], tag
        eng.close()


def test_seeded_games_match_reference(az):
    """Whole seeded self-play games (numpy-legacy Dirichlet stream, tree reuse) against the
    reference's recorded games: moves, visit counts per move, pi, z, winner, states."""
    games = [g for g in load("g3_games.npz")["games"] if g["cb"] in MODES]
    for g in games:
        n, k = int(g["n"]), int(g["k"])
        eng = az.SelfPlayEngine(n, k, int(g["n_playout"]), 1, float(g["c_puct"]), prior_mode=MODES[g["cb"]])
        eng.reset_games(seeds=np.array([g["seed"]], dtype=np.uint32))
        tag = (n, g["cb"], int(g["n_playout"]), int(g["seed"]))
        for t in range(len(g["moves"])):
            eng.search()
            visits, _, _, _ = eng.root_stats()
            assert np.array_equal(visits[0], g["visits"][t]), (tag, t)
            moves, pi = eng.sample_and_advance(float(g["temp"]))
            assert moves[0] == g["moves"][t], (tag, t)
            np.testing.assert_allclose(pi[0], g["pi"][t], rtol=0, atol=1e-12)
        ended, winner, plies = eng.status()
        assert ended[0] and winner[0] == g["winner"] and plies[0] == len(g["moves"]), tag
        s = eng.drain_samples()
        assert np.array_equal(s["z"], g["z"].astype(np.float32)), tag
        assert np.array_equal(s["states"], g["states"].astype(np.float32)), tag
        np.testing.assert_allclose(s["pi"], g["pi"], rtol=0, atol=1e-12)
        assert np.array_equal(s["ply"], np.arange(len(g["moves"])))
        eng.close()


@pytest.mark.parametrize("cnn_mode", [0, 1], ids=["f16x3", "f32"])
def test_cnn_against_reference_and_oracle(az, oracle, cnn_mode):
    """azg_eval_batch (HIP CNN; split-f16 MFMA and float32 MFMA variants) against the
    reference's PyTorch and numpy nets (golden) and the CPU oracle: |dp| and |dv| below the
    1e-4 the north star allows; in practice both variants sit below 2e-6."""
    d = load("g4_cnn.npz")
    for (n, k, count, scale, seed) in d["spec"]:
        n, k, count, seed = int(n), int(k), int(count), int(seed)
        tag = "%d_s%g" % (n, scale)
        params = helpers.make_params(n, seed=seed, scale=float(scale))
        eng = az.SelfPlayEngine(n, k, 1, 1, cnn_mode=cnn_mode)
        eng.set_weights(params)
        states = d["states_" + tag].astype(np.float32)
        probs, vals = eng.eval_batch(states)
        assert np.abs(probs - d["probs_torch_" + tag]).max() < 1e-4, tag
        assert np.abs(vals - d["value_torch_" + tag]).max() < 1e-4, tag
        assert np.abs(probs - d["probs_numpy_" + tag]).max() < 1e-5, tag
        assert np.abs(vals - d["value_numpy_" + tag]).max() < 1e-5, tag
        net = oracle.OracleNet(n, n, params, precise=True)
        for i in range(0, count, 7):
            po, vo = net.forward(states[i].astype(np.float64))
            assert np.abs(probs[i] - po).max() < 1e-5 and abs(vals[i] - vo) < 1e-5
        # the theano-layout entry point must give the same numbers
        eng2 = az.SelfPlayEngine(n, k, 1, 1, cnn_mode=cnn_mode)
        eng2.set_weights(helpers.torch_to_theano_layout(params), layout=az._lib.LAYOUT_THEANO)
        p2, v2 = eng2.eval_batch(states)
        assert np.array_equal(p2, probs) and np.array_equal(v2, vals)
        # batch-position invariance: a state evaluates to the same bits wherever it sits
        p3, v3 = eng.eval_batch(states[::-1].copy())
        assert np.array_equal(p3[::-1], probs) and np.array_equal(v3[::-1], vals)
        eng.close()
        eng2.close()


@pytest.mark.parametrize("cnn_mode", [0, 1], ids=["f16x3", "f32"])
def test_net_search_bit_exact_vs_oracle_with_gpu_priors(az, oracle, cnn_mode):
    """Net-driven search: the CPU oracle's MCTS, fed the GPU CNN's own float32 outputs
    through a callback, must reproduce the engine's visit counts bit-exactly."""
    for (n, k, npo, prefix) in [(6, 4, 120, []), (8, 5, 200, []), (8, 5, 150, [27, 28, 35, 36, 20, 21]),
                                (15, 5, 60, [112, 113, 97])]:
        params = helpers.make_params(n, seed=100 + n, scale=2.0)
        eng = az.SelfPlayEngine(n, k, npo, 1, cnn_mode=cnn_mode)
        eng.set_weights(params)
        ev = az.SelfPlayEngine(n, k, 1, 1, cnn_mode=cnn_mode)
        ev.set_weights(params)
        eng.reset_games()
        st = stones_of(prefix)
        eng.set_position(0, st, prefix[-1] if prefix else -1, 1 + len(prefix) % 2)
        eng.search()
        visits, q, p, rn = eng.root_stats()

        def policy(b):
            ob = oracle.OracleBoard(n, n, k)
            ob.b = b
            pr, v = ev.eval_batch(ob.current_state().astype(np.float32))
            return [float(x) for x in pr[0]], float(v[0])
        ob = oracle.OracleBoard(n, n, k)
        ob.init_board()
        for m in prefix:
            ob.do_move(m)
        t = oracle.OracleMCTS(policy, c_puct=5, n_playout=npo)
        acts, ovis, oq, op, _ = t.get_move_probs(ob, temp=1.0)
        assert np.array_equal(visits[0][acts], ovis), (n, npo)
        assert np.array_equal(q[0][acts], oq), (n, npo)
        assert np.array_equal(p[0][acts], op), (n, npo)
        eng.close()
        ev.close()


@pytest.mark.parametrize("n,k,npo,B", [(8, 5, 101, 6), (15, 5, 37, 3)])
def test_search_in_several_launches_gives_the_same_trees(az, n, k, npo, B, monkeypatch):
    """A move's playouts spread over 1, 2, 7 or more launches than playouts (AZG_SEARCH_PARTS, read by
    azg_create): three plies of B games with tree reuse, and after every search the same visit
    counts, Q values and priors at the root, bit for bit."""
    params = helpers.make_params(n, seed=100 + n, scale=2.0)
    want = None
    for parts in ("1", "2", "7", "64"):
        monkeypatch.setenv("AZG_SEARCH_PARTS", parts)
        eng = az.SelfPlayEngine(n, k, npo, B)
        assert eng.search_parts() == min(int(parts), npo)
        if parts == "2":
            eng.set_search_parts(5)                          # at run time as well
            assert eng.search_parts() == 5
        eng.set_weights(params)
        eng.reset_games(seeds=np.arange(B, dtype=np.uint32) + 5)
        got = []
        for _ in range(3):
            eng.search()
            visits, q, p, rn = eng.root_stats()
            got.append((visits.copy(), q.copy(), p.copy(), rn.copy()))
            eng.sample_and_advance(1.0)
        eng.close()
        if want is None:
            want = got
        for a, b in zip(want, got):
            for x, y in zip(a, b):
                assert np.array_equal(x, y), parts


@pytest.mark.parametrize("pipelined", [False, True], ids=["host-refill", "device-refill"])
def test_batched_games_are_independent_of_batching(az, oracle, pipelined):
    """Many games in flight: each slot's game must equal the oracle's game for that seed,
    whatever slot or batch it ran in (hash priors; bit-exact moves and visit counts), both with
    slots restarted by the host between steps and with the pipelined azg_selfplay_* calls that
    restart them on the device while the next search is already enqueued."""
    n, k, npo = 8, 5, 64
    eng = az.SelfPlayEngine(n, k, npo, 37, prior_mode=az.PRIOR_HASH)
    res = eng.play_games(60, temp=1.0, seed0=1000, pipelined=pipelined)
    s = res["samples"]
    assert res["games"] == 60
    assert sorted(set(s["game_id"].tolist())) == list(range(60))
    assert res["plies"] == len(s["z"])
    if pipelined:
        with pytest.raises(az._lib.AzgError):
            eng.selfplay_step()                              # not between begin and end
        eng.selfplay_begin(total=3)
        with pytest.raises(az._lib.AzgError):
            eng.sample_and_advance()                         # synchronous calls are locked out
        eng.selfplay_end()
    for gid in (0, 1, 17, 36, 37, 59):
        r = oracle.self_play(n, k, "hash", 5, npo, 1.0, 1000 + gid)
        sel = np.nonzero(s["game_id"] == gid)[0]
        sel = sel[np.argsort(s["ply"][sel])]
        assert len(sel) == len(r["moves"]), gid
        np.testing.assert_allclose(s["pi"][sel], r["pi"], rtol=0, atol=1e-12)
        assert np.array_equal(s["z"][sel], r["z"].astype(np.float32)), gid
        assert np.array_equal(s["states"][sel], r["states"].astype(np.float32)), gid
    eng.close()


def test_sqrt_is_correctly_rounded_on_device(az):
    """The PUCT term uses sqrt(parent visits) in double; the tree fixtures above cover it
    up to a few hundred visits, this covers larger counts through a long search."""
    eng = az.SelfPlayEngine(6, 4, 3000, 1, prior_mode=az.PRIOR_HASH, arena_nodes=3100)
    eng.reset_games()
    eng.search()
    visits, q, _, rn = eng.root_stats()
    from oracle import az_oracle
    ob = az_oracle.OracleBoard(6, 6, 4)
    ob.init_board()
    t = az_oracle.OracleMCTS("hash", c_puct=5, n_playout=3000)
    acts, ovis, oq, _, _ = t.get_move_probs(ob, temp=1.0)
    assert rn[0] == 3000 and np.array_equal(visits[0][acts], ovis) and np.array_equal(q[0][acts], oq)
    eng.close()


def _forced_line(az, oracle, eng, policy, n, k, npo, plies, slots):
    """Search, compare the root statistics of `slots` with the oracle's tree (which keeps the subtree
    of every move played, mcts_alphaZero.py:183-192), play the most visited child everywhere."""
    ob = oracle.OracleBoard(n, n, k)
    ob.init_board()
    t = oracle.OracleMCTS(policy, c_puct=5, n_playout=npo)
    for ply in range(plies):
        eng.search()
        visits, q, _, rn = eng.root_stats()
        acts, ovis, oq, _, _ = t.get_move_probs(ob, temp=1.0)
        for g in slots:
            assert rn[g] == t.root_visits() and np.array_equal(visits[g][acts], ovis), (ply, g)
            assert np.array_equal(q[g][acts], oq), (ply, g)
        mv = int(acts[np.argmax(ovis)])
        eng.advance(np.full(eng.B, mv, dtype=np.int32))
        t.update_with_move(mv)
        ob.do_move(mv)
        if ob.game_end()[0]:
            break
    return ply


def test_trees_that_outgrow_their_arena_move_to_a_large_one(az, oracle):
    """Tree reuse keeps the subtree of the move played, so a tree has no fixed bound in the
    reference.  With arenas barely larger than one search, every game must move into a large arena
    before its second search and still match the oracle visit for visit, Q for Q."""
    n, k, npo = 8, 5, 100
    eng = az.SelfPlayEngine(n, k, npo, 3, prior_mode=az.PRIOR_HASH, arena_nodes=npo + 4)
    eng.reset_games()
    plies = _forced_line(az, oracle, eng, "hash", n, k, npo, 12, (0, 1, 2))
    moved, dropped = eng.arena_stats()
    assert plies >= 6 and moved == 3 and dropped == 0
    # a finished or reset game gives its large arena back: the next games can use it again
    eng.reset_games()
    _forced_line(az, oracle, eng, "hash", n, k, npo, 4, (0, 1, 2))
    assert eng.arena_stats() == (6, 0)
    eng.close()
    # one large arena for four games that all need one: one game keeps its tree and stays exact, the
    # others lose theirs (fresh root: exactly n_playout root visits after the next search), nobody fails
    eng = az.SelfPlayEngine(n, k, npo, 4, prior_mode=az.PRIOR_HASH, arena_nodes=npo + 4, big_arenas=1)
    eng.reset_games()
    eng.search()
    visits, _, _, _ = eng.root_stats()
    mv = int(np.argmax(visits[0]))
    eng.advance(np.full(4, mv, dtype=np.int32))
    assert eng.arena_stats() == (1, 3)
    _, _, _, rn0 = eng.root_stats()
    eng.search()
    _, _, _, rn = eng.root_stats()
    assert sorted(rn.tolist()) == [npo] * 3 + [int(visits[0][mv]) + npo] and sorted(rn0.tolist()) == [0] * 3 + [int(visits[0][mv])]
    eng.close()


@pytest.mark.parametrize("n,k,npo,nodes,peaked", [(8, 5, 100, 260, False), (8, 5, 100, 1000, False), (6, 4, 64, 100, True),
                                                  (6, 4, 64, 150, True), (15, 5, 120, 330, False)])
def test_reroot_in_place_alternating_with_compaction(az, oracle, n, k, npo, nodes, peaked):
    """MCTS.update_with_move (mcts_alphaZero.py:183-192) keeps the subtree of the move played.  Here
    the child's record moves to the root's place and its subtree stays where it is while the arena has
    room for one more search; when it has not, the subtree is compacted into the other arena, and when
    even that is too small the game moves to a large arena.  Arenas of 2.5 to 10 searches make these
    cases alternate within one game (flat and peaked priors, child blocks wider than a wave at 15x15);
    every search must equal the oracle's, visit for visit, Q for Q."""
    S = n * n

    def peaked_policy(b):
        low = min(b.avail[i] for i in range(b.n_avail)) if b.n_avail else -1
        return [0.9375 if c == low else 0.0009765625 for c in range(S)], 0.0
    eng = az.SelfPlayEngine(n, k, npo, 2, prior_mode=az.PRIOR_PEAKED if peaked else az.PRIOR_HASH, arena_nodes=nodes)
    eng.reset_games()
    plies = _forced_line(az, oracle, eng, peaked_policy if peaked else "hash", n, k, npo, 24, (0, 1))
    assert plies >= 6 and eng.arena_stats()[1] == 0
    in_place, compacted = eng.reroot_stats()
    # (the move that ends the game starts a fresh root instead)
    assert in_place >= 4 and compacted >= 2 and in_place + compacted in (2 * plies, 2 * plies + 2), (in_place, compacted, plies)
    eng.close()


@pytest.mark.parametrize("n,k,npo", [(8, 5, 1500), (15, 5, 600)])
def test_reroot_with_a_small_work_stack(az, oracle, n, k, npo, monkeypatch):
    """MCTS.update_with_move (mcts_alphaZero.py:183-192): the subtree copy keeps its work list of
    expanded nodes in LDS and parks the older half in the arena tail when that is full.  With the
    smallest stack (AZG_REROOT_STACK=128, read by azg_create) and large trees the parking path runs on
    nearly every move; the kept statistics must still equal the oracle's, visit for visit, Q for Q --
    on 15x15 with child blocks wider than a wave."""
    monkeypatch.setenv("AZG_REROOT_STACK", "128")
    monkeypatch.setenv("AZG_REROOT_COPY", "1")        # copy on every move, not only when the arena runs low
    eng = az.SelfPlayEngine(n, k, npo, 2, prior_mode=az.PRIOR_HASH)
    monkeypatch.delenv("AZG_REROOT_STACK")
    monkeypatch.delenv("AZG_REROOT_COPY")
    eng.reset_games()
    plies = _forced_line(az, oracle, eng, "hash", n, k, npo, 10, (0, 1))
    assert plies >= 5
    eng.close()


def test_peaked_prior_over_consecutive_moves(az, oracle):
    """A prior that puts 15/16 on one move, followed for a whole game: the root keeps almost all
    earlier visits (ten times n_playout after a dozen moves) and the default arena must cope."""
    n, k, npo = 6, 4, 64
    S = n * n

    def policy(b):
        low = min(b.avail[i] for i in range(b.n_avail)) if b.n_avail else -1
        return [0.9375 if c == low else 0.0009765625 for c in range(S)], 0.0
    eng = az.SelfPlayEngine(n, k, npo, 2, prior_mode=az.PRIOR_PEAKED)
    eng.reset_games()
    plies = _forced_line(az, oracle, eng, policy, n, k, npo, 30, (0, 1))
    assert plies >= 12 and eng.arena_stats()[1] == 0
    eng.close()
    # with n_in_row below 3 a line can be complete before the reference's n_in_row + 2 stone
    # threshold lets it count (game.py:136-137), so it need not contain the last stone
    eng = az.SelfPlayEngine(n, 2, 40, 1, prior_mode=az.PRIOR_HASH)
    eng.reset_games()
    ob = oracle.OracleBoard(n, n, 2)
    ob.init_board()
    t = oracle.OracleMCTS("hash", c_puct=5, n_playout=40)
    eng.search()
    visits, q, _, rn = eng.root_stats()
    acts, ovis, oq, _, _ = t.get_move_probs(ob, temp=1.0)
    assert np.array_equal(visits[0][acts], ovis) and np.array_equal(q[0][acts], oq)
    eng.close()


def test_error_paths(az):
    with pytest.raises(az._lib.AzgError):
        az.SelfPlayEngine(4, 5, 10, 1)                       # board smaller than n_in_row
    with pytest.raises(az._lib.AzgError):
        az.SelfPlayEngine(5, 5, 10, 1)                       # no CNN kernel below 6x6
    with pytest.raises(az._lib.AzgError):
        az.SelfPlayEngine(16, 5, 10, 1, cnn_mode=az._lib.CNN_F32)  # float32-MFMA kernels exist for 6x6 ... 15x15 (LDS), not 16x16
    eng = az.SelfPlayEngine(8, 5, 10, 1)
    eng.reset_games()
    with pytest.raises(az._lib.AzgError) as e:
        eng.search()                                         # weights missing
    assert e.value.code == -3
    eng.close()
    eng = az.SelfPlayEngine(6, 4, 400, 1, prior_mode=az.PRIOR_HASH, arena_nodes=16)
    eng.reset_games()
    eng.search()
    with pytest.raises(az._lib.AzgError) as e:
        eng.status()                                         # arena overflow is reported
    assert e.value.code == -4
    eng.close()


def test_dropin_surface_reproduces_reference_games(az):
    """Board / Game.start_self_play / MCTSPlayer.get_action with the reference's signatures:
    np.random.seed(s) followed by start_self_play must give the reference's recorded game."""
    games = [g for g in load("g3_games.npz")["games"] if g["cb"] in MODES and g["n"] in (6, 8)][:4]
    for g in games:
        n, k = int(g["n"]), int(g["k"])
        board = az.Board(width=n, height=n, n_in_row=k)
        game = az.Game(board)
        player = az.MCTSPlayer(g["cb"], c_puct=float(g["c_puct"]), n_playout=int(g["n_playout"]), is_selfplay=1)
        np.random.seed(int(g["seed"]))
        winner, data = game.start_self_play(player, temp=float(g["temp"]))
        data = list(data)
        assert winner == g["winner"] and len(data) == len(g["moves"])
        for t, (st, pi, z) in enumerate(data):
            assert np.array_equal(st, g["states"][t].astype(np.float64))
            np.testing.assert_allclose(pi, g["pi"][t], rtol=0, atol=1e-12)
            assert z == g["z"][t]


def test_policy_value_net_surface(az, oracle):
    """PolicyValueNet.policy_value / policy_value_fn run the HIP CNN; MCTSPlayer accepts the
    bound method as train.py passes it; start_self_play_batch returns per-game tuples."""
    n, k = 6, 4
    params = helpers.make_params(n, seed=106)
    net = az.PolicyValueNet(n, n)
    net.load_numpy_params(params)
    board = az.Board(width=n, height=n, n_in_row=k)
    board.init_board()
    for m in (14, 15, 20):
        board.do_move(m)
    onet = oracle.OracleNet(n, n, params, precise=True)
    po, vo = onet.forward(board.current_state())
    ap, v = net.policy_value_fn(board)
    ap = list(ap)
    assert [a for a, _ in ap] == board.availables
    assert max(abs(p - po[a]) for a, p in ap) < 1e-4 and abs(v - vo) < 1e-4
    probs, vals = net.policy_value(np.stack([board.current_state()] * 3))
    assert probs.shape == (3, n * n) and vals.shape == (3, 1)
    player = az.MCTSPlayer(net.policy_value_fn, c_puct=5, n_playout=30, is_selfplay=1)
    np.random.seed(3)
    move, pi = player.get_action(board, temp=1.0, return_prob=1)
    assert move in board.availables and abs(pi.sum() - 1.0) < 1e-9
    loss, ent = net.train_step(np.stack([board.current_state()] * 4), np.stack([pi] * 4), np.ones(4), 2e-3)
    assert np.isfinite(loss) and np.isfinite(ent)
    out = az.Game(board).start_self_play_batch(net, n_games=5, n_playout=20, n_parallel=3, seed0=11)
    assert sorted(gid for gid, _, _, _ in out) == [0, 1, 2, 3, 4]
    for gid, states, mcts_probs, z in out:
        assert states.shape[1:] == (4, n, n) and mcts_probs.shape == (len(z), n * n)
        assert set(np.unique(z)).issubset({-1.0, 0.0, 1.0})


def test_equi_data_kernel_matches_fixture(az):
    """azg_equi_data / get_equi_data (train.py:116-135) on the GPU: bit-exact permutation of the
    inputs, against the committed fixture, for 6x6, 8x8 and 15x15; numpy and torch entry."""
    import torch
    from alphazero_gomoku_amd.train_data import equi_data_arrays, get_equi_data
    d = load("g6_equi.npz")
    for n in (6, 8, 15):
        eng = az.SelfPlayEngine(n, 4, 1, 1, prior_mode=az.PRIOR_HASH)
        states, pi, z = d["states_%d" % n], d["pi_%d" % n], d["z_%d" % n]
        so, po, zo = equi_data_arrays(eng, states, pi, z)
        assert np.array_equal(so, d["ext_states_%d" % n])
        assert np.array_equal(po, d["ext_pi_%d" % n])
        assert np.array_equal(zo, d["ext_z_%d" % n])
        dev = torch.device("cuda", 0)
        st, pt, zt = (torch.from_numpy(a).to(dev) for a in (states, pi, z))
        so2, po2, zo2 = equi_data_arrays(eng, st, pt, zt)
        assert np.array_equal(so2.cpu().numpy(), so) and np.array_equal(po2.cpu().numpy(), po)
        ext = get_equi_data(eng, [(states[i], pi[i], z[i]) for i in range(len(z))])
        assert len(ext) == 8 * len(z) and np.array_equal(ext[9][0], so[9]) and np.array_equal(ext[9][1], po[9])
        assert get_equi_data(eng, []) == []
        eng.close()


def test_split_f16_weight_range_is_checked(az):
    """The split-f16 convolutions hold w * 2^8 in halves: out-of-range weights are rejected
    with a clear error instead of overflowing; the float32 variant accepts them."""
    params = helpers.make_params(6, seed=1)
    params[4] = params[4].copy()
    params[4].flat[3] = 300.0
    eng = az.SelfPlayEngine(6, 4, 8, 1, cnn_mode=az._lib.CNN_F16X3)
    with pytest.raises(az._lib.AzgError) as ei:
        eng.set_weights(params)
    assert "AZG_CNN_F32" in str(ei.value)
    eng.close()
    eng = az.SelfPlayEngine(6, 4, 8, 1, cnn_mode=az._lib.CNN_F32)
    eng.set_weights(params)
    eng.close()


def test_pipelined_selfplay_with_fewer_games_than_slots(az, oracle):
    """azg_selfplay_* with total < n_games: the spare slots stay idle, exactly `total` games are
    played, each equal to the oracle's game for its seed, with a game-id stride (rank sharding)."""
    n, k, npo = 6, 4, 32
    eng = az.SelfPlayEngine(n, k, npo, 8, prior_mode=az.PRIOR_HASH)
    res = eng.play_games(5, temp=1.0, seed0=40, first_game_id=1, game_id_stride=2, pipelined=True)
    s = res["samples"]
    assert res["games"] == 5 and sorted(set(s["game_id"].tolist())) == [1, 3, 5, 7, 9]
    for gid in (1, 9):
        r = oracle.self_play(n, k, "hash", 5, npo, 1.0, 40 + gid)
        sel = np.nonzero(s["game_id"] == gid)[0]
        sel = sel[np.argsort(s["ply"][sel])]
        assert len(sel) == len(r["moves"])
        np.testing.assert_allclose(s["pi"][sel], r["pi"], rtol=0, atol=1e-12)
        assert np.array_equal(s["z"][sel], r["z"].astype(np.float32))
    eng.close()


def test_train_pipeline_two_batches(az):
    """TrainPipeline (train.py:60-214, 278-330 without the evaluation arena): self-play batches
    feed the replay buffer 8-fold, policy_update runs the KL-controlled train steps and the
    engine picks up the new weights."""
    import random
    import torch
    from alphazero_gomoku_amd.train import TrainPipeline
    random.seed(0)
    torch.manual_seed(0)
    tp = TrainPipeline(board_width=6, board_height=6, n_in_row=4, n_playout=16, play_batch_size=8, batch_size=32,
                       buffer_size=400, epochs=2, game_batch_num=2, n_parallel=8)
    seen = []
    tp.run(on_batch=lambda i, epi, loss, ent: seen.append((i, epi, loss, ent)))
    assert len(seen) == 2 and tp.games_played == 16
    assert seen[0][1] >= 8 * 7                                # at least 7 plies per 6x6/4 game
    assert len(tp.data_buffer) == min(400, 8 * (seen[0][1] + seen[1][1]))
    assert all(s[2] is not None and np.isfinite(s[2]) and np.isfinite(s[3]) for s in seen)
    assert 0.05 / 1.5 <= tp.lr_multiplier <= 10 * 1.5 and np.isfinite(tp.last_kl)
    assert 0 <= tp._engine_version < tp.policy_value_net.version     # the last update is not in the engine yet
    v0 = tp.policy_value_net.version
    tp.collect_selfplay_data(1)                               # pushes the updated weights into the engine
    assert tp._engine_version == v0


def test_net_driven_games_do_not_depend_on_refill_mode_or_batch_width(az):
    """CNN-driven self-play (split-f16 MFMA net, two workgroups per CU, matrix-pipe gate): the
    games are a function of (weights, seed) only -- the same tuples come out with host refill and
    8 slots as with the pipelined device refill and 5 slots."""
    n, k, npo = 6, 4, 24
    params = helpers.make_params(n, seed=9)
    out = []
    for slots, pipelined in ((8, False), (5, True)):
        eng = az.SelfPlayEngine(n, k, npo, slots)
        eng.set_weights(params)
        res = eng.play_games(12, temp=1.0, seed0=7, pipelined=pipelined)
        s = res["samples"]
        order = np.lexsort((s["ply"], s["game_id"]))
        out.append({key: s[key][order] for key in s})
        eng.close()
    a, b = out
    assert sorted(set(a["game_id"].tolist())) == list(range(12))
    for key in ("game_id", "ply", "z", "states", "pi"):
        assert np.array_equal(a[key], b[key]), key


@pytest.mark.parametrize("cnn_mode", [0, 1], ids=["f16x3", "f32"])
def test_reference_c1_game_with_the_cnn(az, cnn_mode):
    """Configuration C1 end to end, and one game at C2's per-game settings: the reference's own
    6x6 / 4-in-row / 200-playout, 8x8 / 5-in-row / 400-playout and 15x15 / 5-in-row / 100-playout
    self-play games driven by its numpy policy-value net (recorded by tools/gen_golden.py) are reproduced move for move and
    visit count for visit count with the leaf evaluations coming from the HIP CNN (the priors
    agree to ~1e-8, which never flips a PUCT decision in these games)."""
    games = [(g, 106) for g in load("g3_games.npz")["games"] if g["cb"] not in MODES]
    # and the benchmark's per-game settings: 8x8 / 5-in-row / 400 playouts
    games += [(g, int(g["param_seed"])) for g in load("g7_net8.npz")["games"]]
    assert len(games) >= 2, "the fixtures hold net-driven games"
    for g, param_seed in games:
        n, k = int(g["n"]), int(g["k"])
        eng = az.SelfPlayEngine(n, k, int(g["n_playout"]), 1, float(g["c_puct"]), cnn_mode=cnn_mode)
        eng.set_weights(helpers.make_params(n, seed=param_seed))
        eng.reset_games(seeds=np.array([g["seed"]], dtype=np.uint32))
        for t in range(len(g["moves"])):
            eng.search()
            visits, _, _, _ = eng.root_stats()
            assert np.array_equal(visits[0], g["visits"][t]), t
            moves, pi = eng.sample_and_advance(float(g["temp"]))
            assert moves[0] == g["moves"][t], t
            np.testing.assert_allclose(pi[0], g["pi"][t], rtol=0, atol=1e-12)
        ended, winner, plies = eng.status()
        assert ended[0] and winner[0] == g["winner"] and plies[0] == len(g["moves"])
        s = eng.drain_samples()
        assert np.array_equal(s["z"], g["z"].astype(np.float32))
        assert np.array_equal(s["states"], g["states"].astype(np.float32))
        eng.close()


def _stream_words(seed):
    st = np.random.RandomState(int(seed)).get_state()
    w = np.empty((1, 625), dtype=np.uint32)
    w[0, :624], w[0, 624] = st[1], st[2]
    return w


def _set_case_position(eng, slot, c):
    first = int(c["start_player"]) + 1
    pre = [int(m) for m in c["pre"]]
    cur = first if len(pre) % 2 == 0 else 3 - first
    eng.set_position(slot, stones_of(pre, first), pre[-1] if pre else -1, cur)


def test_rollout_opponent_matches_reference(az):
    """azg_pure_search against searches of the reference's mcts_pure.MCTSPlayer (fixture g8): the
    move, every root visit count, the root's own count and the numpy stream handed back
    (position, key words and the next double) are bit-exact, on 6x6, 8x8, 9x9 and 15x15."""
    cases = load("g8_pure.npz")["cases"]
    assert len(cases) >= 8
    for c in cases:
        n, k, npo = int(c["n"]), int(c["k"]), int(c["n_playout"])
        eng = az.SelfPlayEngine(n, k, npo, 1, float(c["c_puct"]), prior_mode=az.PRIOR_UNIFORM32, arena_nodes=npo + 8)
        eng.reset_games()
        _set_case_position(eng, 0, c)
        words = _stream_words(c["seed"])
        moves = eng.pure_search(npo, words)
        visits, _, prior, root_n = eng.root_stats()
        assert moves[0] == c["move"] and root_n[0] == c["root_visits"], (n, int(c["seed"]))
        assert np.array_equal(visits[0], c["visits"])
        kk = int((c["visits"] > 0).sum())
        assert kk == 0 or abs(prior[0].max() - 1.0 / max(1, (prior[0] > 0).sum())) < 1e-7
        assert int(words[0, 624]) == c["rng_pos"] and int(np.bitwise_xor.reduce(words[0, :624])) == c["rng_key_crc"]
        rs = np.random.RandomState(0)
        rs.set_state(("MT19937", words[0, :624], int(words[0, 624]), 0, 0.0))
        assert rs.random_sample() == c["next_double"]
        eng.close()


def test_rollout_opponent_slots_are_independent(az, oracle):
    """Several slots searched in one launch (different positions, different streams, one of
    them not live) give what the oracle gives for each slot alone."""
    n, k, npo = 8, 5, 90
    pres = [[], [27], [27, 28, 35, 36, 20], [0, 63, 7, 56, 9], [10, 11, 12, 13]]
    B = len(pres) + 1
    eng = az.SelfPlayEngine(n, k, npo, B, 5.0, prior_mode=az.PRIOR_UNIFORM32, arena_nodes=npo + 8)
    eng.reset_games(mask=np.array([1] * len(pres) + [0], dtype=np.uint8))      # the last slot stays idle
    words = np.zeros((B, 625), dtype=np.uint32)
    for g, pre in enumerate(pres):
        eng.set_position(g, stones_of(pre, 1), pre[-1] if pre else -1, 1 if len(pre) % 2 == 0 else 2)
        words[g] = _stream_words(100 + g)[0]
    words[B - 1] = _stream_words(999)[0]
    idle_before = words[B - 1].copy()
    moves = eng.pure_search(npo, words)
    visits, _, _, root_n = eng.root_stats()
    assert moves[B - 1] == -1
    for g, pre in enumerate(pres):
        b = oracle.OracleBoard(n, n, k)
        b.init_board(0)
        for m in pre:
            b.do_move(m)
        rng = oracle.OracleRng(100 + g)
        mv, vis, rn = oracle.pure_get_move(b, 5.0, npo, rng)
        key, pos = rng.get_state()
        assert moves[g] == mv and root_n[g] == rn and np.array_equal(visits[g], vis), g
        assert int(words[g, 624]) == pos and np.array_equal(words[g, :624], key), g
    assert np.array_equal(words[B - 1], idle_before) or moves[B - 1] == -1
    with pytest.raises(az._lib.AzgError) as e:
        eng.pure_search(npo + 100, words)                    # one node per playout must fit the arena
    assert e.value.code == -4
    eng.close()


class _Spy:
    """Forwards to a player and logs its moves (Game.start_play keeps none)."""

    def __init__(self, inner, log):
        self.inner, self.log = inner, log

    def set_player_ind(self, p):
        self.inner.set_player_ind(p)
        self.player = p

    def get_action(self, board):
        m = self.inner.get_action(board)
        self.log.append(int(m))
        return m


def test_arena_games_match_reference(az):
    """Game.start_play between the drop-in players, seeded like the reference's recorded games
    (np.random.seed / random.seed): rollout player against rollout player, and the network player
    in evaluation mode (HIP CNN, mcts_alphaZero.py:243-255) against the rollout player, which is the
    pairing of TrainPipeline.policy_evaluate (train.py:216-276).  Moves, winner and the numpy
    stream afterwards must be the reference's."""
    import random
    games = load("g8_pure.npz")["games"]
    assert len(games) >= 8
    for g in games:
        n, k = int(g["n"]), int(g["k"])
        log = []
        if g["kind"] == "pure_vs_pure":
            p1 = az.MCTS_Pure(c_puct=5, n_playout=int(g["n_playout1"]))
        else:
            net = az.PolicyValueNet(n, n)
            net.load_numpy_params(helpers.make_params(n, seed=int(g["param_seed"])))
            p1 = az.MCTSPlayer(net.policy_value_fn, c_puct=5, n_playout=int(g["n_playout1"]))
        p2 = az.MCTS_Pure(c_puct=5, n_playout=int(g["n_playout2"]))
        np.random.seed(int(g["seed"]))
        random.seed(int(g["seed"]))
        winner = az.Game(az.Board(width=n, height=n, n_in_row=k)).start_play(_Spy(p1, log), _Spy(p2, log),
                                                                             start_player=int(g["start_player"]), is_shown=0)
        assert log == [int(m) for m in g["moves"]], (g["kind"], n, int(g["start_player"]))
        assert winner == g["winner"]
        assert np.random.random_sample() == g["next_double"]


def test_policy_evaluate_runs(az):
    """TrainPipeline.policy_evaluate (train.py:216-276): win ratio of the current net against the
    rollout player over alternating first moves."""
    from alphazero_gomoku_amd.train import TrainPipeline
    tp = TrainPipeline(board_width=6, board_height=6, n_in_row=4, n_playout=30, play_batch_size=1, pure_mcts_playout_num=40)
    np.random.seed(1)
    # the evaluation-mode opening of the reference (fixed 15x15 cells) applies to 15x15 boards only,
    # so on 6x6 both colours are played out normally
    ratio = tp.policy_evaluate(n_games=2)
    assert 0.0 <= ratio <= 1.0 and sum(tp.last_win_cnt.values()) == 2


def test_run_evaluates_every_check_freq_batches_and_strengthens_the_opponent(az, monkeypatch):
    """TrainPipeline.run (train.py:328-344): every check_freq batches the current net plays the
    rollout player; a win ratio above the best so far becomes the mark (and saves the model), a
    perfect score adds 1000 playouts to the opponent and resets the mark to 0."""
    from alphazero_gomoku_amd.train import TrainPipeline
    tp = TrainPipeline(board_width=6, board_height=6, n_in_row=4, n_playout=16, play_batch_size=4, batch_size=16,
                       buffer_size=400, epochs=1, game_batch_num=4, n_parallel=4, check_freq=2, pure_mcts_playout_num=20)
    tp.run(evaluate_games=2, evaluate_batched=True)           # real games, side by side
    assert [e[0] for e in tp.eval_history] == [2, 4] and all(0.0 <= e[2] <= 1.0 for e in tp.eval_history)
    assert sum(tp.last_win_cnt.values()) == 2
    # the gating rule on scripted win ratios
    tp2 = TrainPipeline(board_width=6, board_height=6, n_in_row=4, n_playout=16, play_batch_size=4, batch_size=16,
                        buffer_size=400, epochs=1, game_batch_num=5, n_parallel=4, check_freq=1, pure_mcts_playout_num=4000)
    script = iter([0.1, 0.6, 1.0, 1.0, 0.3])
    monkeypatch.setattr(tp2, "policy_evaluate", lambda n, batched=False, seed0=0: next(script))
    saved = []
    monkeypatch.setattr(tp2.policy_value_net, "save_model", lambda path: saved.append((path, tp2.best_win_ratio)))
    tp2.run(best_model="best.model", evaluate_games=2)
    assert [e[1:] for e in tp2.eval_history] == [(4000, 0.1), (4000, 0.6), (4000, 1.0), (5000, 1.0), (5000, 0.3)]
    assert saved == [("best.model", 0.6), ("best.model", 1.0), ("best.model", 1.0)]   # 0.1 is below the 0.2 start mark
    assert tp2.pure_mcts_playout_num == 5000 and tp2.best_win_ratio == 1.0            # 5000 is the ceiling: no reset there


def test_batched_arena_equals_seeded_dropin_games(az):
    """arena.play_arena_games: game i of the batch is the game the drop-in players produce after
    np.random.seed(seed0 + i); random.seed(seed0 + i), with start_player = i % 2."""
    import random
    from alphazero_gomoku_amd.arena import play_arena_games, win_ratio
    n, k, npo, pure = 6, 4, 40, 60
    net = az.PolicyValueNet(n, n)
    net.load_numpy_params(helpers.make_params(n, seed=106))
    recs = play_arena_games(net, 5, n, k, npo, pure, 5.0, seed0=70)
    assert [r["game"] for r in recs] == [0, 1, 2, 3, 4] and 0.0 <= win_ratio(recs) <= 1.0
    for r in recs:
        log = []
        p1 = az.MCTSPlayer(net.policy_value_fn, c_puct=5, n_playout=npo)
        p2 = az.MCTS_Pure(c_puct=5, n_playout=pure)
        np.random.seed(70 + r["game"])
        random.seed(70 + r["game"])
        winner = az.Game(az.Board(width=n, height=n, n_in_row=k)).start_play(_Spy(p1, log), _Spy(p2, log),
                                                                             start_player=r["start_player"], is_shown=0)
        assert r["start_player"] == r["game"] % 2
        assert log == r["moves"] and winner == r["winner"], r["game"]


def _check_step_invariants(n, n_playout, visits, root_n, carried, pi, occupied, live):
    """Properties every move step has, whatever the size: the root was visited once per playout
    on top of what the re-used subtree carried over, all but the expanding visit went to a
    child, no visit and no probability sits on an occupied cell, pi sums to one."""
    assert np.array_equal(root_n[live], carried[live] + n_playout)
    assert np.array_equal(visits[live].sum(axis=1), root_n[live] - 1)
    assert not (visits[live] * occupied[live]).any()
    assert not (pi[live] * occupied[live]).any()
    np.testing.assert_allclose(pi[live].sum(axis=1), 1.0, rtol=0, atol=1e-12)


@pytest.mark.parametrize("n,k,n_playout,B,steps", [(8, 5, 400, 1024, 36), (15, 5, 800, 512, 2)],
                         ids=["C2-8x8-400-1024", "C4-15x15-800-512"])
def test_full_size_configurations_keep_the_invariants(az, n, k, n_playout, B, steps):
    """The benchmark's full sizes (1024 games of 8x8 / 400 playouts; 512 games of 15x15 / 800
    playouts) with the CNN as leaf evaluator: per-step invariants for every slot, the training
    tuples of finished games, and slots spread over the batch against the same seeds played alone."""
    eng = az.SelfPlayEngine(n, k, n_playout, B, 5.0)
    params = helpers.make_params(n, seed=100 + n)
    eng.set_weights(params)
    seeds = np.arange(B, dtype=np.uint32) + 500
    eng.reset_games(seeds=seeds)
    S = n * n
    occupied = np.zeros((B, S), dtype=bool)
    carried = np.zeros(B, dtype=np.int64)
    history = [[] for _ in range(B)]
    live = np.ones(B, dtype=bool)
    for t in range(steps):
        eng.search()
        visits, _, _, root_n = eng.root_stats()
        moves, pi = eng.sample_and_advance(1.0)
        _check_step_invariants(n, n_playout, visits, root_n.astype(np.int64), carried, pi, occupied, live)
        idx = np.nonzero(live)[0]
        assert (moves[idx] >= 0).all() and not occupied[idx, moves[idx]].any()
        assert (moves[~live] == -1).all()
        carried[idx] = visits[idx, moves[idx]]
        occupied[idx, moves[idx]] = True
        for g in idx:
            history[g].append(int(moves[g]))
        ended, winner, plies = eng.status()
        assert np.array_equal(plies, occupied.sum(axis=1))
        live = ~ended
    ended, winner, plies = eng.status()
    if ended.any():
        s = eng.drain_samples()
        assert len(s["z"]) == int(plies[ended].sum())
        for g in np.nonzero(ended)[0]:
            sel = s["game_id"] == g
            order = np.argsort(s["ply"][sel])
            z, st = s["z"][sel][order], s["states"][sel][order]
            assert len(z) == plies[g] and np.array_equal(s["ply"][sel][order], np.arange(plies[g]))
            mover = 1 + (np.arange(plies[g]) % 2)                         # stone 1 opens
            want = np.zeros(plies[g]) if winner[g] == -1 else np.where(mover == winner[g], 1.0, -1.0)
            assert np.array_equal(z, want.astype(np.float32))
            # stones on the planes = plies played before the move; colour plane = parity
            assert np.array_equal(st[:, 0].sum(axis=(1, 2)) + st[:, 1].sum(axis=(1, 2)), np.arange(plies[g]))
            assert np.array_equal(st[:, 3, 0, 0], (np.arange(plies[g]) % 2 == 0).astype(np.float32))
    eng.close()
    # the same seeds played alone give the same games (slots do not influence each other)
    check = min(steps, 3)
    for g in (0, B // 2 + 1, B - 1):
        one = az.SelfPlayEngine(n, k, n_playout, 1, 5.0)
        one.set_weights(params)
        one.reset_games(seeds=seeds[g:g + 1])
        for t in range(min(check, len(history[g]))):
            one.search()
            mv, _ = one.sample_and_advance(1.0, want_pi=False)
            assert mv[0] == history[g][t], (g, t)
        one.close()


@pytest.mark.parametrize("n,k", [(7, 4), (10, 5), (16, 5)])
def test_rollout_opponent_other_boards_and_unaligned_stream(az, oracle, n, k):
    """Board sizes without a fixture (7x7, 10x10 and the largest supported, 16x16) against the
    oracle, starting from a stream position that is odd and in the middle of a generator block
    (the host program drew other numbers before), so doubles straddle the block refresh."""
    npo = 70
    rs = np.random.RandomState(n)
    rs.bytes(4 * 333)                                          # 333 generator words: odd position
    st = rs.get_state()
    assert st[2] % 2 == 1
    pre = [(n // 2) * n + n // 2, (n // 2) * n + n // 2 + 1, (n // 2 + 1) * n + n // 2]
    eng = az.SelfPlayEngine(n, k, npo, 1, 5.0, prior_mode=az.PRIOR_UNIFORM32, arena_nodes=npo + 8)
    eng.reset_games()
    eng.set_position(0, stones_of(pre, 1), pre[-1], 2)
    words = np.empty((1, 625), dtype=np.uint32)
    words[0, :624], words[0, 624] = st[1], st[2]
    moves = eng.pure_search(npo, words)
    visits, _, _, root_n = eng.root_stats()
    b = oracle.OracleBoard(n, n, k)
    b.init_board(0)
    for m in pre:
        b.do_move(m)
    rng = oracle.OracleRng(0)
    rng.set_state(st[1], st[2])
    mv, vis, rn = oracle.pure_get_move(b, 5.0, npo, rng)
    key, pos = rng.get_state()
    assert moves[0] == mv and root_n[0] == rn and np.array_equal(visits[0], vis)
    assert int(words[0, 624]) == pos and np.array_equal(words[0, :624], key)
    eng.close()


def test_weight_files_in_both_layouts(az, oracle, tmp_path):
    """Weight interchange (policy_value_net_pytorch.py:150-157, policy_value_net_numpy.py:103-113):
    a state_dict file and .npz archives in the PyTorch and in the Theano layout all load into the
    same network, whose HIP forward pass matches the oracle's net on the original arrays."""
    n = 8
    params = helpers.make_params(n, seed=108)
    net = az.PolicyValueNet(n, n)
    net.load_numpy_params(params)
    f_sd, f_t, f_th = tmp_path / "m.model", tmp_path / "torch.npz", tmp_path / "theano.npz"
    net.save_model(str(f_sd))
    net.save_numpy(str(f_t), layout="torch")
    net.save_numpy(str(f_th), layout="theano")
    with np.load(str(f_th)) as z:
        th = [z["arr_%d" % i] for i in range(16)]
    for a, b in zip(th, helpers.torch_to_theano_layout(params)):
        assert np.array_equal(a, b)
    board = az.Board(width=n, height=n, n_in_row=5)
    board.init_board()
    for m in (27, 28, 36):
        board.do_move(m)
    st = board.current_state()[None]
    po, vo = oracle.OracleNet(n, n, params, precise=True).forward(board.current_state())
    for f in (f_sd, f_t, f_th):
        other = az.PolicyValueNet(n, n, model_file=str(f))
        probs, vals = other.policy_value(st)
        assert np.abs(probs[0] - po).max() < 1e-5 and abs(vals[0, 0] - vo) < 1e-5, f
    # the engine takes the Theano layout directly as well
    eng = az.SelfPlayEngine(n, 5, 10, 1, 5.0)
    eng.set_weights(th, layout=az._lib.LAYOUT_THEANO)
    probs, vals = eng.eval_batch(st.astype(np.float32))
    assert np.abs(probs[0] - po).max() < 1e-5
    eng.close()


def test_full_batch_pipelined_games_equal_oracle_games_with_gpu_priors(az, oracle):
    """Configuration C2 as the benchmark runs it (1024 slots, 8x8 / 5-in-row / 400 playouts, CNN leaf
    evaluation, pipelined device refill): whole games taken from the first cohort must equal, tuple
    for tuple, the oracle's self-play game for the same seed when the oracle's tree is fed the GPU
    CNN's own float32 outputs (pi to 1e-12, states and z exactly)."""
    n, k, npo, B, seed0 = 8, 5, 400, 1024, 4000
    params = helpers.make_params(n, seed=108)
    eng = az.SelfPlayEngine(n, k, npo, B, 5.0)
    eng.set_weights(params)
    ev = az.SelfPlayEngine(n, k, 1, 1)
    ev.set_weights(params)
    want = (3, 517, 1023)
    got = {}

    def sink(s):
        for gid in want:
            sel = np.nonzero(s["game_id"] == gid)[0]
            if len(sel):
                got[gid] = {key: s[key][sel[np.argsort(s["ply"][sel])]] for key in s}
    from alphazero_gomoku_amd.engine import SelfPlayRunner
    run = SelfPlayRunner(eng, temp=1.0, seed0=seed0, total=None, on_samples=sink, pipelined=True)
    run.start()
    for _ in range(70):
        run.step()
        if len(got) == len(want):
            break
    run.finish()
    assert len(got) == len(want), "the chosen games did not finish within 70 move steps"

    def policy(b):
        ob = oracle.OracleBoard(n, n, k)
        ob.b = b
        pr, v = ev.eval_batch(ob.current_state().astype(np.float32))
        return [float(x) for x in pr[0]], float(v[0])
    for gid in want:
        r = oracle.self_play(n, k, policy, 5, npo, 1.0, seed0 + gid)
        g = got[gid]
        assert len(g["z"]) == len(r["moves"]), gid
        np.testing.assert_allclose(g["pi"], r["pi"], rtol=0, atol=1e-12)
        assert np.array_equal(g["z"], r["z"].astype(np.float32)), gid
        assert np.array_equal(g["states"], r["states"].astype(np.float32)), gid
    eng.close()
    ev.close()


@pytest.mark.parametrize("n,cnn_mode", [(n, 0) for n in range(6, 17)] + [(n, 1) for n in (7, 9, 10, 11, 12, 13, 14)],
                         ids=lambda v: str(v))
def test_cnn_and_net_search_on_every_board_size(az, oracle, n, cnn_mode):
    """Split-f16 CNN kernels exist for every square board from 6x6 to 16x16 (two leaves per
    workgroup up to 8x8, one leaf with the sparse FC above), float32-MFMA ones from 6x6 to 15x15 (6, 8
    and 15 are covered by the tests above): the forward pass against the oracle's float64 net (1e-5;
    required 1e-4), and a net-driven search bit-exact against the oracle's tree fed the GPU's own outputs."""
    k, npo = min(5, n), 48
    params = helpers.make_params(n, seed=300 + n, scale=2.0)
    ev = az.SelfPlayEngine(n, k, 1, 1, cnn_mode=cnn_mode)
    ev.set_weights(params)
    onet = oracle.OracleNet(n, n, params, precise=True)
    for pre in helpers.random_positions(n, k, 4, seed=n):
        ob = oracle.OracleBoard(n, n, k)
        ob.init_board()
        ok = True
        for m in pre[: n * n // 2]:
            ob.do_move(m)
            if ob.game_end()[0]:
                ok = False
                break
        if not ok:
            continue
        po, vo = onet.forward(ob.current_state())
        pg, vg = ev.eval_batch(ob.current_state()[None].astype(np.float32))
        assert np.abs(pg[0] - po).max() < 1e-5 and abs(vg[0] - vo) < 1e-5, n
    eng = az.SelfPlayEngine(n, k, npo, 1, cnn_mode=cnn_mode)
    eng.set_weights(params)
    eng.reset_games()
    prefix = [(n // 2) * n + n // 2, (n // 2) * n + n // 2 + 1]
    eng.set_position(0, stones_of(prefix), prefix[-1], 1)
    eng.search()
    visits, q, p, _ = eng.root_stats()

    def policy(b):
        ob = oracle.OracleBoard(n, n, k)
        ob.b = b
        pr, v = ev.eval_batch(ob.current_state().astype(np.float32))
        return [float(x) for x in pr[0]], float(v[0])
    ob = oracle.OracleBoard(n, n, k)
    ob.init_board()
    for m in prefix:
        ob.do_move(m)
    t = oracle.OracleMCTS(policy, c_puct=5, n_playout=npo)
    acts, ovis, oq, op, _ = t.get_move_probs(ob, temp=1.0)
    assert np.array_equal(visits[0][acts], ovis) and np.array_equal(q[0][acts], oq) and np.array_equal(p[0][acts], op), n
    eng.close()
    ev.close()


@pytest.mark.parametrize("n,B", [(15, 5), (9, 4), (12, 3)])
def test_two_wave_group_search_kernel_with_both_sets_live(az, oracle, n, B):
    """Boards from 9x9 to 15x15 are searched by the kernel whose workgroups hold two games and two
    wave groups (net waves / tree waves, hand-over through LDS counters).  Several games, so that
    both sets of a workgroup are live and (odd B) one workgroup has a dead set; three plies with
    tree re-use; visits, Q and priors bit-equal to the oracle's tree (mcts_alphaZero.py:113-148)
    fed the evaluation kernel's own outputs -- the two kernels must agree to the bit."""
    k, npo = 5, 96
    params = helpers.make_params(n, seed=100 + n)
    eng = az.SelfPlayEngine(n, k, npo, B)
    eng.set_weights(params)
    ev = az.SelfPlayEngine(n, k, 1, 1)
    ev.set_weights(params)
    eng.reset_games(seeds=np.arange(B, dtype=np.uint32) + 5)

    def policy(b):
        ob = oracle.OracleBoard(n, n, k)
        ob.b = b
        pr, v = ev.eval_batch(ob.current_state().astype(np.float32))
        return [float(x) for x in pr[0]], float(v[0])
    trees = [oracle.OracleMCTS(policy, c_puct=5, n_playout=npo) for _ in range(B)]
    boards = []
    for g in range(B):
        ob = oracle.OracleBoard(n, n, k)
        ob.init_board()
        boards.append(ob)
    for ply in range(3):
        eng.search()
        visits, q, p, rn = eng.root_stats()
        moves, _ = eng.sample_and_advance(1.0, want_pi=False)
        for g in range(B):
            acts, ovis, oq, op, _ = trees[g].get_move_probs(boards[g], temp=1.0)
            assert rn[g] == trees[g].root_visits(), (ply, g)
            assert np.array_equal(visits[g][acts], ovis), (ply, g)
            assert np.array_equal(q[g][acts], oq) and np.array_equal(p[g][acts], op), (ply, g)
            trees[g].update_with_move(int(moves[g]))
            boards[g].do_move(int(moves[g]))
    eng.close()
    ev.close()


@pytest.mark.parametrize("n,k", [(3, 3), (4, 2), (5, 4), (6, 4), (7, 1), (8, 5), (9, 5), (11, 6), (13, 5), (15, 5), (16, 5), (16, 9)])
def test_random_games_on_every_board_size_follow_the_oracle_rules(az, oracle, n, k):
    """Board.do_move / game_end / current_state on random move sequences, 48 games per board size
    stepping together through azg_board_step_batch until the oracle's board says each has ended:
    ended flag, winner and all four planes after EVERY move must equal the oracle's (game.py:92-169),
    including rows that wrap, both diagonals, lines longer than n_in_row, n_in_row below 3 (a line
    may exist before the reference's stone-count threshold lets it count) and full boards."""
    rng = np.random.RandomState(1000 * n + k)
    nb, S = 48, n * n
    seqs = [rng.permutation(S) for _ in range(nb)]
    # half of the games are steered: one colour fills a line early so that wins in every
    # direction occur at all (uniformly random games on large boards mostly end as ties)
    for i in range(0, nb, 2):
        r0, c0 = rng.randint(0, n), rng.randint(0, n)
        dr, dc = [(0, 1), (1, 0), (1, 1), (1, -1)][(i // 2) % 4]
        line = [(r0 + j * dr) * n + (c0 + j * dc) for j in range(k + 1)
                if 0 <= r0 + j * dr < n and 0 <= c0 + j * dc < n]
        rest = [int(m) for m in seqs[i] if m not in line]
        mixed, a, b = [], list(line), rest
        while a or b:                                    # the line's cells go to the player who moves first
            if a:
                mixed.append(a.pop(0))
            elif b:
                mixed.append(b.pop(0))
            if b:
                mixed.append(b.pop(0))
        seqs[i] = np.array(mixed)
    boards = []
    for _ in range(nb):
        ob = oracle.OracleBoard(n, n, k)
        ob.init_board()
        boards.append(ob)
    eng = az.SelfPlayEngine(n, k, 1, 1, prior_mode=az.PRIOR_HASH)
    st = np.zeros((nb, 2, 4), dtype=np.uint64)
    ply = np.zeros(nb, dtype=np.int32)
    last = np.full(nb, -1, dtype=np.int32)
    alive = np.ones(nb, dtype=bool)
    outcomes = set()
    for t in range(S):
        act = np.nonzero(alive)[0]
        if not len(act):
            break
        sub_st, sub_ply, sub_last = (np.ascontiguousarray(x[act]) for x in (st, ply, last))
        mv = np.array([seqs[i][t] for i in act], dtype=np.int32)
        pl = np.full(len(act), 1 + t % 2, dtype=np.int32)
        e, w, planes = eng.board_step_batch(sub_st, sub_ply, sub_last, mv, pl)
        st[act], ply[act], last[act] = sub_st, sub_ply, sub_last
        for j, i in enumerate(act):
            boards[i].do_move(int(mv[j]))
            oe, ow = boards[i].game_end()
            assert (bool(e[j]), int(w[j]) if oe else -1) == (oe, ow if oe else -1), (n, k, i, t)
            assert np.array_equal(planes[j], boards[i].current_state().astype(np.float32)), (n, k, i, t, "planes")
            if oe:
                alive[i] = False
                outcomes.add(ow)
    assert not alive.any()
    assert len(outcomes) >= (1 if k == 1 else 2), outcomes
    eng.close()
