"""The CPU oracle (oracle/az_oracle.c) against the golden vectors that were produced by
importing the reference's Python modules (tools/gen_golden.py).  CPU only."""
import os

import numpy as np
import pytest

import helpers

G = helpers.GOLDEN


def load(name):
    return helpers.load_golden(name)


def test_rules_replay(oracle):
    d = load("g1_rules.npz")
    meta, moves, ends, winners, states = d["meta"], d["moves"], d["ends"], d["winners"], d["states"]
    mo = so = 0
    for (n, k, ln) in meta:
        b = oracle.OracleBoard(n, n, k)
        b.init_board()
        mid = None
        for t in range(ln):
            if t == ln // 2:
                mid = b.current_state()
            b.do_move(int(moves[mo + t]))
            e, w = b.game_end()
            assert (int(e), w) == (int(ends[mo + t]), int(winners[mo + t])), (n, k, t)
        if mid is None:
            mid = b.current_state()
        exp = states[so:so + 2 * 4 * n * n].reshape(2, 4, n, n)
        assert np.array_equal(mid, exp[0].astype(np.float64))
        assert np.array_equal(b.current_state(), exp[1].astype(np.float64))
        mo += ln
        so += 2 * 4 * n * n
    assert mo == len(moves) and so == len(states)


def test_board_errors(oracle):
    b = oracle.OracleBoard(3, 3, 4)
    with pytest.raises(Exception):
        b.init_board()
    b = oracle.OracleBoard(6, 6, 4)
    b.init_board()
    b.do_move(5)
    with pytest.raises(ValueError):
        b.do_move(5)


def test_hash_policy_matches_python(oracle):
    import ctypes as C
    for n, k, moves in [(6, 4, [0, 7, 3]), (8, 5, [27, 28, 35, 36, 20]), (15, 5, [112, 200, 3, 64, 65, 128])]:
        b = oracle.OracleBoard(n, n, k)
        b.init_board()
        states = {}
        for i, m in enumerate(moves):
            states[m] = 1 + i % 2
            b.do_move(m)
        S = n * n
        assert b.state_key() == helpers.state_key(states, b.current_player, S)
        probs = (C.c_double * S)()
        v = C.c_double()
        oracle.lib().azo_policy_hash(C.byref(b.b), probs, C.byref(v), None)
        pe, ve = helpers.hash_policy(states, b.current_player, S)
        assert np.array_equal(np.array(probs[:]), pe) and v.value == ve


def test_tree_fixtures_bit_exact(oracle):
    cases = load("g2_tree.npz")["cases"]
    for c in cases:
        n, k = c["n"], c["k"]
        b = oracle.OracleBoard(n, n, k)
        b.init_board()
        for m in c["prefix"]:
            b.do_move(m)
        t = oracle.OracleMCTS(c["cb"], c_puct=c["c_puct"], n_playout=c["n_playout"])
        acts, visits, q, p, probs = t.get_move_probs(b, temp=1.0)
        tag = (n, c["cb"], c["n_playout"], tuple(c["prefix"]))
        assert np.array_equal(acts, c["acts"]), tag
        assert np.array_equal(visits, c["visits"]), tag
        assert np.array_equal(q, c["q"]), tag          # bit-exact doubles
        assert np.array_equal(p, c["p"]), tag
        np.testing.assert_allclose(probs, c["probs"], rtol=0, atol=1e-12)
        t.update_with_move(c["move"])
        assert t.root_visits() == c["root_n_before"]
        b.do_move(c["move"])
        if len(c["acts2"]):
            a2, v2, q2, p2, _ = t.get_move_probs(b, temp=1.0)
            assert np.array_equal(a2, c["acts2"]) and np.array_equal(v2, c["visits2"]), tag
            assert np.array_equal(q2, c["q2"]) and np.array_equal(p2, c["p2"]), tag
            assert t.root_visits() == c["root_n_after"]


def test_rng_matches_numpy_and_fixture(oracle):
    d = load("g5_rng.npz")
    for key in d:
        seed = int(key.split("_")[1])
        r = oracle.OracleRng(seed)
        rs = np.random.RandomState(seed)
        got = []
        for kk in (64, 36, 7, 1, 225):
            dd = r.dirichlet(0.1, kk)
            u = r.random_sample()
            assert np.array_equal(dd, rs.dirichlet(0.1 * np.ones(kk)))
            assert u == rs.random_sample()
            got.append(np.concatenate([dd, [u]]))
        assert np.array_equal(np.concatenate(got), d[key])


def test_seeded_games_bit_exact(oracle):
    games = load("g3_games.npz")["games"]
    for g in games:
        if g["cb"] not in ("uniform32", "hash"):
            continue
        r = oracle.self_play(g["n"], g["k"], g["cb"], g["c_puct"], g["n_playout"], g["temp"], g["seed"])
        tag = (g["n"], g["cb"], g["n_playout"], g["seed"])
        assert np.array_equal(r["moves"], g["moves"]), tag
        assert r["winner"] == g["winner"], tag
        assert np.array_equal(r["visits"], g["visits"]), tag
        assert np.array_equal(r["z"], g["z"]), tag
        np.testing.assert_allclose(r["pi"], g["pi"], rtol=0, atol=1e-12)
        assert np.array_equal(r["states"], g["states"].astype(np.float64)), tag


def test_cnn_against_reference_outputs(oracle):
    d = load("g4_cnn.npz")
    for (n, k, count, scale, seed) in d["spec"]:
        n, count, seed = int(n), int(count), int(seed)
        tag = "%d_s%g" % (n, scale)
        params = helpers.make_params(n, seed=seed, scale=float(scale))
        for precise, tol in ((True, 1e-9), (False, 2e-5)):
            net = oracle.OracleNet(n, n, params, precise=precise)
            for i in range(count):
                p, v = net.forward(d["states_" + tag][i].astype(np.float64))
                np.testing.assert_allclose(p, d["probs_numpy_" + tag][i], rtol=0, atol=tol)
                assert abs(v - d["value_numpy_" + tag][i]) < max(tol, 1e-9) * 10
                np.testing.assert_allclose(p, d["probs_torch_" + tag][i], rtol=0, atol=1e-5)


def _assert_same_game(r, g, tag):
    assert np.array_equal(r["moves"], g["moves"]) and r["winner"] == g["winner"], tag
    assert np.array_equal(r["visits"], g["visits"]), tag
    assert np.array_equal(r["z"], g["z"]), tag
    np.testing.assert_allclose(r["pi"], g["pi"], rtol=0, atol=1e-12)
    assert np.array_equal(r["states"], g["states"].astype(np.float64)), tag


def test_config1_game_with_net(oracle):
    """Configuration C1 (6x6, 4 in a row, 200 playouts, numpy net) as the reference played it:
    the oracle's net sums in another order than the numpy net, but both round their outputs to
    float32 before the tree sees them, and the whole game comes out the same: every move, every
    visit count, pi, z, the winner."""
    games = load("g3_games.npz")["games"]
    g = [x for x in games if x["cb"] == "numpy_net_f32"][0]
    net = oracle.OracleNet(6, 6, helpers.make_params(6, seed=106), precise=True)
    r = oracle.self_play(6, 4, "net", 5, 200, 1.0, 0, net=net)
    _assert_same_game(r, g, "C1")


def test_bundled_trained_nets(oracle):
    """The reference's trained 6x6 and 8x8 nets (mode/best_policy_6_6_4.model, best_policy_8_8_5.model,
    read without unpickling by tools/gen_golden_r2.py): the oracle's forward pass against the
    reference numpy net's outputs on random and played positions, and the reference's seeded
    self-play game with each net (C1's exact setting for 6x6) move for move."""
    d = load("g10_trained.npz")
    for g in d["games"]:
        n, k = g["n"], g["k"]
        params = helpers.torch_to_theano_layout([d["params_%d_%d" % (n, i)] for i in range(16)])   # the mapping is its own inverse
        net = oracle.OracleNet(n, n, params, precise=True)
        for st, pr, v in zip(d["states_%d" % n], d["probs_%d" % n], d["values_%d" % n]):
            p, val = net.forward(st.astype(np.float64))
            np.testing.assert_allclose(p, pr, rtol=0, atol=1e-9)
            assert abs(val - v) < 1e-8
        r = oracle.self_play(n, k, "net", g["c_puct"], g["n_playout"], g["temp"], g["seed"], net=net)
        _assert_same_game(r, g, g["model"])


def test_tie_games(oracle):
    """Seeded 3x3 games of the reference, a fifth of them ties (game.py:396-399: every z is 0,
    winner -1)."""
    games = load("g11_ties.npz")["games"]
    assert sum(g["winner"] == -1 for g in games) >= 2
    for g in games:
        r = oracle.self_play(3, 3, "hash", 5, g["n_playout"], 1.0, g["seed"])
        _assert_same_game(r, g, g["seed"])


def test_equi_data_restatement_matches_fixture(oracle):
    """get_equi_data (train.py:116-135): the oracle's numpy restatement against the outputs of the
    reference's own function body (tools/gen_golden_r2.py compiles that one method out of
    train.py's syntax tree; the module as a whole cannot be imported)."""
    d = np.load(os.path.join(os.path.dirname(__file__), "golden", "g6_equi.npz"))
    for n in (6, 8, 15):
        states, pi, z = d["states_%d" % n], d["pi_%d" % n], d["z_%d" % n]
        ext = oracle.equi_data([(states[i], pi[i], z[i]) for i in range(len(z))], n, n)
        assert len(ext) == 8 * len(z)
        assert np.array_equal(np.stack([e[0] for e in ext]), d["ext_states_%d" % n])
        assert np.array_equal(np.stack([e[1] for e in ext]), d["ext_pi_%d" % n])
        assert np.array_equal(np.asarray([e[2] for e in ext], dtype=np.float32), d["ext_z_%d" % n])


def test_config2_game_with_net(oracle):
    """8x8, 5 in a row, 400 playouts (the benchmark's per-game settings), net-driven game recorded
    from the reference: the oracle must play the reference's
    game: every move, every visit count, pi, z and the winner."""
    g = [x for x in load("g7_net8.npz")["games"] if int(x["n"]) == 8][0]
    net = oracle.OracleNet(8, 8, helpers.make_params(8, seed=int(g["param_seed"])), precise=True)
    r = oracle.self_play(8, 5, "net", 5, 400, 1.0, int(g["seed"]), net=net)
    _assert_same_game(r, g, "C2 game")


def _oracle_board(oracle, n, k, start_player, pre=()):
    b = oracle.OracleBoard(n, n, k)
    b.init_board(start_player)
    for m in pre:
        b.do_move(int(m))
    return b


def test_move_target_is_the_grown_bounding_box(oracle):
    """Board.get_move_target (game.py:35-58): rows/columns 5..9 on an empty board, else the
    stones' bounding box grown by two and clipped."""
    assert _oracle_board(oracle, 6, 4, 0).move_target() == [35]
    assert _oracle_board(oracle, 8, 5, 0).move_target() == [45, 46, 47, 53, 54, 55, 61, 62, 63]
    b = _oracle_board(oracle, 8, 5, 0, [27])
    assert b.move_target() == [r * 8 + c for r in range(1, 6) for c in range(1, 6) if r * 8 + c != 27]
    b = _oracle_board(oracle, 8, 5, 0, [0, 63])
    assert b.move_target() == [m for m in range(64) if m not in (0, 63)]
    b = _oracle_board(oracle, 15, 5, 0, [112, 113])
    assert b.move_target() == [r * 15 + c for r in range(5, 10) for c in range(5, 11) if r * 15 + c not in (112, 113)]


def test_rollout_opponent_searches_match_reference(oracle):
    """mcts_pure.MCTSPlayer searches recorded from the reference: move, root visit counts and
    the position of the numpy stream afterwards, bit-exact."""
    cases = load("g8_pure.npz")["cases"]
    assert len(cases) >= 8
    for c in cases:
        n, k = int(c["n"]), int(c["k"])
        b = _oracle_board(oracle, n, k, int(c["start_player"]), c["pre"])
        rng = oracle.OracleRng(int(c["seed"]))
        move, visits, root_visits = oracle.pure_get_move(b, float(c["c_puct"]), int(c["n_playout"]), rng)
        assert move == c["move"] and root_visits == c["root_visits"], (n, int(c["seed"]))
        assert np.array_equal(visits, c["visits"])
        key, pos = rng.get_state()
        assert pos == c["rng_pos"] and int(np.bitwise_xor.reduce(key)) == c["rng_key_crc"]
        assert rng.random_sample() == c["next_double"]


def _oracle_arena_game(oracle, g, max_plies=None):
    """Game.start_play (game.py:211-290) between the players of a g8 game record, on the oracle."""
    import random
    n, k = int(g["n"]), int(g["k"])
    b = _oracle_board(oracle, n, k, int(g["start_player"]))
    rng = oracle.OracleRng(int(g["seed"]))
    random.seed(int(g["seed"]))
    az = None
    if g["kind"] == "az_vs_pure":
        net = oracle.OracleNet(n, n, helpers.make_params(n, seed=int(g["param_seed"])), precise=True)
        az = oracle.OracleMCTS("net", 5, int(g["n_playout1"]), net=net)
    moves = []
    while True:
        first = b.current_player == 1                    # player1 owns stone 1 (game.py:228-231)
        if first and az is not None:
            # MCTSPlayer.get_action in evaluation mode (mcts_alphaZero.py:243-255)
            acts, _, _, _, probs = az.get_move_probs(b, 1e-3)
            move = int(acts[helpers.choice_index(probs, rng.random_sample())])
            if len(moves) == 0:
                move = random.choice([96, 97, 98, 111, 112, 113, 126, 127, 128])
            az.update_with_move(-1)
        else:
            move, _, _ = oracle.pure_get_move(b, 5, int(g["n_playout1"] if first else g["n_playout2"]), rng)
        moves.append(move)
        b.do_move(move)
        end, winner = b.game_end()
        if end or (max_plies and len(moves) >= max_plies):
            return moves, winner, rng


def test_arena_games_match_reference(oracle):
    """Whole Game.start_play games recorded from the reference.  Rollout player against rollout
    player: every move, the winner and the stream position.  Network player against rollout
    player: the oracle's CPU net sums in another order than the reference's numpy net, so only
    the opening (which includes the rollout player's first search) is required to agree."""
    games = load("g8_pure.npz")["games"]
    assert sum(g["kind"] == "pure_vs_pure" for g in games) >= 4 and sum(g["kind"] == "az_vs_pure" for g in games) >= 4
    for g in games:
        if g["kind"] == "pure_vs_pure":
            moves, winner, rng = _oracle_arena_game(oracle, g)
            assert moves == [int(m) for m in g["moves"]] and winner == g["winner"]
            assert rng.random_sample() == g["next_double"]
        else:
            moves, _, _ = _oracle_arena_game(oracle, g, max_plies=2)
            assert moves == [int(m) for m in g["moves"][:2]], (int(g["n"]), int(g["start_player"]))
