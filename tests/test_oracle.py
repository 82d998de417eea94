"""The CPU oracle (oracle/az_oracle.c) against the golden vectors that were produced by
importing the reference's Python modules (tools/gen_golden.py).  CPU only."""
import os

import numpy as np
import pytest

import helpers

G = helpers.GOLDEN


def load(name):
    return np.load(os.path.join(G, name), allow_pickle=True)


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
    for key in d.files:
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


def test_config1_game_with_net(oracle):
    """6x6, 4 in a row, 200 playouts, net-driven game: the oracle's net is not bitwise the
    numpy net (different summation order), so only the opening plies are required to match."""
    games = load("g3_games.npz")["games"]
    g = [x for x in games if x["cb"] == "numpy_net_f32"][0]
    net = oracle.OracleNet(6, 6, helpers.make_params(6, seed=106), precise=True)
    r = oracle.self_play(6, 4, "net", 5, 200, 1.0, 0, net=net)
    assert np.array_equal(r["visits"][0], g["visits"][0])
    same = sum(int(a == b) for a, b in zip(r["moves"], g["moves"]))
    assert same >= 1


def test_equi_data_restatement_matches_fixture(oracle):
    """get_equi_data (train.py:116-135): the oracle's numpy restatement against the fixture built
    from explicit index loops (train.py itself is not importable here: parity unpinned by a
    reference run)."""
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
    from the reference: the oracle's first search from the empty board must give the reference's
    visit counts; later plies may part ways (summation order of the two CPU nets differs)."""
    g = [x for x in load("g7_net8.npz")["games"] if int(x["n"]) == 8][0]
    net = oracle.OracleNet(8, 8, helpers.make_params(8, seed=int(g["param_seed"])), precise=True)
    r = oracle.self_play(8, 5, "net", 5, 400, 1.0, int(g["seed"]), net=net)
    assert np.array_equal(r["visits"][0], g["visits"][0])
    assert r["moves"][0] == g["moves"][0]
