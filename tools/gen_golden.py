#!/usr/bin/env python3
"""Generate tests/golden/*.npz by importing the reference's own Python modules.

Runs only in the build container (needs /root/reference).  The fixtures hold
inputs and expected outputs only; no reference source is copied.  Before
writing anything the script re-derives the known-answer tests that were
measured on the reference during the survey (KAT-A .. KAT-F) and aborts if
any of them does not reproduce.

    PYTHONDONTWRITEBYTECODE=1 python tools/gen_golden.py
"""
import ast
import os
import sys
import warnings

sys.dont_write_bytecode = True
REF = os.environ.get("AZ_REFERENCE", "/root/reference")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REF)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402

warnings.filterwarnings("ignore")

import game as ref_game  # noqa: E402
import mcts_alphaZero as ref_mcts  # noqa: E402
import policy_value_net_numpy as ref_np_net  # noqa: E402

import helpers  # noqa: E402

OUT = helpers.GOLDEN
os.makedirs(OUT, exist_ok=True)


# ---------------------------------------------------------------- callbacks
def cb_uniform32(board):
    S = board.width * board.height
    p, v = helpers.uniform32_policy(len(board.availables), S)
    return zip(board.availables, [float(x) for x in p[board.availables]]), v


def cb_hash(board):
    S = board.width * board.height
    p, v = helpers.hash_policy(board.states, board.current_player, S)
    return zip(board.availables, [float(x) for x in p[board.availables]]), v


CALLBACKS = {"uniform32": cb_uniform32, "hash": cb_hash}


def new_board(n, k, moves=()):
    b = ref_game.Board(width=n, height=n, n_in_row=k)
    b.init_board()
    for m in moves:
        b.do_move(m)
    return b


def root_snapshot(mcts):
    items = list(mcts._root._children.items())
    acts = np.array([a for a, _ in items], dtype=np.int32)
    visits = np.array([nd._n_visits for _, nd in items], dtype=np.int32)
    q = np.array([float(nd._Q) for _, nd in items], dtype=np.float64)
    p = np.array([float(nd._P) for _, nd in items], dtype=np.float64)
    return acts, visits, q, p


# ---------------------------------------------------------------- KATs
def check_kats():
    # KAT-A: tree mechanics and tie-break
    b = new_board(6, 4)

    def fn(board):
        k = len(board.availables)
        return zip(board.availables, np.ones(k) / k), 0.0
    m = ref_mcts.MCTS(fn, c_puct=5, n_playout=50)
    acts, _ = m.get_move_probs(b, temp=1.0)
    _, visits, _, _ = root_snapshot(m)
    assert tuple(acts) == tuple(range(36)) and m._root._n_visits == 50
    assert list(visits) == [2] * 13 + [1] * 23, visits
    # KAT-B: full self-play incl. RNG stream
    np.random.seed(42)
    pl = ref_mcts.MCTSPlayer(fn, c_puct=5, n_playout=100, is_selfplay=1)
    g = ref_game.Game(new_board(6, 4))
    winner, data = g.start_self_play(pl, temp=1.0)
    data = list(data)
    assert winner == 2 and len(data) == 18, (winner, len(data))
    assert [d[2] for d in data] == [-1.0, 1.0] * 9
    assert list(np.rint(data[0][1] * 99).astype(int)) == [3] * 27 + [2] * 9
    assert [int(data[0][0][i].sum()) for i in range(4)] == [0, 0, 0, 36]
    assert [int(data[1][0][i].sum()) for i in range(4)] == [0, 1, 1, 0]
    # KAT-C: rules
    b = new_board(6, 4, [0, 6, 1, 7, 2, 8])
    assert b.game_end() == (False, -1)
    b.do_move(3)
    assert b.game_end() == (True, 1) and b.current_player == 2
    # KAT-D: feature planes
    b = new_board(8, 5, [27, 28, 35, 36, 20])
    nz = sorted(zip(*[x.tolist() for x in np.nonzero(b.current_state())]))
    assert nz == [(0, 3, 4), (0, 4, 4), (1, 3, 3), (1, 4, 3), (1, 5, 4), (2, 5, 4)], nz
    # KAT-F: choice == dirichlet first, then one uniform through the cdf
    rs1, rs2 = np.random.RandomState(7), np.random.RandomState(7)
    probs = np.arange(1, 11) / 55.0
    q = 0.75 * probs + 0.25 * rs1.dirichlet(0.1 * np.ones(10))
    mv = rs1.choice(np.arange(10), p=q)
    q2 = 0.75 * probs + 0.25 * rs2.dirichlet(0.1 * np.ones(10))
    cdf = np.cumsum(q2)
    cdf /= cdf[-1]
    assert mv == cdf.searchsorted(rs2.random_sample(), side="right")
    print("KAT-A,B,C,D,F reproduce")


# ---------------------------------------------------------------- G1 rules
def play_record(n, k, moves):
    """Replay moves; per ply (end, winner, current_player) and the state planes."""
    b = new_board(n, k)
    ends, winners, states = [], [], [b.current_state().astype(np.uint8)]
    for m in moves:
        b.do_move(m)
        e, w = b.game_end()
        ends.append(int(e))
        winners.append(int(w))
        states.append(b.current_state().astype(np.uint8))
    return ends, winners, states


def gen_rules():
    games = []  # (n, k, moves)
    # scripted cases
    games.append((6, 4, [0, 6, 1, 7, 2, 8, 3]))                    # KAT-C horizontal win
    games.append((6, 4, [3, 0, 4, 1, 5, 2, 6, 12, 7, 13, 8]))      # row wrap 3,4,5,6 is not a win; later 6,7,8 + ... check
    games.append((6, 4, [0, 1, 6, 2, 12, 3, 18]))                  # vertical win
    games.append((6, 4, [0, 1, 7, 2, 14, 3, 21]))                  # diagonal up-right
    games.append((6, 4, [3, 0, 8, 1, 13, 2, 18]))                  # diagonal up-left
    games.append((6, 4, [0, 30, 1, 31, 2, 32, 4, 33]))             # player 2 wins on top row
    games.append((6, 4, [0, 35, 1, 34, 3, 33, 4, 20, 2]))          # overline (5 on a 4-in-row board)
    games.append((8, 5, [0, 8, 1, 9, 2, 10, 3, 11, 4]))            # 8x8 horizontal
    games.append((8, 5, [4, 60, 5, 61, 6, 62, 7, 63, 8, 59]))      # wrap 4..8 no win, then p2 59..63 wins
    games.append((8, 5, [7, 0, 14, 1, 21, 2, 28, 3, 35]))          # anti-diagonal
    games.append((15, 5, [112, 0, 113, 1, 114, 2, 115, 3, 116]))   # 15x15 horizontal
    games.append((15, 5, [14, 100, 15, 101, 16, 102, 17, 103, 18, 120, 19]))  # wrap over row edge 14|15.. : 15..19 wins
    games.append((3, 3, [0, 1, 3, 4, 6]))                          # 5 stones = n+2: column win found
    games.append((3, 3, [0, 3, 1, 4, 2]))                          # row win at exactly n+2 stones
    # full-board tie on 3x3 with n_in_row 3: X O X / X O O / O X X (move ids, no line)
    games.append((3, 3, [0, 1, 2, 4, 3, 5, 7, 6, 8]))
    # random games to the end
    rs = np.random.RandomState(2024)
    for (n, k, cnt) in [(6, 4, 30), (8, 5, 30), (15, 5, 12), (9, 5, 6)]:
        for _ in range(cnt):
            perm = [int(x) for x in rs.permutation(n * n)]
            b = new_board(n, k)
            mv = []
            for m in perm:
                b.do_move(m)
                mv.append(m)
                if b.game_end()[0]:
                    break
            games.append((n, k, mv))
    # Yixin game records shipped with the reference (logs/steps.log): 15x15
    with open(os.path.join(REF, "logs", "steps.log")) as f:
        lines = [ln for ln in f if ln.startswith("{'steps'")]
    picked = 0
    for ln in lines[::97]:
        try:
            rec = ast.literal_eval(ln.strip())
        except Exception:
            continue
        mv = [int(x) for x in rec["steps"]]
        if len(mv) < 9 or len(set(mv)) != len(mv) or max(mv) >= 225 or min(mv) < 0:
            continue
        games.append((15, 5, mv))
        picked += 1
        if picked >= 40:
            break
    meta, moves_flat, ends_flat, winners_flat, states_list = [], [], [], [], []
    for (n, k, mv) in games:
        ends, winners, states = play_record(n, k, mv)
        meta.append((n, k, len(mv)))
        moves_flat += mv
        ends_flat += ends
        winners_flat += winners
        # keep the planes after the last move and after ~half of the moves
        states_list.append(np.stack([states[len(mv) // 2], states[-1]]).reshape(-1))
    np.savez_compressed(
        os.path.join(OUT, "g1_rules.npz"),
        meta=np.array(meta, dtype=np.int32), moves=np.array(moves_flat, dtype=np.int32),
        ends=np.array(ends_flat, dtype=np.int8), winners=np.array(winners_flat, dtype=np.int8),
        states=np.concatenate(states_list).astype(np.uint8))
    print("g1_rules: %d games" % len(games))


# ---------------------------------------------------------------- G2 tree
def gen_tree():
    cases = []
    specs = [
        # (n, k, prefix moves, callback, c_puct, [playout counts in sequence with re-rooting])
        (6, 4, [], "uniform32", 5, [1, 2, 50, 200]),
        (6, 4, [], "hash", 5, [1, 2, 50, 200, 400]),
        (8, 5, [], "uniform32", 5, [50, 400]),
        (8, 5, [], "hash", 5, [1, 2, 50, 200, 400]),
        (8, 5, [27, 28, 35, 36, 20], "hash", 5, [400]),
        # near-terminal: both sides have open threes/fours so terminal leaves occur
        (6, 4, [0, 6, 1, 7, 2, 8], "hash", 5, [200]),
        (6, 4, [0, 6, 1, 7, 2, 8], "uniform32", 5, [200]),
        (8, 5, [0, 8, 1, 9, 2, 10, 3, 11], "hash", 5, [400]),
        (8, 5, [0, 8, 1, 9, 2, 10, 3, 11], "hash", 1.5, [300]),
        (15, 5, [], "hash", 5, [120]),
        (15, 5, [112, 113, 97, 98, 127], "hash", 5, [150]),
        (9, 5, [40, 41], "hash", 3, [150]),
        # almost full board (few children, ties and terminal leaves everywhere)
        (3, 3, [0, 1, 2, 4], "hash", 5, [60]),
    ]
    for (n, k, prefix, cbn, c_puct, counts) in specs:
        for npo in counts:
            b = new_board(n, k, prefix)
            m = ref_mcts.MCTS(CALLBACKS[cbn], c_puct=c_puct, n_playout=npo)
            acts, probs = m.get_move_probs(b, temp=1.0)
            a1, v1, q1, p1 = root_snapshot(m)
            # temp 1e-3 softmax on the same visits
            probs_t = ref_mcts.softmax(1.0 / 1e-3 * np.log(np.array(v1) + 1e-10))
            # re-root at the most visited child (first on ties) and search again
            mv = int(a1[int(np.argmax(v1))])
            m.update_with_move(mv)
            root_n_before = m._root._n_visits
            b.do_move(mv)
            e, _ = b.game_end()
            if not e:
                m.get_move_probs(b, temp=1.0)
                a2, v2, q2, p2 = root_snapshot(m)
                root_n_after = m._root._n_visits
            else:
                a2 = v2 = np.zeros(0, np.int32)
                q2 = p2 = np.zeros(0)
                root_n_after = root_n_before
            cases.append(dict(n=n, k=k, prefix=np.asarray(prefix, dtype=np.int32), cb=cbn, c_puct=c_puct, n_playout=npo,
                              acts=a1, visits=v1, q=q1, p=p1, probs=np.array(probs),
                              probs_t=probs_t, move=mv, root_n_before=root_n_before,
                              root_n_after=root_n_after, acts2=a2, visits2=v2, q2=q2, p2=p2))
    np.savez_compressed(os.path.join(OUT, "g2_tree.npz"), **helpers.pack_records("cases", cases))
    print("g2_tree: %d cases" % len(cases))


# ---------------------------------------------------------------- G3 games
def record_game(n, k, fn, c_puct, n_playout, temp, seed):
    np.random.seed(seed)
    board = new_board(n, k)
    player = ref_mcts.MCTSPlayer(fn, c_puct=c_puct, n_playout=n_playout, is_selfplay=1)
    game = ref_game.Game(board)
    visits_log, moves = [], []
    orig = player.get_action

    def spy(bd, temp=1e-3, return_prob=0):
        acts, probs = player.mcts.get_move_probs(bd, temp)
        a, v, _, _ = root_snapshot(player.mcts)
        vis = np.zeros(n * n, dtype=np.int32)
        vis[a] = v
        visits_log.append(vis)
        move_probs = np.zeros(n * n)
        move_probs[list(acts)] = probs
        move = np.random.choice(acts, p=0.75 * probs + 0.25 * np.random.dirichlet(0.1 * np.ones(len(probs))))
        player.mcts.update_with_move(move)
        moves.append(int(move))
        return move, move_probs
    # the spy repeats MCTSPlayer.get_action's self-play branch statement by statement so
    # that the visit counts can be observed before the tree is re-rooted
    player.get_action = spy
    winner, data = game.start_self_play(player, temp=temp)
    player.get_action = orig
    data = list(data)
    return dict(n=n, k=k, c_puct=c_puct, n_playout=n_playout, temp=temp, seed=seed,
                winner=int(winner), moves=np.array(moves, dtype=np.int32),
                visits=np.stack(visits_log),
                pi=np.stack([d[1] for d in data]),
                z=np.array([d[2] for d in data]),
                states=np.stack([d[0] for d in data]).astype(np.uint8))


def check_spy_equivalence():
    """The spy must consume the RNG and produce moves exactly like the unmodified path."""
    np.random.seed(5)
    pl = ref_mcts.MCTSPlayer(cb_hash, c_puct=5, n_playout=60, is_selfplay=1)
    w0, d0 = ref_game.Game(new_board(6, 4)).start_self_play(pl, temp=1.0)
    d0 = list(d0)
    r = record_game(6, 4, cb_hash, 5, 60, 1.0, 5)
    assert w0 == r["winner"] and len(d0) == len(r["moves"])
    assert all(np.array_equal(a[1], b) for a, b in zip(d0, r["pi"]))
    print("spy equivalence ok")


def gen_games(net_params_6):
    games = []
    specs = [
        (6, 4, "uniform32", 5, 100, 1.0, 42),
        (6, 4, "hash", 5, 200, 1.0, 0),
        (6, 4, "hash", 5, 200, 1.0, 1),
        (6, 4, "hash", 5, 200, 1e-3, 2),
        (8, 5, "hash", 5, 400, 1.0, 0),
        (8, 5, "hash", 5, 400, 1.0, 1),
        (8, 5, "hash", 5, 400, 1.0, 2),
        (8, 5, "uniform32", 5, 400, 1.0, 3),
        (8, 5, "hash", 2.5, 128, 0.5, 4),
        (15, 5, "hash", 5, 100, 1.0, 0),
        (9, 5, "hash", 5, 64, 1.0, 7),
    ]
    for (n, k, cbn, c, npo, temp, seed) in specs:
        g = record_game(n, k, CALLBACKS[cbn], c, npo, temp, seed)
        g["cb"] = cbn
        games.append(g)
        print("  game %dx%d %s npo=%d seed=%d: %d plies winner %d" % (n, n, cbn, npo, seed, len(g["moves"]), g["winner"]))
    # configuration 1 of the benchmark list: 6x6, 4 in a row, 200 playouts, numpy net
    net = ref_np_net.PolicyValueNetNumpy(6, 6, helpers.torch_to_theano_layout(net_params_6))

    def cb_net(board):
        ap, v = net.policy_value_fn(board)
        # float32 convention: the net's outputs are rounded to float32 and widened once
        return [(a, float(np.float32(p))) for a, p in ap], float(np.float32(v))
    g = record_game(6, 4, cb_net, 5, 200, 1.0, 0)
    g["cb"] = "numpy_net_f32"
    games.append(g)
    print("  game 6x6 numpy net: %d plies winner %d" % (len(g["moves"]), g["winner"]))
    np.savez_compressed(os.path.join(OUT, "g3_games.npz"), **helpers.pack_records("games", games))
    print("g3_games: %d games" % len(games))


# ---------------------------------------------------------------- G4 CNN
def gen_cnn():
    import torch
    import policy_value_net_pytorch as ref_torch_net
    out = {}
    for (n, k, count, scale) in [(6, 4, 48, 1.0), (8, 5, 64, 1.0), (8, 5, 32, 3.0), (15, 5, 12, 1.0)]:
        params = helpers.make_params(n, seed=100 + n, scale=scale)
        tnet = ref_torch_net.Net(n, n)
        sd = tnet.state_dict()
        for key, p in zip(sd.keys(), params):
            assert tuple(sd[key].shape) == p.shape, (key, sd[key].shape, p.shape)
            sd[key] = torch.from_numpy(p.copy())
        tnet.load_state_dict(sd)
        nnet = ref_np_net.PolicyValueNetNumpy(n, n, helpers.torch_to_theano_layout(params))
        pvn = ref_torch_net.PolicyValueNet.__new__(ref_torch_net.PolicyValueNet)
        pvn.use_gpu, pvn.board_width, pvn.board_height, pvn.policy_value_net = False, n, n, tnet
        states, probs_t, vals_t, probs_n, vals_n = [], [], [], [], []
        for mv in helpers.random_positions(n, k, count, seed=7 * n):
            b = new_board(n, k, mv)
            st = b.current_state().copy()
            with torch.no_grad():
                p_t, v_t = pvn.policy_value(st.reshape(1, 4, n, n))
            ap, v_n = nnet.policy_value_fn(b)
            pn = np.zeros(n * n)
            # the numpy net returns only the legal entries; recompute all S for the fixture
            X = st.reshape(-1, 4, n, n)
            for i in [0, 2, 4]:
                X = ref_np_net.relu(ref_np_net.conv_forward(X, nnet.params[i], nnet.params[i + 1]))
            Xp = ref_np_net.relu(ref_np_net.conv_forward(X, nnet.params[6], nnet.params[7], padding=0))
            pn = ref_np_net.softmax(ref_np_net.fc_forward(Xp.flatten(), nnet.params[8], nnet.params[9]))
            for a, p in ap:
                assert abs(pn[a] - p) < 1e-15
            states.append(st.astype(np.uint8))
            probs_t.append(p_t[0].astype(np.float32))
            vals_t.append(np.float32(v_t[0][0]))
            probs_n.append(pn)
            vals_n.append(float(v_n))
        tag = "%d_s%g" % (n, scale)
        out["states_" + tag] = np.stack(states)
        out["probs_torch_" + tag] = np.stack(probs_t)
        out["value_torch_" + tag] = np.array(vals_t)
        out["probs_numpy_" + tag] = np.stack(probs_n)
        out["value_numpy_" + tag] = np.array(vals_n)
        d = np.abs(out["probs_torch_" + tag] - out["probs_numpy_" + tag]).max()
        dv = np.abs(out["value_torch_" + tag] - out["value_numpy_" + tag]).max()
        print("  cnn %s: torch-vs-numpy max|dp|=%.2e max|dv|=%.2e" % (tag, d, dv))
        assert d < 1e-5 and dv < 1e-5          # KAT-E weight mapping (float32 torch vs float64 numpy)
    out["spec"] = np.array([(6, 4, 48, 1.0, 106), (8, 5, 64, 1.0, 108), (8, 5, 32, 3.0, 108), (15, 5, 12, 1.0, 115)])
    np.savez_compressed(os.path.join(OUT, "g4_cnn.npz"), **out)
    print("g4_cnn written")


# ---------------------------------------------------------------- G5 RNG
def gen_rng():
    out = {}
    for seed in (0, 1, 42, 2**31 + 5):
        rs = np.random.RandomState(seed)
        draws = []
        for kk in (64, 36, 7, 1, 225):
            d = rs.dirichlet(0.1 * np.ones(kk))
            u = rs.random_sample()
            draws.append(np.concatenate([d, [u]]))
        out["seed_%d" % seed] = np.concatenate(draws)
    np.savez_compressed(os.path.join(OUT, "g5_rng.npz"), **out)
    print("g5_rng written")


# ---------------------------------------------------------------- G7 net-driven 8x8 game
def gen_net_game8():
    """One game at the benchmark's per-game settings (8x8, 5 in a row, 400 playouts, c_puct 5,
    temp 1) played by the reference with its numpy net on seeded random weights."""
    n = 8
    net = ref_np_net.PolicyValueNetNumpy(n, n, helpers.torch_to_theano_layout(helpers.make_params(n, seed=108)))

    def cb_net(board):
        ap, v = net.policy_value_fn(board)
        return [(a, float(np.float32(p))) for a, p in ap], float(np.float32(v))
    g = record_game(n, 5, cb_net, 5, 400, 1.0, 3)
    g["cb"] = "numpy_net_f32"
    g["param_seed"] = 108
    games = [g]
    print("g7_net8: 8x8 %d plies winner %d" % (len(g["moves"]), g["winner"]))
    # a 15x15 game (wide action space, 4 bit-set words per colour), 100 playouts to keep it short
    n = 15
    net15 = ref_np_net.PolicyValueNetNumpy(n, n, helpers.torch_to_theano_layout(helpers.make_params(n, seed=115)))

    def cb_net15(board):
        ap, v = net15.policy_value_fn(board)
        return [(a, float(np.float32(p))) for a, p in ap], float(np.float32(v))
    g = record_game(n, 5, cb_net15, 5, 100, 1.0, 5)
    g["cb"] = "numpy_net_f32"
    g["param_seed"] = 115
    games.append(g)
    print("g7_net8: 15x15 %d plies winner %d" % (len(g["moves"]), g["winner"]))
    np.savez_compressed(os.path.join(OUT, "g7_net8.npz"), **helpers.pack_records("games", games))


# ---------------------------------------------------------------- G6 augmentation
def gen_equi():
    """train.py cannot be imported (Keras import at module level), so nothing here is a run of
    the reference.  The expected values are built from the definition of the 8 images by
    explicit index loops and must agree with oracle.az_oracle.equi_data, which restates
    train.py:116-135 with the reference's own numpy calls."""
    sys.path.insert(0, ROOT)
    from oracle import az_oracle

    def image(m, i, flip):
        # counter-clockwise rotation by i*90 degrees, then optional left-right mirror
        n = m.shape[0]
        r = m
        for _ in range(i % 4):
            q = np.empty_like(r)
            for a in range(n):
                for b in range(n):
                    q[a][b] = r[b][n - 1 - a]
            r = q
        if flip:
            r = r[:, ::-1]
        return r

    out = {}
    rng = np.random.RandomState(7)
    for n in (6, 8, 15):
        cnt = 5
        states = (rng.rand(cnt, 4, n, n) < 0.3).astype(np.float32)
        pi = rng.rand(cnt, n * n)
        pi /= pi.sum(axis=1, keepdims=True)
        z = rng.choice([-1.0, 0.0, 1.0], size=cnt).astype(np.float32)
        ext = []
        for t in range(cnt):
            for i in (1, 2, 3, 4):
                for flip in (False, True):
                    st = np.stack([image(states[t][c], i, flip) for c in range(4)])
                    # pi lives on the un-flipped board; states are stored row-flipped
                    pm = image(pi[t].reshape(n, n)[::-1], i, flip)[::-1].reshape(-1)
                    ext.append((st, pm, z[t]))
        chk = az_oracle.equi_data([(states[i], pi[i], z[i]) for i in range(cnt)], n, n)
        assert len(chk) == len(ext)
        for a, b in zip(ext, chk):
            assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2] == b[2]
        out["states_%d" % n], out["pi_%d" % n], out["z_%d" % n] = states, pi, z
        out["ext_states_%d" % n] = np.stack([e[0] for e in ext]).astype(np.float32)
        out["ext_pi_%d" % n] = np.stack([e[1] for e in ext])
        out["ext_z_%d" % n] = np.asarray([e[2] for e in ext], dtype=np.float32)
    np.savez_compressed(os.path.join(OUT, "g6_equi.npz"), **out)
    print("g6_equi written")


# ---------------------------------------------------------------- G8 rollout opponent / arena
def gen_pure():
    """mcts_pure.MCTSPlayer (the rollout opponent of train.py:216-276) run by the reference:
    single searches from given positions (move, root visit counts, position of the global numpy
    stream afterwards) and whole Game.start_play games, rollout player against rollout player and
    the AlphaZero player (numpy net, float32 convention, is_selfplay=0) against the rollout player."""
    import contextlib
    import io
    import random
    import mcts_pure as ref_pure

    def quiet():
        return contextlib.redirect_stdout(io.StringIO())

    cases = []
    specs = [
        (8, 5, [], 1, 3, 200), (6, 4, [], 0, 1, 100), (8, 5, [27, 28, 35, 36, 20], 0, 7, 150),
        (8, 5, [0, 63, 7, 56], 0, 11, 300), (9, 5, [40, 41, 31], 1, 2, 80),
        (15, 5, [112, 113, 97, 98, 127], 0, 5, 60), (15, 5, [], 0, 6, 40),
        (6, 4, [0, 1, 6, 7, 12, 13, 2, 8], 0, 9, 120),
        (8, 5, [27, 0, 28, 1, 29, 2, 30, 3], 0, 4, 64),            # four in a row on the board: wins inside the tree
        (6, 4, list(range(0, 30)), 0, 8, 50),                       # nearly full board
        (8, 5, [36], 1, 12, 1000),                                  # train.py's pure_mcts_playout_num
    ]
    for (n, k, pre, sp, seed, npo) in specs:
        b = ref_game.Board(width=n, height=n, n_in_row=k)
        b.init_board(sp)
        ok = True
        for m in pre:
            b.do_move(m)
            if b.game_end()[0]:
                ok = False
        if not ok:
            print("  skipped (position already ended):", n, k, pre)
            continue
        pl = ref_pure.MCTSPlayer(c_puct=5, n_playout=npo)
        np.random.seed(seed)
        with quiet():
            mv = pl.mcts.get_move(b)
        vis = np.zeros(n * n, dtype=np.int32)
        for a, c in pl.mcts._root._children.items():
            vis[a] = c._n_visits
        st = np.random.get_state()
        cases.append(dict(n=n, k=k, pre=np.array(pre, dtype=np.int32), start_player=sp, seed=seed, n_playout=npo,
                          c_puct=5.0, move=int(mv), visits=vis, root_visits=int(pl.mcts._root._n_visits),
                          rng_pos=int(st[2]), rng_key_crc=int(np.bitwise_xor.reduce(st[1])),
                          next_double=float(np.random.random_sample())))
        print("  pure case %dx%d npo=%d seed=%d -> move %d" % (n, n, npo, seed, mv))

    class Spy:
        """Forwards to a player and logs its moves (the reference's start_play keeps none)."""
        def __init__(self, inner, log):
            self.inner, self.log = inner, log

        def set_player_ind(self, p):
            self.inner.set_player_ind(p)
            self.player = p

        def get_action(self, board):
            with quiet():
                m = self.inner.get_action(board)
            self.log.append(int(m))
            return m

    games = []
    for (n, k, npo1, npo2, sp, seed) in [(6, 4, 60, 120, 0, 21), (6, 4, 60, 120, 1, 22), (8, 5, 100, 200, 0, 23), (8, 5, 150, 80, 1, 24)]:
        log = []
        np.random.seed(seed)
        g = ref_game.Game(ref_game.Board(width=n, height=n, n_in_row=k))
        w = g.start_play(Spy(ref_pure.MCTSPlayer(c_puct=5, n_playout=npo1), log), Spy(ref_pure.MCTSPlayer(c_puct=5, n_playout=npo2), log),
                         start_player=sp, is_shown=0)
        games.append(dict(kind="pure_vs_pure", n=n, k=k, n_playout1=npo1, n_playout2=npo2, start_player=sp, seed=seed,
                          moves=np.array(log, dtype=np.int32), winner=int(w), next_double=float(np.random.random_sample())))
        print("  pure vs pure %dx%d start %d: %d plies winner %d" % (n, n, sp, len(log), w))

    # AlphaZero player (evaluation mode) against the rollout player.  The reference's evaluation-mode
    # get_action replaces the first move on an EMPTY board by one of nine 15x15 centre cells
    # (mcts_alphaZero.py:251-253), which is an illegal cell on smaller boards, so the games in which
    # the AlphaZero player opens are 15x15 only.
    for (n, k, pseed, npo_az, npo_pure, sp, seed) in [(6, 4, 106, 100, 150, 1, 31), (8, 5, 108, 120, 200, 1, 32),
                                                      (15, 5, 115, 30, 40, 0, 33), (15, 5, 115, 30, 40, 1, 34)]:
        net = ref_np_net.PolicyValueNetNumpy(n, n, helpers.torch_to_theano_layout(helpers.make_params(n, seed=pseed)))

        def cb_net(board, net=net):
            ap, v = net.policy_value_fn(board)
            return [(a, float(np.float32(p))) for a, p in ap], float(np.float32(v))
        log = []
        np.random.seed(seed)
        random.seed(seed)
        g = ref_game.Game(ref_game.Board(width=n, height=n, n_in_row=k))
        az = ref_mcts.MCTSPlayer(cb_net, c_puct=5, n_playout=npo_az)
        w = g.start_play(Spy(az, log), Spy(ref_pure.MCTSPlayer(c_puct=5, n_playout=npo_pure), log), start_player=sp, is_shown=0)
        games.append(dict(kind="az_vs_pure", n=n, k=k, param_seed=pseed, n_playout1=npo_az, n_playout2=npo_pure, start_player=sp,
                          seed=seed, moves=np.array(log, dtype=np.int32), winner=int(w),
                          next_double=float(np.random.random_sample())))
        print("  az vs pure %dx%d start %d: %d plies winner %d" % (n, n, sp, len(log), w))
    np.savez_compressed(os.path.join(OUT, "g8_pure.npz"), **helpers.pack_records("cases", cases), **helpers.pack_records("games", games))
    print("g8_pure written: %d cases, %d games" % (len(cases), len(games)))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "pure":
        gen_pure()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "equi":
        gen_equi()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "net8":
        gen_net_game8()
        sys.exit(0)
    check_kats()
    check_spy_equivalence()
    gen_rules()
    gen_tree()
    gen_rng()
    gen_cnn()
    gen_games(helpers.make_params(6, seed=106))
    gen_equi()
    gen_net_game8()
    gen_pure()
    print("done ->", OUT)
