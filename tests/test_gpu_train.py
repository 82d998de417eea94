"""GPU tests of what surrounds the search: the training step and loop, the bundled trained nets,
ties, packed samples, the host-evaluated playout, the activation-range guard, C4 at full size."""
import numpy as np
import pytest

import helpers

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def az():
    import alphazero_gomoku_amd as pkg
    return pkg


def _policy_from(ev, oracle, n, k):
    """Oracle policy callback that returns the GPU CNN's own float32 outputs."""
    def policy(b):
        ob = oracle.OracleBoard(n, n, k)
        ob.b = b
        pr, v = ev.eval_batch(ob.current_state().astype(np.float32))
        return [float(x) for x in pr[0]], float(v[0])
    return policy


@pytest.mark.parametrize("n,batch", [(8, 37), (8, 512), (6, 64), (15, 20), (11, 3), (16, 5), (7, 130)])
def test_native_gradients_match_autograd(az, n, batch):
    """The library's float32 backward pass (azg_train_backward: implicit-GEMM convolutions on the
    float32 matrix cores, heads, fully connected layers, both losses) against the reference's loss
    (policy_value_net_pytorch.py:128-141) differentiated by PyTorch autograd in FLOAT64 on the CPU, with
    PyTorch's own float32 GPU gradients as the yardstick: every one of the 16 gradient tensors is at
    most three times as far from the float64 gradient as PyTorch's float32 one is (plus 2e-6 of the
    tensor's largest entry; large batches put a few head pre-activations within float32 noise of
    zero, which moves both float32 gradients by the same 1e-4); loss and entropy to 1e-5; the forward
    pass alone (azg_train_forward) to 1e-6.  Batches that do not fill the 128-row tiles included."""
    import copy
    import torch
    import torch.nn.functional as F
    from alphazero_gomoku_amd.net import PARAM_ORDER
    S = n * n
    rs = np.random.RandomState(n * 1000 + batch)
    net = az.PolicyValueNet(n, n)
    net.load_numpy_params(helpers.make_params(n, seed=300 + n, scale=1.5))
    s = (rs.rand(batch, 4, n, n) < 0.3).astype(np.float32)
    pi = rs.dirichlet(np.full(S, 0.3), size=batch).astype(np.float32)
    z = rs.choice([-1.0, 0.0, 1.0], size=batch).astype(np.float32)
    grad, loss, ent = net.native_gradients(s, pi, z)

    def autograd(mod, dev, dt):
        mod.zero_grad()
        logp, v = mod(torch.as_tensor(s, device=dev, dtype=dt))
        lo = F.mse_loss(v.view(-1), torch.as_tensor(z, device=dev, dtype=dt)) - torch.mean(torch.sum(torch.as_tensor(pi, device=dev, dtype=dt) * logp, 1))
        lo.backward()
        en = -torch.mean(torch.sum(torch.exp(logp) * logp, 1))
        return {k: p.grad.detach().double().cpu().reshape(-1) for k, p in mod.named_parameters()}, float(lo.detach()), float(en.detach()), logp.detach(), v.detach()
    g32, _, _, logp32, v32 = autograd(net.policy_value_net, net.device, torch.float32)
    g64, loss64, ent64, _, _ = autograd(copy.deepcopy(net.policy_value_net).double().cpu(), "cpu", torch.float64)
    assert abs(loss - loss64) < 1e-5 and abs(ent - ent64) < 1e-5
    off = 0
    for key in PARAM_ORDER:
        k = g64[key].numel()
        mine = grad[off:off + k].double().cpu()
        off += k
        scale = float(g64[key].abs().max())
        err, err32 = float((mine - g64[key]).abs().max()), float((g32[key] - g64[key]).abs().max())
        assert err <= 3.0 * err32 + 2e-6 * scale + 1e-10, (key, err, err32, scale)
    assert off == grad.numel()
    probs, vals = net.policy_value_in_training(torch.as_tensor(s, device=net.device))
    assert float((probs - torch.exp(logp32)).abs().max()) < 1e-6 and float((vals - v32).abs().max()) < 1e-6
    # the KL test's number (train.py:189-193) against the same formula in float64; old = a perturbed copy
    old = torch.softmax(torch.log(probs) + 0.3 * torch.randn_like(probs), dim=1).contiguous()
    probs_again, _, kl = net.policy_value_in_training(torch.as_tensor(s, device=net.device), old)
    o, q = old.double(), probs.double()
    want = float(torch.mean(torch.sum(o * (torch.log(o + 1e-10) - torch.log(q + 1e-10)), dim=1)))
    assert torch.equal(probs, probs_again) and abs(kl - want) <= 1e-12 + 1e-9 * abs(want), (kl, want)
    # and it is the same bits every time
    grad2, loss2, ent2 = net.native_gradients(s, pi, z)
    assert torch.equal(grad, grad2) and loss == loss2 and ent == ent2


@pytest.mark.parametrize("native", [True, False], ids=["hip", "autograd"])
def test_train_step_matches_reference(az, native):
    """PolicyValueNet.train_step on the GPU against the reference's train_step (fixture g9,
    policy_value_net_pytorch.py:117-148): loss and entropy of three steps to 1e-5, the parameters
    after one and three Adam steps, and the HIP CNN's outputs with the updated weights against the
    reference net's own outputs (1e-4).  Adam's first steps move every parameter by about the
    learning rate in the direction of its gradient's sign, so a parameter whose gradient is within
    float32 noise of zero can land 2*lr away on another device; those are counted, not hidden."""
    d = helpers.load_golden("g9_train.npz")
    n, k, seed = [int(x) for x in d["spec"]]
    net = az.PolicyValueNet(n, n, native_train=native)
    net.load_numpy_params(helpers.make_params(n, seed=seed))
    v0 = net.version
    s, p, z = d["states"].astype(np.float32), d["pi"], d["z"]
    total = sum(int(np.prod(x.shape)) for x in helpers.make_params(n, seed=seed))
    for step, lr in enumerate(d["lrs"]):
        loss, ent = net.train_step(s, p, z, float(lr))
        assert abs(loss - d["loss"][step]) < 1e-5 and abs(ent - d["entropy"][step]) < 1e-5, step
        if step in (0, 2):
            far, worst = 0, 0.0
            for i, t in enumerate(net.param_tensors()):
                diff = np.abs(t.detach().cpu().numpy() - d["params_after_%d_%d" % (step + 1, i)])
                far += int((diff > 1e-6).sum())
                worst = max(worst, float(np.median(diff)))
            assert far <= total // 2000 and worst < 1e-7, (step, far, worst)
    assert net.version == v0 + 3
    probs, vals = net.policy_value(s[:16])
    assert np.abs(probs - d["probs_after_3"]).max() < 1e-4 and np.abs(vals.reshape(-1) - d["values_after_3"]).max() < 1e-4


@pytest.mark.parametrize("cnn_mode", [0, 1], ids=["f16x3", "f32"])
def test_bundled_trained_nets_through_the_engine(az, cnn_mode):
    """The reference's trained 6x6 and 8x8 nets (sharply peaked policies, values up to +-1): both CNN
    variants within 1e-4 of the reference numpy net on random and played positions, and the
    reference's seeded self-play game with each net reproduced move for move, visit for visit."""
    d = helpers.load_golden("g10_trained.npz")
    for g in d["games"]:
        n, k = g["n"], g["k"]
        params = [d["params_%d_%d" % (n, i)] for i in range(16)]
        eng = az.SelfPlayEngine(n, k, g["n_playout"], 1, float(g["c_puct"]), cnn_mode=cnn_mode)
        eng.set_weights(params, layout=az._lib.LAYOUT_THEANO)
        probs, vals = eng.eval_batch(d["states_%d" % n].astype(np.float32))
        dp, dv = np.abs(probs - d["probs_%d" % n]).max(), np.abs(vals - d["values_%d" % n]).max()
        assert dp < 1e-4 and dv < 1e-4, (g["model"], dp, dv)
        assert dp < 1e-5 and dv < 1e-5, (g["model"], dp, dv)          # what both variants actually reach
        eng.reset_games(seeds=np.array([g["seed"]], dtype=np.uint32))
        for t in range(len(g["moves"])):
            eng.search()
            visits, _, _, _ = eng.root_stats()
            assert np.array_equal(visits[0], g["visits"][t]), (g["model"], t)
            moves, pi = eng.sample_and_advance(float(g["temp"]))
            assert moves[0] == g["moves"][t], (g["model"], t)
            np.testing.assert_allclose(pi[0], g["pi"][t], rtol=0, atol=1e-12)
        ended, winner, _ = eng.status()
        assert ended[0] and winner[0] == g["winner"]
        assert np.array_equal(eng.drain_samples()["z"], g["z"].astype(np.float32))
        eng.close()


def test_policy_value_net_loads_the_references_pickled_model_format(az, tmp_path):
    """PolicyValueNet(model_file=...) on a file in the format of the reference's bundled
    best_policy_*.model (a protocol-2 pickle of the 16 Theano-layout arrays,
    policy_value_net.py:113-121), written here from the fixture's arrays: read without unpickling
    (model_files.read_pickled_arrays) and evaluated by the HIP CNN within 1e-5 of the outputs the
    reference's numpy net gave for the same weights."""
    import pickle
    d = helpers.load_golden("g10_trained.npz")
    for n in (6, 8):
        f = tmp_path / ("best_policy_%d.model" % n)
        with open(f, "wb") as fh:
            pickle.dump([np.asarray(d["params_%d_%d" % (n, i)]) for i in range(16)], fh, protocol=2)
        net = az.PolicyValueNet(n, n, model_file=str(f))
        probs, vals = net.policy_value(d["states_%d" % n].astype(np.float32))
        assert np.abs(probs - d["probs_%d" % n]).max() < 1e-5 and np.abs(vals[:, 0] - d["values_%d" % n]).max() < 1e-5


def test_activation_range_guard(az):
    """Split-f16 operands overflow above 65504: an activation that large must raise (sticky, named),
    never return numbers; the float32 variant evaluates the same net."""
    n = 8
    params = helpers.make_params(n, seed=1)
    params[1] = np.full(32, 1.0e5, dtype=np.float32)                    # conv1 bias: activations of 1e5
    st = np.zeros((2, 4, n, n), dtype=np.float32)
    st[:, 3] = 1.0
    ok = az.SelfPlayEngine(n, 5, 16, 1, cnn_mode=az._lib.CNN_F32)
    ok.set_weights(params)
    probs, vals = ok.eval_batch(st)
    assert np.isfinite(probs).all() and np.isfinite(vals).all()
    ok.close()
    eng = az.SelfPlayEngine(n, 5, 16, 1)
    eng.set_weights(params)
    with pytest.raises(az._lib.AzgError, match="AZG_ERANGE"):
        eng.eval_batch(st)
    eng.reset_games()
    eng.search()
    with pytest.raises(az._lib.AzgError, match="AZG_ERANGE"):
        eng.status()
    eng.close()


def test_tie_games_through_pipelined_selfplay(az):
    """Seeded 3x3 games recorded from the reference, ties among them (game.py:396-399: z = 0 for
    every ply), played through azg_selfplay_* with finished slots restarted on the device."""
    games = helpers.load_golden("g11_ties.npz")["games"]
    seed0 = min(g["seed"] for g in games)
    by_seed = {g["seed"]: g for g in games}
    assert sorted(by_seed) == list(range(seed0, seed0 + len(games)))
    eng = az.SelfPlayEngine(3, 3, games[0]["n_playout"], 7, 5.0, prior_mode=az.PRIOR_HASH)
    res = eng.play_games(len(games), temp=1.0, seed0=seed0, pipelined=True)
    s = res["samples"]
    ties = 0
    for gid in range(len(games)):
        g = by_seed[seed0 + gid]
        sel = np.nonzero(s["game_id"] == gid)[0]
        sel = sel[np.argsort(s["ply"][sel])]
        assert len(sel) == len(g["moves"]), gid
        assert np.array_equal(s["z"][sel], g["z"].astype(np.float32)), gid
        assert np.array_equal(s["states"][sel], g["states"].astype(np.float32)), gid
        np.testing.assert_allclose(s["pi"][sel], g["pi"], rtol=0, atol=1e-12)
        if g["winner"] == -1:
            ties += 1
            assert len(sel) == 9 and not s["z"][sel].any()
    assert ties >= 2
    eng.close()


def test_packed_samples_and_unfetched_samples(az):
    """azg_selfplay_fetch_packed + azg_unpack_states give the planes azg_selfplay_fetch gives, and a
    move step taken while ready samples are unfetched is refused instead of dropping them."""
    import torch
    from alphazero_gomoku_amd.engine import SelfPlayRunner
    n, k = 8, 5
    outs = []
    for packed in (False, True):
        eng = az.SelfPlayEngine(n, k, 32, 24, prior_mode=az.PRIOR_HASH)
        got = []
        run = SelfPlayRunner(eng, 1.0, 77, 0, 1, 60, got.append, device_samples=True, pipelined=True, packed_samples=packed)
        run.start()
        while run.games_done < 60:
            run.step()
        run.finish()
        t = {key: torch.cat([c[key] for c in got]) for key in got[0]}
        if packed:
            t["states"] = eng.unpack_states(t["stones"], t["last"], t["ply"])
        outs.append({key: v.cpu().numpy() for key, v in t.items()})
        eng.close()
    a, b = outs
    assert len(a["z"]) == len(b["z"]) > 500
    # which slot a finished slot's next game id lands in is decided by an atomic counter on the
    # device, so two runs deliver the same games in different orders: compare by (game id, ply)
    oa, ob = (np.lexsort((o["ply"], o["game_id"])) for o in (a, b))
    for key in ("states", "pi", "z", "game_id", "ply"):
        assert np.array_equal(a[key][oa], b[key][ob]), key
    eng = az.SelfPlayEngine(n, k, 32, 24, prior_mode=az.PRIOR_HASH)
    eng.selfplay_begin(seed0=77)
    with pytest.raises(az._lib.AzgError, match="not fetched"):
        for _ in range(64):
            eng.selfplay_step()                                  # some game ends, its samples become ready ...
    ready = eng.selfplay_end()
    eng.selfplay_fetch(ready)
    eng.close()


def test_callback_player_reproduces_reference_game(az, oracle):
    """MCTSPlayer with an arbitrary policy_value_fn (mcts_alphaZero.py:98-103): configuration C1's
    game of the reference (6x6, 4 in a row, 200 playouts, numpy net) with a host-side net as the
    callback -- the GPU selects, the host evaluates, the GPU expands and backs up."""
    g = [x for x in helpers.load_golden("g3_games.npz")["games"] if x["cb"] == "numpy_net_f32"][0]
    n, k = g["n"], g["k"]
    net = oracle.OracleNet(n, n, helpers.make_params(n, seed=106), precise=True)
    calls = [0]

    def policy_value_fn(board):
        calls[0] += 1
        p, v = net.forward(board.current_state())
        return zip(board.availables, [float(np.float32(p[a])) for a in board.availables]), float(np.float32(v))
    player = az.MCTSPlayer(policy_value_fn, c_puct=g["c_puct"], n_playout=g["n_playout"], is_selfplay=1)
    np.random.seed(g["seed"])
    winner, data = az.Game(az.Board(width=n, height=n, n_in_row=k)).start_self_play(player, temp=g["temp"])
    data = list(data)
    assert winner == g["winner"] and len(data) == len(g["moves"])
    for t, (st, pi, z) in enumerate(data):
        assert np.array_equal(st, g["states"][t]) and z == g["z"][t], t
        np.testing.assert_allclose(pi, g["pi"][t], rtol=0, atol=1e-12)
    assert calls[0] > 0.9 * g["n_playout"] * len(g["moves"]) - 400


def test_callback_values_outside_the_unit_range(az, oracle):
    """azg_playout_begin / azg_playout_end accept any finite value and priors of any size or sign
    (mcts_alphaZero.py:98-103 puts no bound on what policy_value_fn returns): with values of +-50 and
    signed, unnormalised priors the tree must still be the oracle's, visit for visit and Q for Q --
    the descent that serves these calls takes the double-precision pass at every level."""
    n, k, npo = 6, 4, 300
    S = n * n

    def fn(cells_key):
        rs = np.random.RandomState(cells_key % (2 ** 31))
        pri = (rs.rand(S) * 40.0 - 20.0).astype(np.float32)
        # many exact ties and near-ties between children, far from the unit range
        pri[rs.rand(S) < 0.5] = np.float32(7.25)
        val = float(np.float32(rs.rand() * 100.0 - 50.0))
        return pri, val

    def key_of(s1, s2, cur):
        return (s1 * 1000003 + s2 * 7919 + cur) % (2 ** 61 - 1)

    eng = az.SelfPlayEngine(n, k, npo, 1, prior_mode=az.PRIOR_HASH)
    eng.reset_games(seeds=np.array([5], dtype=np.uint32))
    for _ in range(npo):
        s1, s2, ply, last, cur, term = eng.playout_begin(0)
        if term:
            eng.playout_end(None, 0.0, 0)
        else:
            pri, val = fn(key_of(s1, s2, cur))
            eng.playout_end(pri, val, 0)
    visits, q, p, rn = eng.root_stats()
    eng.close()

    def policy(b):
        s1 = s2 = 0
        for c in range(S):
            if b.cell[c] == 1:
                s1 |= 1 << c
            elif b.cell[c] == 2:
                s2 |= 1 << c
        pri, val = fn(key_of(s1, s2, b.current_player))
        return [float(x) for x in pri], val
    ob = oracle.OracleBoard(n, n, k)
    ob.init_board()
    t = oracle.OracleMCTS(policy, c_puct=5, n_playout=npo)
    acts, ovis, oq, op, _ = t.get_move_probs(ob, temp=1.0)
    assert int(rn[0]) == npo and np.array_equal(visits[0][acts], ovis)
    assert np.array_equal(q[0][acts], oq) and np.array_equal(p[0][acts], op)


def test_matrix_gate_table_is_cleared_and_give_ups_are_counted(az):
    """The per-CU matrix gate only schedules: with every gate word left set (as a launch that died
    inside a gated span would leave them) a search gives the same tree, the waits that ran out show
    up in azg_gate_stats, and the next azg_create / azg_selfplay_begin start from a clean table."""
    n, k, npo = 8, 5, 24
    params = helpers.make_params(n, seed=108, scale=2.0)
    seeds = np.arange(4, dtype=np.uint32) + 11

    def run(poke):
        eng = az.SelfPlayEngine(n, k, npo, 4)
        eng.set_weights(params)
        eng.reset_games(seeds=seeds)
        if poke:
            eng.gate_poke(1)
            assert eng.gate_stats()[1] == 4096
        g0 = eng.gate_stats()[0]
        eng.search()
        out = eng.root_stats()
        g1 = eng.gate_stats()[0]
        eng.close()
        return out, g1 - g0
    clean, _ = run(False)
    dirty, gave_up = run(True)
    for a, b in zip(clean, dirty):
        assert np.array_equal(a, b)
    assert gave_up > 0
    eng = az.SelfPlayEngine(n, k, npo, 4)
    eng.gate_poke(1)
    eng2 = az.SelfPlayEngine(n, k, npo, 4)                   # azg_create clears the table of the GPU
    assert eng2.gate_stats()[1] == 0
    eng2.close()
    eng.gate_poke(1)
    eng.set_weights(params)
    eng.selfplay_begin(1.0, 0, 0, 1, None)                   # and so does azg_selfplay_begin
    eng.selfplay_step()
    eng.selfplay_fetch(0)
    eng.selfplay_end()
    eng.close()


def test_c4_full_size_slots_against_oracle(az, oracle):
    """Configuration C4 at full size (512 games of 15x15, 5 in a row, 800 playouts, CNN): slots 0,
    255 and 511 over three plies against the oracle's tree fed the GPU CNN's own outputs -- every
    root visit count and Q bit-equal, tree reuse included."""
    n, k, npo, B = 15, 5, 800, 512
    params = helpers.make_params(n, seed=115)
    eng = az.SelfPlayEngine(n, k, npo, B, 5.0)
    eng.set_weights(params)
    ev = az.SelfPlayEngine(n, k, 1, 1)
    ev.set_weights(params)
    eng.reset_games(seeds=np.arange(B, dtype=np.uint32) + 9000)
    slots = (0, 255, 511)
    trees = {g: oracle.OracleMCTS(_policy_from(ev, oracle, n, k), c_puct=5, n_playout=npo) for g in slots}
    boards = {}
    for g in slots:
        boards[g] = oracle.OracleBoard(n, n, k)
        boards[g].init_board()
    for ply in range(3):
        eng.search()
        visits, q, p, rn = eng.root_stats()
        moves, _ = eng.sample_and_advance(1.0, want_pi=False)
        for g in slots:
            acts, ovis, oq, op, _ = trees[g].get_move_probs(boards[g], temp=1.0)
            assert rn[g] == trees[g].root_visits(), (g, ply)
            assert np.array_equal(visits[g][acts], ovis) and np.array_equal(q[g][acts], oq) and np.array_equal(p[g][acts], op), (g, ply)
            trees[g].update_with_move(int(moves[g]))
            boards[g].do_move(int(moves[g]))
    eng.close()
    ev.close()


def test_training_loop_c5_shape(az):
    """The loop of configuration C5 in small: 8x8, two batches of 256 games with policy updates.
    The replay buffer holds 8 tuples per ply in get_equi_data's order (train.py:116-135), the
    engine plays the second batch with updated weights, every update trains on the whole mini-batch."""
    import torch
    from alphazero_gomoku_amd.train import TrainPipeline
    from alphazero_gomoku_amd.train_data import equi_data_arrays
    torch.manual_seed(0)
    tp = TrainPipeline(board_width=8, board_height=8, n_in_row=5, n_playout=48, play_batch_size=256, n_parallel=256,
                       buffer_size=400000, batch_size=512, epochs=3, game_batch_num=2, check_freq=0, updates_per_batch=4)
    log = []
    v0 = tp.policy_value_net.version
    tp.run(on_batch=lambda i, n_samples, loss, ent: log.append((i, n_samples, loss, ent, tp._engine_version)))
    assert tp.games_played == 512 and tp.policy_updates == 8
    assert len(tp.data_buffer) == 8 * (log[0][1] + log[1][1])
    assert all(np.isfinite(l[2]) and np.isfinite(l[3]) for l in log)
    assert log[0][4] == v0 and log[1][4] > v0               # the engine's weights moved on between the batches
    assert tp.policy_value_net.version >= v0 + 8
    # buffer rows 0..7 are the eight images of one tuple; image 6 is the rotation by 4 * 90 degrees,
    # i.e. the tuple itself (train.py:121-128)
    buf = tp.data_buffer
    assert buf.start == 0
    so, po, zo = equi_data_arrays(tp.engine, buf.states[6:7].float(), buf.pi[6:7], buf.z[6:7])
    assert torch.equal(so, buf.states[0:8].float()) and torch.equal(po, buf.pi[0:8]) and torch.equal(zo, buf.z[0:8])
    # continuous mode: slots never idle, updates run between move steps
    tp2 = TrainPipeline(board_width=8, board_height=8, n_in_row=5, n_playout=48, play_batch_size=128, n_parallel=128,
                        buffer_size=400000, batch_size=512, epochs=2, game_batch_num=3, check_freq=0, updates_per_batch=2,
                        continuous=True)
    tp2.run()
    assert tp2.games_played == 384 and tp2.policy_updates == 6 and len(tp2.data_buffer) > 8 * 384 * 8


def test_device_replay_buffer_matches_deque(az):
    """DeviceReplayBuffer against collections.deque(maxlen) + random.sample (train.py:82,155,167)."""
    import collections
    import random
    import torch
    from alphazero_gomoku_amd.train_data import DeviceReplayBuffer
    n = 6
    dev = torch.device("cuda", 0)
    rng = np.random.RandomState(3)
    buf, ref = DeviceReplayBuffer(50, n, dev, rng=random), collections.deque(maxlen=50)
    for kk in (3, 19, 7, 30, 64, 1, 49, 50, 12):
        st = (rng.rand(kk, 4, n, n) < 0.4).astype(np.float32)
        pi = rng.rand(kk, n * n)
        z = rng.choice([-1.0, 0.0, 1.0], size=kk).astype(np.float32)
        buf.extend(torch.from_numpy(st).to(dev), torch.from_numpy(pi).to(dev), torch.from_numpy(z).to(dev))
        ref.extend([(st[i], pi[i], z[i]) for i in range(kk)])
        assert len(buf) == len(ref)
        m = min(8, len(ref))
        random.seed(11)
        want = random.sample(ref, m)
        random.seed(11)
        s2, p2, z2 = [x.cpu().numpy() for x in buf.sample(m)]
        for i in range(m):
            assert np.array_equal(s2[i], want[i][0]) and np.array_equal(p2[i], want[i][1].astype(np.float32)) and z2[i] == want[i][2]


def test_batched_training_loop_learns(az):
    """The loop as a whole must LEARN when games are played in batches.  (At 8x8 with 1024 games per
    batch the reference's controller, train.py:196-200, raised the learning rate to 0.023 while the KL
    of mini-batches of fresh tuples stayed small; after seven updates every ReLU of the net was dead
    and the loss sat at 1.0 + log(S); and a net whose 1x1 head convolutions start switched off by their
    biases cannot learn at all.  The pipeline therefore never raises the multiplier above 1 when
    play_batch_size > 1 and starts a net of its own with zero head biases.)  Here, small: both
    measures are in force, the loss falls, and the net still tells positions apart."""
    import torch
    from alphazero_gomoku_amd.train import TrainPipeline
    torch.manual_seed(0)
    tp = TrainPipeline(board_width=6, board_height=6, n_in_row=4, n_playout=48, play_batch_size=256, n_parallel=256,
                       batch_size=256, buffer_size=60000, game_batch_num=6, updates_per_batch=24, check_freq=0)
    assert tp.max_lr_multiplier == 1.0
    net = tp.policy_value_net.policy_value_net
    assert float(net.act_conv1.bias.abs().max()) == 0.0 and float(net.val_conv1.bias.abs().max()) == 0.0
    assert float(net.conv3.bias.abs().max()) > 0.0                           # only the head convolutions
    tp.run()
    h = tp.loss_history
    assert len(h) == 5 * 24 or len(h) == 6 * 24
    first, last = np.mean([l for l, _ in h[:8]]), np.mean([l for l, _ in h[-8:]])
    assert last < first - 0.3, (first, last)
    assert tp.lr_multiplier <= 1.0
    s, p, z = tp.data_buffer.take(tp.data_buffer.draw(256))
    probs, v = tp.policy_value_net.policy_value(s)
    assert float(probs.std(0).mean()) > 1e-3 and float(v.std()) > 1e-2      # not a constant function


def _native_dp_worker(rank, world, port, q):
    import os
    import torch
    import torch.distributed as dist
    import alphazero_gomoku_amd as pkg
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        net = pkg.PolicyValueNet(8, 8, device=0)
        net.load_numpy_params(helpers.make_params(8, seed=41))
        net.data_parallel = True
        s, pi, z = _dp_batch()
        sl = slice(rank, None, world)                          # this rank's share, as TrainPipeline.policy_update slices it
        out = [net.train_step(torch.as_tensor(s[sl], device="cuda:0"), torch.as_tensor(pi[sl], device="cuda:0"),
                              torch.as_tensor(z[sl], device="cuda:0"), 2e-3) for _ in range(2)]
        q.put((rank, out, [t.detach().cpu().numpy() for t in net.param_tensors()]))
    finally:
        dist.destroy_process_group()


def _dp_batch():
    rs = np.random.RandomState(77)
    s = (rs.rand(64, 4, 8, 8) < 0.3).astype(np.float32)
    pi = rs.dirichlet(np.full(64, 0.3), size=64).astype(np.float32)
    z = rs.choice([-1.0, 0.0, 1.0], size=64).astype(np.float32)
    return s, pi, z


def test_native_data_parallel_step_two_ranks_on_one_gpu(az):
    """Two ranks on GPU 0 (gloo, gradients staged through the host), each with its share of a mini-batch:
    azg_train_backward, ONE all_reduce of the flat gradient vector, azg_train_adam.  After two steps
    both ranks hold the same parameters, and they equal a single process's two steps on the whole
    mini-batch (policy_value_net_pytorch.py:117-148) to float32 rounding."""
    import os
    import torch
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 32500 + os.getpid() % 2000
    procs = [ctx.Process(target=_native_dp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    try:
        res = sorted((q.get(timeout=240) for _ in range(2)), key=lambda t: t[0])
    finally:
        for p in procs:
            p.join(timeout=60)
            if p.is_alive():
                p.kill()
    net = az.PolicyValueNet(8, 8, device=0)
    net.load_numpy_params(helpers.make_params(8, seed=41))
    s, pi, z = _dp_batch()
    for _ in range(2):
        net.train_step(torch.as_tensor(s, device="cuda:0"), torch.as_tensor(pi, device="cuda:0"), torch.as_tensor(z, device="cuda:0"), 2e-3)
    want = [t.detach().cpu().numpy() for t in net.param_tensors()]
    for a, b, w in zip(res[0][2], res[1][2], want):
        assert np.array_equal(a, b)
        np.testing.assert_allclose(a, w, rtol=0, atol=3e-6)


def test_training_batch_beyond_the_32_bit_offsets_is_refused(az):
    """The training GEMMs address activations with 32-bit byte offsets: a batch whose widest activation
    array (batch x cells x 128 floats) would reach 2 GiB is refused with a clear error, nothing is launched."""
    import torch
    net = az.PolicyValueNet(8, 8)
    too_many = (1 << 31) // (512 * 64)
    with pytest.raises(Exception, match="too large"):
        net.policy_value_in_training(torch.zeros((too_many, 4, 8, 8), device=net.device))
    probs, vals = net.policy_value_in_training(torch.zeros((3, 4, 8, 8), device=net.device))     # the handle still works
    assert probs.shape == (3, 64) and bool(torch.isfinite(probs).all())


def test_train_epoch_equals_step_then_forward(az):
    """azg_train_epoch (one call, one wait) against azg_train_step followed by azg_train_forward with the KL
    number: the same parameters, loss, entropy, probabilities, values and KL, bit for bit, over three epochs."""
    import torch
    rs = np.random.RandomState(12)
    s = torch.as_tensor((rs.rand(96, 4, 8, 8) < 0.3).astype(np.float32), device="cuda:0")
    pi = torch.as_tensor(rs.dirichlet(np.full(64, 0.3), size=96).astype(np.float32), device="cuda:0")
    z = torch.as_tensor(rs.choice([-1.0, 0.0, 1.0], size=96).astype(np.float32), device="cuda:0")
    nets = [az.PolicyValueNet(8, 8) for _ in range(2)]
    for net in nets:
        net.load_numpy_params(helpers.make_params(8, seed=77))
    old, _ = nets[0].policy_value_in_training(s)
    for epoch in range(3):
        loss_a, ent_a, probs_a, vals_a, kl_a = nets[0].train_epoch(s[:64], pi[:64], z[:64], 2e-3, s, old)
        loss_b, ent_b = nets[1].train_step(s[:64], pi[:64], z[:64], 2e-3)
        probs_b, vals_b, kl_b = nets[1].policy_value_in_training(s, old)
        assert (loss_a, ent_a, kl_a) == (loss_b, ent_b, kl_b), epoch
        assert torch.equal(probs_a, probs_b) and torch.equal(vals_a, vals_b)
        for a, b in zip(nets[0].param_tensors(), nets[1].param_tensors()):
            assert torch.equal(a, b)
