"""CPU-only checks of the host side: the C-ABI library loads and exports every symbol the
header declares, the host Board mirrors the reference's rules (golden vectors), and the
product refuses to run without a GPU instead of falling back to anything."""
import os
import re

import numpy as np
import pytest

import helpers

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    import alphazero_gomoku_amd as az
    hdr = open(os.path.join(ROOT, "include", "azgomoku.h")).read()
    declared = sorted(set(re.findall(r"\b(azg_[a-z_]+)\s*\(", hdr)))
    assert declared, "no declarations found"
    lib = az._lib.load()
    for name in declared:
        assert hasattr(lib, name), name
    assert sorted(az._lib.SYMBOLS) == declared


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import alphazero_gomoku_amd as az
    with pytest.raises(az._lib.AzgError):
        az.SelfPlayEngine(8, 5, 10, 1)
    with pytest.raises(RuntimeError):
        az.PolicyValueNet(8, 8)
    # an arbitrary callback is accepted (evaluated per playout), but the search itself still needs the GPU
    player = az.MCTSPlayer(lambda board: ([], 0.0), n_playout=2)
    board = az.Board(width=6, height=6, n_in_row=4)
    board.init_board()
    with pytest.raises(az._lib.AzgError):
        player.get_action(board)
    with pytest.raises(TypeError):
        az.MCTSPlayer(3.14)


def test_host_board_matches_reference_rules():
    import alphazero_gomoku_amd as az
    d = np.load(os.path.join(helpers.GOLDEN, "g1_rules.npz"))
    meta, moves, ends, winners, states = d["meta"], d["moves"], d["ends"], d["winners"], d["states"]
    mo = so = 0
    for (n, k, ln) in meta:
        b = az.Board(width=int(n), height=int(n), n_in_row=int(k))
        b.init_board()
        mid = None
        for t in range(ln):
            if t == ln // 2:
                mid = b.current_state()
            b.do_move(int(moves[mo + t]))
            assert b.game_end() == (bool(ends[mo + t]), int(winners[mo + t])), (n, k, t)
            assert len(b.availables) == n * n - t - 1 and b.last_move == moves[mo + t]
        if mid is None:
            mid = b.current_state()
        exp = states[so:so + 8 * n * n].reshape(2, 4, n, n)
        assert np.array_equal(mid, exp[0]) and np.array_equal(b.current_state(), exp[1])
        mo += ln
        so += 8 * n * n


def test_host_board_move_target_matches_oracle(oracle):
    """Board.get_move_target (game.py:35-58) of the host mirror against the oracle's restatement
    (itself pinned by the reference's recorded rollout searches) along random games."""
    import alphazero_gomoku_amd as az
    for n, k, seed in ((6, 4, 1), (8, 5, 2), (15, 5, 3), (16, 5, 4)):
        rs = np.random.RandomState(seed)
        b = az.Board(width=n, height=n, n_in_row=k)
        b.init_board()
        ob = oracle.OracleBoard(n, n, k)
        ob.init_board()
        for m in rs.permutation(n * n)[:40]:
            b.get_move_target()
            assert b.move_target == ob.move_target()
            if b.game_end()[0]:
                break
            b.do_move(int(m))
            ob.do_move(int(m))


def test_host_board_errors_and_helpers():
    import alphazero_gomoku_amd as az
    b = az.Board(width=3, height=3, n_in_row=4)
    with pytest.raises(Exception):
        b.init_board()
    b = az.Board(width=6, height=6, n_in_row=4)
    b.init_board(start_player=1)
    assert b.current_player == 2 and b.get_current_player() == 2
    b.do_move(7)
    with pytest.raises(ValueError):
        b.do_move(7)
    assert b.move_to_location(7) == [1, 1] and b.location_to_move([1, 1]) == 7
    assert b.location_to_move([9, 9]) == -1 and b.location_to_move([1]) == -1
    assert b.states == {7: 2} and b.current_player == 1


def _gather_worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    from alphazero_gomoku_amd import dist as azdist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n = 3 + 2 * rank                      # ragged: 3 and 5 samples
    first, stride, count = azdist.shard_game_ids(rank, world, 7)
    t = {"z": torch.arange(n, dtype=torch.float32) + 100 * rank,
         "pi": torch.full((n, 4), float(rank), dtype=torch.float64),
         "game_id": torch.arange(n, dtype=torch.int64) * stride + first}
    out = azdist.gather_samples(t)
    q.put((rank, out["z"].tolist(), out["pi"].shape, out["game_id"].tolist(), (first, stride, count)))
    dist.destroy_process_group()


def test_sample_gather_two_ranks_gloo():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_gather_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
    exp_z = [0.0, 1.0, 2.0, 100.0, 101.0, 102.0, 103.0, 104.0]
    for rank, z, pishape, gid, shard in res:
        assert z == exp_z and tuple(pishape) == (8, 4)
        assert gid == [0, 2, 4, 1, 3, 5, 7, 9]
    assert res[0][4] == (0, 2, 4) and res[1][4] == (1, 2, 3)


def test_replay_buffer_matches_deque_and_random_sample():
    """ReplayBuffer against collections.deque(maxlen) + random.sample (train.py:82,155,167)."""
    import collections
    import random
    from alphazero_gomoku_amd.train_data import ReplayBuffer
    n = 6
    rng = np.random.RandomState(3)
    buf, ref = ReplayBuffer(50, n), collections.deque(maxlen=50)
    for _ in range(7):
        k = int(rng.randint(1, 20))
        st = rng.rand(k, 4, n, n).astype(np.float32)
        pi = rng.rand(k, n * n)
        z = rng.choice([-1.0, 1.0], size=k).astype(np.float32)
        buf.extend(st, pi, z)
        ref.extend([(st[i], pi[i], z[i]) for i in range(k)])
        assert len(buf) == len(ref)
        m = min(8, len(ref))
        random.seed(11)
        want = random.sample(ref, m)
        random.seed(11)
        s2, p2, z2 = buf.sample(m)
        for i in range(m):
            assert np.array_equal(s2[i], want[i][0]) and np.array_equal(p2[i], want[i][1]) and z2[i] == want[i][2]


def _train_batch(n):
    rng = np.random.RandomState(5)
    st = (rng.rand(16, 4, n, n) < 0.3).astype(np.float32)
    pi = rng.rand(16, n * n).astype(np.float32)
    pi /= pi.sum(axis=1, keepdims=True)
    z = rng.choice([-1.0, 1.0], size=16).astype(np.float32)
    return st, pi, z


def _train_once(net, opt, st, pi, z, reduce_fn=None):
    """One optimiser step with the loss of policy_value_net_pytorch.py:117-148."""
    import torch
    import torch.nn.functional as F
    opt.zero_grad()
    logp, v = net(torch.from_numpy(st))
    loss = F.mse_loss(v.view(-1), torch.from_numpy(z)) - torch.mean(torch.sum(torch.from_numpy(pi) * logp, 1))
    loss.backward()
    if reduce_fn:
        reduce_fn(net)
    opt.step()
    return [p.detach().clone() for p in net.parameters()]


def _dp_worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    from alphazero_gomoku_amd import dist as azdist
    from alphazero_gomoku_amd.net import Net
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    net = Net(6, 6)
    opt = torch.optim.Adam(net.parameters(), lr=2e-3, weight_decay=1e-4)
    st, pi, z = _train_batch(6)
    sl = slice(rank * 8, rank * 8 + 8)                     # each rank: its half of the mini-batch
    out = _train_once(net, opt, st[sl], pi[sl], z[sl], azdist.allreduce_gradients)
    q.put((rank, [p.numpy() for p in out]))
    dist.destroy_process_group()


def test_data_parallel_train_step_two_ranks_gloo():
    """Two ranks, half a mini-batch each, gradients averaged with one flat all_reduce: the
    parameters after the step equal a single-process step on the whole mini-batch."""
    import torch
    import torch.multiprocessing as mp
    from alphazero_gomoku_amd.net import Net
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=180) for _ in range(2)), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
    torch.manual_seed(0)
    net = Net(6, 6)
    opt = torch.optim.Adam(net.parameters(), lr=2e-3, weight_decay=1e-4)
    want = _train_once(net, opt, *_train_batch(6))
    for a, b, w in zip(res[0][1], res[1][1], want):
        assert np.array_equal(a, b)
        np.testing.assert_allclose(a, w.numpy(), rtol=0, atol=2e-6)


def test_policy_update_schedule_matches_reference():
    """TrainPipeline.policy_update against the reference's own method (train.py:161-214, run by
    tools/gen_golden_r2.py on scripted nets): for every scripted sequence of KL values the same
    number of train steps at the same learning rates, the same lr_multiplier afterwards, the same
    return value."""
    import collections
    import random
    from alphazero_gomoku_amd.train import TrainPipeline
    from alphazero_gomoku_amd.train_data import ReplayBuffer
    scen = helpers.load_golden("g9_train.npz")["policy_update"]
    assert len(scen) >= 6

    class Scripted:
        def __init__(self, tables):
            self.tables, self.calls, self.lrs = tables, 0, []

        def policy_value(self, state_batch):
            t = self.tables[min(self.calls, len(self.tables) - 1)]
            self.calls += 1
            return t, np.linspace(-0.5, 0.5, len(state_batch)).reshape(-1, 1)

        def train_step(self, s, p, w, lr):
            self.lrs.append(lr)
            return 1.0 + len(self.lrs), 2.0
    for sc in scen:
        nb, S = sc["tables"].shape[1], sc["tables"].shape[2]
        tp = TrainPipeline.__new__(TrainPipeline)
        tp.batch_size, tp.epochs, tp.learn_rate, tp.kl_targ = nb, 5, 2e-3, 0.02
        tp.lr_multiplier = sc["lr_multiplier_before"]
        tp.rank, tp.world, tp.policy_updates = 0, 1, 0
        tp.max_lr_multiplier = 10.0                                   # the reference's ceiling (train.py:199)
        tp.data_buffer = ReplayBuffer(100, 3, rng=random.Random(3))
        tp.data_buffer.extend(np.zeros((nb, 4, 3, 3), dtype=np.float32), np.ones((nb, S)) / S,
                              np.array([float((-1) ** i) for i in range(nb)], dtype=np.float32))
        tp.policy_value_net = Scripted(list(sc["tables"]))
        loss, ent = tp.policy_update()
        assert len(tp.policy_value_net.lrs) == sc["train_steps"], sc["name"]
        np.testing.assert_allclose(tp.policy_value_net.lrs, sc["lrs"], rtol=1e-15)
        assert tp.lr_multiplier == sc["lr_multiplier_after"], sc["name"]
        assert (loss, ent) == (sc["loss"], sc["entropy"]), sc["name"]
        assert abs(tp.last_kl - sc["kls"][sc["train_steps"] - 1]) < 1e-15


def test_lr_multiplier_ceiling():
    """The multiplier rises (x1.5) only while below max_lr_multiplier: 10 in the reference (so it can
    reach 11.39), 1 by default when games are played in batches (it starts at 1 and can only fall)."""
    import random
    from alphazero_gomoku_amd.train import TrainPipeline
    from alphazero_gomoku_amd.train_data import ReplayBuffer

    class Still:                                                      # a net whose policy never moves: KL = 0
        def policy_value(self, state_batch):
            n = len(state_batch)
            return np.full((n, 9), 1.0 / 9), np.zeros((n, 1))

        def train_step(self, s, p, w, lr):
            return 1.0, 2.0
    for ceiling, top in ((10.0, 1.5 ** 6), (2.0, 2.25), (1.0, 1.0)):
        tp = TrainPipeline.__new__(TrainPipeline)
        tp.batch_size, tp.epochs, tp.learn_rate, tp.kl_targ, tp.lr_multiplier = 8, 5, 2e-3, 0.02, 1.0
        tp.rank, tp.world, tp.policy_updates, tp.max_lr_multiplier = 0, 1, 0, ceiling
        tp.data_buffer = ReplayBuffer(100, 3, rng=random.Random(3))
        tp.data_buffer.extend(np.zeros((16, 4, 3, 3), dtype=np.float32), np.ones((16, 9)) / 9, np.zeros(16, dtype=np.float32))
        tp.policy_value_net = Still()
        for _ in range(12):
            tp.policy_update()
        assert abs(tp.lr_multiplier - top) < 1e-12, (ceiling, tp.lr_multiplier)


def test_controller_defaults_are_the_references_with_one_game_per_batch():
    """play_batch_size 1 (the reference's configuration): ceiling 10 (train.py:199) and an untouched
    initialisation; batched play deviates on purpose and says so; explicit arguments win."""
    from alphazero_gomoku_amd.train import TrainPipeline
    assert TrainPipeline.controller_defaults(1) == (10.0, False)
    assert TrainPipeline.controller_defaults(0) == (10.0, False)
    assert TrainPipeline.controller_defaults(1024) == (1.0, True)
    assert TrainPipeline.controller_defaults(1024, max_lr_multiplier=10, zero_head_biases=False) == (10.0, False)
    assert TrainPipeline.controller_defaults(1, max_lr_multiplier=2, zero_head_biases=True) == (2.0, True)


def test_train_step_matches_reference_on_cpu():
    """train_step_tensors (the arithmetic of PolicyValueNet.train_step) on the CPU against the
    reference's PolicyValueNet.train_step (policy_value_net_pytorch.py:117-148, run by
    tools/gen_golden_r2.py): parameters after one and after three Adam steps at changing learning
    rates, loss and entropy of every step."""
    import torch
    from alphazero_gomoku_amd.net import Net, PARAM_ORDER, train_step_tensors
    d = helpers.load_golden("g9_train.npz")
    n, k, seed = [int(x) for x in d["spec"]]
    net = Net(n, n)
    sd = net.state_dict()
    for key, p in zip(PARAM_ORDER, helpers.make_params(n, seed=seed)):
        sd[key].copy_(torch.from_numpy(p.copy()))
    opt = torch.optim.Adam(net.parameters(), weight_decay=1e-4)
    s, p, z = (torch.from_numpy(d[key].astype(np.float32)) for key in ("states", "pi", "z"))
    for step, lr in enumerate(d["lrs"]):
        loss, ent = train_step_tensors(net, opt, s, p, z, float(lr))
        assert abs(float(loss) - d["loss"][step]) < 1e-6 and abs(float(ent) - d["entropy"][step]) < 1e-6
        if step in (0, 2):
            for i, key in enumerate(PARAM_ORDER):
                np.testing.assert_allclose(net.state_dict()[key].numpy(), d["params_after_%d_%d" % (step + 1, i)],
                                           rtol=0, atol=1e-7, err_msg="%s after step %d" % (key, step + 1))


def _packed_worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    from alphazero_gomoku_amd import dist as azdist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n, S = 2 + 3 * rank, 36                                # ragged: 2 and 5 tuples
    g = torch.Generator().manual_seed(rank)
    t = {"stones": torch.randint(-2 ** 62, 2 ** 62, (n, 2, 4), generator=g, dtype=torch.int64),
         "last": torch.arange(n, dtype=torch.int32) - 1, "ply": torch.arange(n, dtype=torch.int32) + 10 * rank,
         "z": torch.tensor([(-1.0) ** i for i in range(n)]), "game_id": torch.arange(n, dtype=torch.int64) * world + rank,
         "pi": torch.rand((n, S), generator=g, dtype=torch.float64)}
    rows = azdist.pack_samples(t)
    back = azdist.unpack_samples(rows, S)
    ok = all(torch.equal(back[key], t[key]) for key in t) and tuple(rows.shape) == (n, 84 + 8 * S)
    out = azdist.gather_packed(t, S)
    q.put((rank, ok, {key: v.numpy() for key, v in out.items()}, {key: v.numpy() for key, v in t.items()}))
    dist.destroy_process_group()


def test_packed_gather_two_ranks_gloo():
    """Training tuples travel as one byte matrix per rank (stones, last move, ply, z, game id, pi):
    pack / unpack is lossless, and both ranks end up with both ranks' tuples, rank after rank."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + os.getpid() % 2000
    procs = [ctx.Process(target=_packed_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=180) for _ in range(2)), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
    assert res[0][1] and res[1][1]
    for key in res[0][3]:
        want = np.concatenate([res[0][3][key], res[1][3][key]])
        assert np.array_equal(res[0][2][key], want) and np.array_equal(res[1][2][key], want), key


def test_pickled_model_files_are_read_without_unpickling(tmp_path):
    """model_files.read_pickled_arrays on files this test writes in the format of the reference's
    Theano / numpy nets (policy_value_net.py:113-121: pickle.dump(list of arrays, protocol=2)) and in
    Python 3's default protocol: the arrays come back bit for bit; a file whose pickle program would
    call something, hold other data or lie about its sizes is refused and nothing from it runs."""
    import pickle
    from alphazero_gomoku_amd import model_files
    params = helpers.torch_to_theano_layout(helpers.make_params(6, seed=5))
    for proto in (0, 1, 2, 3, 4):
        f = tmp_path / ("m%d.model" % proto)
        with open(f, "wb") as fh:
            pickle.dump([np.asarray(p) for p in params], fh, protocol=proto)
        assert model_files.looks_like_pickle(f) or proto < 2
        got = model_files.read_pickled_arrays(f)
        assert len(got) == 16
        for a, b in zip(got, params):
            assert a.dtype == np.float32 and a.shape == np.asarray(b).shape and np.array_equal(a, b)
    # float64 arrays (what a numpy net trained in double would hold) come back as float32
    f = tmp_path / "d.model"
    with open(f, "wb") as fh:
        pickle.dump([np.arange(6, dtype=np.float64).reshape(2, 3) / 3], fh, protocol=2)
    assert np.array_equal(model_files.read_pickled_arrays(f)[0], (np.arange(6, dtype=np.float64).reshape(2, 3) / 3).astype(np.float32))
    # a pickle that would run a command when unpickled: refused, and the command has not run
    marker = tmp_path / "ran"

    class Evil(object):
        def __reduce__(self):
            return (os.system, ("touch %s" % marker,))
    for payload in ([Evil()], Evil(), [np.zeros(3, dtype=np.float32), Evil()]):
        f = tmp_path / "evil.model"
        with open(f, "wb") as fh:
            pickle.dump(payload, fh, protocol=2)
        with pytest.raises(ValueError):
            model_files.read_pickled_arrays(f)
    assert not marker.exists()
    # other refusals: not a list, integer arrays, object arrays, a dict, truncated file, no pickle at all
    for i, payload in enumerate((np.zeros(3, dtype=np.float32), [np.zeros(3, dtype=np.int32)], [np.array([None, 1], dtype=object)],
                                 {"a": np.zeros(2, dtype=np.float32)}, [1.5, 2.5], [[np.zeros(2, dtype=np.float32)]])):
        f = tmp_path / ("bad%d.model" % i)
        with open(f, "wb") as fh:
            pickle.dump(payload, fh, protocol=2)
        with pytest.raises(ValueError):
            model_files.read_pickled_arrays(f)
    good = tmp_path / "m2.model"
    blob = open(good, "rb").read()
    (tmp_path / "cut.model").write_bytes(blob[: len(blob) // 2])
    (tmp_path / "text.model").write_bytes(b"not a pickle at all")
    for name in ("cut.model", "text.model"):
        with pytest.raises(ValueError):
            model_files.read_pickled_arrays(tmp_path / name)


# ---- bench.py --gpus N without a launcher -----------------------------------------------------------
_STUB_RANK = r'''
import json, os, sys, time
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
mode = sys.argv[1]
print("noise from rank %d" % rank)                      # must never reach the caller's stdout as the result
if mode == "fail" and rank == 1:
    sys.exit(7)
if mode == "hang" and rank == 1:
    time.sleep(600)
if mode == "fail" and rank == 0:
    time.sleep(600)                                      # must be ended by the launcher, not waited for
if rank == 0:
    print(json.dumps({"n_gpus": world, "local_rank": os.environ["LOCAL_RANK"], "addr": os.environ["MASTER_ADDR"],
                      "port": int(os.environ["MASTER_PORT"]), "argv": sys.argv[2:]}))
'''


def _bench_module():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_bench_starts_its_own_ranks(tmp_path, capfd):
    """`python bench.py --gpus N` with no launcher starts N rank processes with the rendezvous
    variables set, relays rank 0's JSON line alone, and fails when a rank fails or hangs."""
    import json
    import sys
    import time
    bench = _bench_module()
    stub = tmp_path / "stub_rank.py"
    stub.write_text(_STUB_RANK)
    rc, line = bench.launch_ranks(3, ["ok", "--steps", "3"], child=[sys.executable, str(stub)], timeout=60)
    assert rc == 0
    rec = json.loads(line)
    assert rec["n_gpus"] == 3 and rec["local_rank"] == "0" and rec["addr"] == "127.0.0.1" and rec["argv"] == ["--steps", "3"]
    assert rec["port"] > 0
    t0 = time.time()
    rc, line = bench.launch_ranks(2, ["fail"], child=[sys.executable, str(stub)], timeout=60)
    assert rc == 7 and line is None and time.time() - t0 < 30          # rank 0 was ended, not waited for
    t0 = time.time()
    rc, line = bench.launch_ranks(2, ["hang"], child=[sys.executable, str(stub)], timeout=2)
    assert rc == 124 and time.time() - t0 < 30
    capfd.readouterr()


def test_bench_refuses_a_gpu_count_the_launcher_did_not_start():
    """Under a launcher --gpus must equal WORLD_SIZE: a mismatch is an error, never a line for fewer ranks."""
    import subprocess
    import sys
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 2 and "WORLD_SIZE=2" in r.stderr and r.stdout.strip() == ""


def test_bench_with_more_gpus_than_the_machine_has_fails_loudly():
    """No launcher, --gpus 2 on a machine without two GPUs: the ranks are started, find no device and
    the whole run exits non-zero without a result line."""
    import subprocess
    import sys
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("two GPUs present")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--launch-timeout", "240"], capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and r.stdout.strip() == "" and "needs GPU" in r.stderr


def test_bench_reads_the_trained_fixture_net_in_the_reference_layout():
    """bench.py --model / --weights: the .npz fixture holds the reference's bundled nets as its model files
    store them (policy_value_net_numpy.py:103-113: kernels rotated by 180 degrees, matrices as (in, out));
    the bench hands the engine PyTorch-layout tensors."""
    import sys
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g10_trained.npz")
    got = bench.trained_params(path, 8, torch.device("cpu"))
    assert bench.trained_params(path, 7, torch.device("cpu")) is None
    with np.load(path) as d:
        raw = [d["params_8_%d" % i] for i in range(16)]
    assert len(got) == 16
    assert tuple(got[8].shape) == (64, 256) and raw[8].shape == (256, 64)          # act_fc1: (out, in) from (in, out)
    assert np.array_equal(got[8].numpy(), raw[8].T)
    assert np.array_equal(got[0].numpy(), raw[0][:, :, ::-1, ::-1])
    assert np.array_equal(got[1].numpy(), raw[1])
