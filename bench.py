#!/usr/bin/env python3
"""Self-play throughput of the HIP engine: completed 8x8 / 5-in-row / n_playout=400 games per
second with 1024 games in flight per GPU, random-init net, synthetic start positions.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

With N > 1 and no launcher (no RANK in the environment) this program starts its N rank processes
itself -- fresh interpreters, one per GPU, started before anything here has touched the GPU --
relays rank 0's line and fails if any rank fails.  Under a launcher --gpus must equal WORLD_SIZE.

A step is one move for every game slot: n_playout playouts per game inside one launch of the
fused search kernel (select / leaf CNN / expand / backup), then sampling the move, recording
the (state, pi) sample, playing the move, re-rooting the tree and restarting finished games.
Rank 0 prints ONE JSON line."""
import argparse
import json
import os
import sys
import time

os.environ["NCCL_DEBUG"] = "WARN"       # before torch / RCCL load: keeps RCCL's banner off stdout (ONE JSON line)

import numpy as np  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

FLOP_PER_EVAL = {6: 6.79e6, 8: 12.09e6, 15: 42.80e6}    # 188416*S + 8*S*S + 128, 2 flop / MAC
PEAK_TFLOPS = {"f16x3": 2500.0, "f32": 157.3}            # dense MFMA peaks of MI355X: f16 inputs / f32 inputs
# the reference's own Python path (mcts_alphaZero.py + policy_value_net_numpy.py), one thread, measured
# where the reference could be imported (the build container, not the GPU box): games/s at the per-game settings
REFERENCE_PYTHON_GAMES_PER_S = {(8, 400): 0.0290, (6, 200): 0.1165}


def pmc_traffic(cnn, n, playouts, games):
    """HBM bytes per search (all its launches) from the newest committed rocprofv3 --pmc summary of this
    workload (profiles/r*_pmc_search_kernel.json for the default 8x8 workload, ..._c4.json for
    15x15/800/512: FETCH_SIZE and WRITE_SIZE from separate passes, in KiB).  None for any other
    workload or when no summary travels with the tree."""
    import glob
    tag = {("f16x3", 8, 400, 1024): "", ("f16x3", 15, 800, 512): "_c4"}.get((cnn, n, playouts, games))
    if tag is None:
        return None, None
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_search_kernel%s.json" % tag)),
                   key=lambda f: (len(os.path.basename(f).split("_")[0]), os.path.basename(f)))
    for f in reversed(files):
        try:
            with open(f) as fh:
                d = json.load(fh)
            # (the summary's figures are per kernel launch; a search may be several launches)
            return ((d["FETCH_SIZE"]["mean_of_last_4"] + d["WRITE_SIZE"]["mean_of_last_4"]) * 1024.0
                    * d.get("_kernel_launches_per_search", 1)), os.path.basename(f)
        except (OSError, KeyError, ValueError):
            continue
    return None, None


def random_init_params(n, device):
    """PyTorch default initialisation of the reference topology under manual_seed(0)."""
    import torch
    from alphazero_gomoku_amd.net import Net, PARAM_ORDER
    torch.manual_seed(0)
    net = Net(n, n)
    sd = net.state_dict()
    return [sd[k].to(device) for k in PARAM_ORDER]


def cpu_baseline(n, k, n_playout, params_np, plies_per_game=None):
    """The CPU oracle (C port of the reference's mcts_alphaZero + numpy-net path) timed on the
    host cores: one whole game per thread, float32 net.  Workloads whose whole games would take
    minutes on the host (large boards, many playouts) are sampled by searches instead: every thread
    runs two successive moves' searches of one game, and games/s follows from the playout rate and
    the mean game length of the GPU run."""
    from oracle import az_oracle
    az_oracle.build()
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = os.cpu_count() or 1
    # a one-GPU share of the host is 16 cores; os.cpu_count() shows the whole machine
    threads = max(1, min(16, usable))
    net = az_oracle.OracleNet(n, n, params_np, precise=False)
    if n * n * n_playout > 2 * 64 * 400:
        from concurrent.futures import ThreadPoolExecutor

        def searches(seed):
            ob = az_oracle.OracleBoard(n, n, k)
            ob.init_board()
            t = az_oracle.OracleMCTS("net", c_puct=5, n_playout=n_playout, net=net)
            for _ in range(2):
                acts, vis, _, _, _ = t.get_move_probs(ob, temp=1.0)
                mv = int(acts[(int(np.argmax(vis)) + seed) % len(acts)])
                t.update_with_move(mv)
                ob.do_move(mv)
            return 2 * n_playout
        t0 = time.time()
        with ThreadPoolExecutor(threads) as ex:                      # the oracle's C code runs without the GIL
            done = sum(ex.map(searches, range(threads)))
        dt = time.time() - t0
        rate = done / dt
        return {"value": rate / (n_playout * plies_per_game) if plies_per_game else None, "unit": "games/s", "cores": threads,
                "kind": "port", "playouts_per_s": rate, "nproc": os.cpu_count(), "cores_usable": usable,
                "sample": "%d searches of %d playouts (two successive moves of one game per thread, tree re-used) in %.1f s; "
                          "games/s = playout rate / (n_playout x %.1f plies per game of the GPU run); C restatement of the "
                          "reference's Python path, float32 net" % (2 * threads, n_playout, dt, plies_per_game or float("nan"))}
    t0 = time.time()
    total_plies, plies, _ = az_oracle.self_play_many(n, k, net, 5.0, n_playout, 1.0, 0, threads, threads)
    dt = time.time() - t0
    # the same port on one core: one game (seed 0)
    t1 = time.time()
    plies1, _, _ = az_oracle.self_play_many(n, k, net, 5.0, n_playout, 1.0, 0, 1, 1)
    dt1 = time.time() - t1
    return {"value": threads / dt, "unit": "games/s", "cores": threads, "kind": "port",
            "sample": "%d games (one per thread, seeds 0..%d), %d plies, %d playouts in %.1f s; "
                      "C restatement of the reference's Python path, float32 net"
                      % (threads, threads - 1, total_plies, total_plies * n_playout, dt),
            "playouts_per_s": total_plies * n_playout / dt,
            "nproc": os.cpu_count(), "cores_usable": usable,
            "single_core": {"value": 1.0 / dt1, "unit": "games/s", "cores": 1,
                            "sample": "1 game (seed 0), %d plies in %.1f s" % (plies1, dt1)},
            "reference_python_other_box": {"value": REFERENCE_PYTHON_GAMES_PER_S.get((n, n_playout)), "unit": "games/s",
                                           "cores": 1, "sample": "the reference's Python (mcts_alphaZero.py + numpy net), 3 games, "
                                                                 "measured on the 8-vCPU build container, not on this box"}}


def visit_l1_vs_cpu(az, az_oracle, n, k, n_playout, params, params_np, samples, device, cnn_mode, count=8):
    """Per-move visit-count L1 between the GPU search and the CPU oracle's search (its own float64
    forward pass on the same float32 weights), both from a fresh root on positions taken from the
    self-play games of the timed run: sum_a |N_gpu(a) - N_cpu(a)| / sum_a N_cpu(a)."""
    st = samples["states"][:4096].to("cpu").numpy()
    if not len(st):
        return None
    pick = np.linspace(0, len(st) - 1, count).astype(int)
    eng = az.SelfPlayEngine(n, k, n_playout, count, 5.0, device=device, cnn_mode=cnn_mode)
    eng.set_weights(params)
    eng.reset_games()
    net = az_oracle.OracleNet(n, n, params_np, precise=True)
    boards = []
    for slot, i in enumerate(pick):
        pl = st[i][:, ::-1, :]                                  # plane row r shows board row n-1-r
        mine = [int(c) for c in np.flatnonzero(pl[0].reshape(-1) > 0)]
        theirs = [int(c) for c in np.flatnonzero(pl[1].reshape(-1) > 0)]
        lastc = [int(c) for c in np.flatnonzero(pl[2].reshape(-1) > 0)]
        last = lastc[0] if lastc else -1
        cur = 1 if pl[3][0][0] > 0 else 2                       # stone 1 opens: it is to move at even ply
        stones = {c: cur for c in mine}
        stones.update({c: 3 - cur for c in theirs})
        eng.set_position(slot, stones, last, cur)
        p1 = [c for c, p in stones.items() if p == 1 and c != last]
        p2 = [c for c, p in stones.items() if p == 2 and c != last]
        if last >= 0:
            (p1 if stones[last] == 1 else p2).append(last)
        ob = az_oracle.OracleBoard(n, n, k)
        ob.init_board()
        for t in range(len(p1) + len(p2)):
            ob.do_move(p1[t // 2] if t % 2 == 0 else p2[t // 2])
        boards.append(ob)
    eng.search()
    visits, _, _, _ = eng.root_stats()
    eng.close()
    l1 = []
    for slot, ob in enumerate(boards):
        t = az_oracle.OracleMCTS("net", c_puct=5, n_playout=n_playout, net=net)
        acts, ovis, _, _, _ = t.get_move_probs(ob, temp=1.0)
        cpu = np.zeros(n * n)
        cpu[acts] = ovis
        l1.append(float(np.abs(visits[slot] - cpu).sum() / max(1.0, cpu.sum())))
    return {"mean": float(np.mean(l1)), "max": float(np.max(l1)), "positions": int(count),
            "identical": int(sum(1 for x in l1 if x == 0.0)), "n_playout": n_playout,
            "note": "fresh-root searches from positions of the timed run's games; CPU side = oracle tree + its float64 net"}


def trained_params(path, n, device):
    """The 16 parameter tensors (state_dict order, PyTorch layout) of a trained net for an n x n board, or None
    when `path` holds none for that size.  An .npz with params_<n>_<0..15> holds the reference's own model
    files (mode/best_policy_*.model: Theano layout, kernels rotated, matrices as (in, out)), as the fixture
    tests/golden/g10_trained.npz does; anything else is read as a model file of PolicyValueNet."""
    import numpy as np
    import torch
    import alphazero_gomoku_amd as az
    if path.endswith(".npz"):
        from alphazero_gomoku_amd.net import theano_to_torch_layout
        with np.load(path) as d:
            if "params_%d_0" % n not in d.files:
                return None
            return [torch.from_numpy(a).to(device) for a in theano_to_torch_layout([d["params_%d_%d" % (n, i)] for i in range(16)])]
    net = az.PolicyValueNet(n, n, model_file=path, device=device.index or 0)
    return [t.detach().clone() for t in net.param_tensors()]


def _free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def launch_ranks(n, argv, child=None, timeout=1500.0, env=None):
    """Start n rank processes (RANK = LOCAL_RANK = 0..n-1, WORLD_SIZE = n, rendezvous on
    127.0.0.1 at a free port), wait for all of them and return (exit code, rank 0's JSON line or
    None).  `child` is the command every rank runs (default: this program in a fresh interpreter)
    followed by argv.  Rank 0's stdout is captured; the other ranks' stdout goes to stderr, so the
    caller's stdout carries nothing but the one line it prints itself.  The first rank that fails
    (or the time limit) ends the others: exactly the processes started here, by their PIDs.
    The caller must not have initialised the GPU: children are separate processes, never an exec
    of this one."""
    import subprocess
    import threading
    child = list(child) if child else [sys.executable, "-u", os.path.abspath(__file__)]
    base = dict(os.environ if env is None else env)
    base.update({"WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n), "MASTER_ADDR": "127.0.0.1",
                 "MASTER_PORT": str(_free_port())})
    procs = []
    for r in range(n):
        e = dict(base, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen(child + list(argv), env=e, stdout=subprocess.PIPE if r == 0 else sys.stderr,
                                      stderr=sys.stderr, text=(r == 0) or None))
    lines = []
    reader = threading.Thread(target=lambda: lines.extend(procs[0].stdout), daemon=True)
    reader.start()
    deadline = time.time() + timeout
    rc = 0
    pending = set(range(n))
    while pending and rc == 0:
        for r in sorted(pending):
            code = procs[r].poll()
            if code is not None:
                pending.discard(r)
                if code != 0:
                    print("bench.py: rank %d of %d exited with code %d" % (r, n, code), file=sys.stderr)
                    rc = code if 0 < code < 256 else 1
        if pending and rc == 0:
            if time.time() > deadline:
                print("bench.py: ranks %s still running after %.0f s" % (sorted(pending), timeout), file=sys.stderr)
                rc = 124
            else:
                time.sleep(0.05)
    for r in sorted(pending):                       # a failure or the time limit: end what was started here
        procs[r].terminate()
    for r in sorted(pending):
        try:
            procs[r].wait(timeout=20)
        except subprocess.TimeoutExpired:
            procs[r].kill()
            procs[r].wait()
    reader.join(timeout=10)
    line = None
    for text in lines:
        text = text.strip()
        if text.startswith("{"):
            try:
                json.loads(text)
                line = text
            except ValueError:
                pass
        elif text:
            print(text, file=sys.stderr)            # anything else a rank wrote to stdout is not the result
    if rc == 0 and line is None:
        print("bench.py: rank 0 printed no JSON line", file=sys.stderr)
        rc = 1
    return rc, line


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None,
                    help="ranks = GPUs of the run (default: WORLD_SIZE under a launcher, else 1); with N > 1 and no "
                         "launcher this program starts the N rank processes itself")
    ap.add_argument("--launch-timeout", type=float, default=1500.0,
                    help="seconds the self-started ranks of --gpus N may take before they are ended")
    ap.add_argument("--steps", type=int, default=96)
    ap.add_argument("--warmup", type=int, default=64)
    ap.add_argument("--games", type=int, default=1024, help="game slots per GPU")
    ap.add_argument("--board", type=int, default=8)
    ap.add_argument("--n-in-row", type=int, default=5)
    ap.add_argument("--playouts", type=int, default=400)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--preroll-max", type=int, default=600,
                    help="upper bound of the untimed pre-roll steps that bring the slot population to steady state")
    ap.add_argument("--host-refill", action="store_true",
                    help="restart finished slots from the host between steps instead of the pipelined azg_selfplay_* calls")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="every rank uses GPU 0 and the exchange goes over gloo: runs the sharded code path with several "
                         "processes on a one-GPU machine (RCCL refuses two ranks on one device); NOT a benchmark")
    ap.add_argument("--arena-nodes", type=int, default=0,
                    help="expanded nodes a game's tree arena holds (0 = the library's default, 16 searches' worth when memory allows)")
    ap.add_argument("--cnn", choices=["f16x3", "f32"], default="f16x3",
                    help="conv arithmetic: split-f16 MFMA (three products, f32 accumulate) or f32-input MFMA")
    ap.add_argument("--model", default=None,
                    help="weights of a TRAINED net for the second, shorter measurement reported as `trained_net` (deep, narrow "
                         "trees: the descent-bound regime): an .npz holding params_<board>_<0..15> in PyTorch's layout, or a "
                         "state_dict file; default tests/golden/g10_trained.npz when it holds this board size; 'none' skips it")
    ap.add_argument("--weights", default=None,
                    help="run the MAIN measurement with these weights instead of the random-init net (same file forms as "
                         "--model; for profiles of the trained-net regime: `value` is then NOT the headline configuration)")
    ap.add_argument("--c4-steps", type=int, default=8,
                    help="timed move steps of the 15x15 / 800 playouts / 512 games measurement reported as `c4` after the "
                         "headline one (one GPU, default workload only; 0 skips it; adds ~20 s, mostly its pre-roll)")
    ap.add_argument("--trained-steps", type=int, default=48, help="timed move steps of the trained-net measurement")
    args = ap.parse_args()

    under_launcher = "RANK" in os.environ
    env_world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus is None:
        args.gpus = env_world if under_launcher else 1
    if args.gpus < 1:
        ap.error("--gpus must be at least 1")
    if under_launcher and args.gpus != env_world:
        print("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks" % (args.gpus, env_world), file=sys.stderr)
        sys.exit(2)
    if args.gpus > 1 and not under_launcher:
        # no launcher: be one.  Nothing in this process has touched the GPU (torch is not even imported).
        rc, line = launch_ranks(args.gpus, sys.argv[1:], timeout=args.launch_timeout)
        if line is not None and rc == 0:
            print(line)
        sys.exit(rc)

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    local_rank = 0 if args.rehearse_on_one_gpu else int(os.environ.get("LOCAL_RANK", "0"))
    world = env_world if under_launcher else 1
    use_dist = under_launcher
    if not args.rehearse_on_one_gpu and local_rank >= torch.cuda.device_count():
        print("bench.py: rank %d needs GPU %d, this machine shows %d" % (rank, local_rank, torch.cuda.device_count()), file=sys.stderr)
        sys.exit(3)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        torch.cuda.set_device(local_rank)
        # RCCL writes a banner (version, host name, library path) to stdout when its communicator
        # comes up; stdout must carry ONE JSON line, so fd 1 points at stderr until that is over
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        try:
            if args.rehearse_on_one_gpu:
                dist.init_process_group("gloo", rank=rank, world_size=world)
            else:
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
            dist.barrier()
            torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(saved_stdout, 1)
            os.close(saved_stdout)
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    import alphazero_gomoku_amd as az
    from alphazero_gomoku_amd import dist as azdist
    from alphazero_gomoku_amd.engine import SelfPlayRunner

    n, k = args.board, args.n_in_row
    params = random_init_params(n, dev)
    if args.weights:
        params = trained_params(args.weights, n, dev)
        if params is None:
            print("bench.py: %s holds no net for a %dx%d board" % (args.weights, n, n), file=sys.stderr)
            sys.exit(2)
    eng = az.SelfPlayEngine(n, k, args.playouts, args.games, 5.0, device=local_rank, arena_nodes=args.arena_nodes,
                            cnn_mode=az._lib.CNN_F16X3 if args.cnn == "f16x3" else az._lib.CNN_F32)
    eng.set_weights(params)
    drained = []
    # sharded runs exchange the tuples in their packed form (stone words instead of planes: one byte
    # buffer per rank); the pipelined runner delivers them that way
    packed = use_dist and not args.host_refill
    run = SelfPlayRunner(eng, temp=1.0, seed0=0, first_game_id=rank, game_id_stride=world,
                         total=None, on_samples=drained.append, device_samples=True, pipelined=not args.host_refill,
                         packed_samples=packed)
    run.start()

    def gather_all(chunks):
        # the (state, pi, z) gather of the sharded configuration: every rank ends up with every
        # rank's tuples (count all_gather + ONE padded all_gather of bytes, RCCL over xGMI), then
        # rebuilds the feature planes of all of them from the stone words
        if packed:
            keys = ("stones", "last", "ply", "z", "game_id", "pi")
            if chunks:
                mine = {key: torch.cat([d[key] for d in chunks]) for key in keys}
            else:
                mine = {"stones": torch.empty((0, 2, 4), dtype=torch.int64, device=dev), "last": torch.empty(0, dtype=torch.int32, device=dev),
                        "ply": torch.empty(0, dtype=torch.int32, device=dev), "z": torch.empty(0, device=dev),
                        "game_id": torch.empty(0, dtype=torch.int64, device=dev), "pi": torch.empty((0, n * n), dtype=torch.float64, device=dev)}
            allr = azdist.gather_packed(mine, n * n)
            total = int(allr["z"].shape[0])
            if total:
                states = eng.unpack_states(allr["stones"], allr["last"], allr["ply"])
                assert states.shape == (total, 4, n, n)
            return total
        if chunks:
            mine = {key: torch.cat([d[key] for d in chunks]) for key in ("states", "pi", "z", "game_id", "ply")}
        else:
            mine = {"states": torch.empty((0, 4, n, n), dtype=torch.float32, device=dev), "pi": torch.empty((0, n * n), dtype=torch.float64, device=dev),
                    "z": torch.empty(0, device=dev), "game_id": torch.empty(0, dtype=torch.int64, device=dev),
                    "ply": torch.empty(0, dtype=torch.int32, device=dev)}
        mine["states"] = mine["states"].to(torch.uint8)   # the planes are 0/1: exact, a quarter of the bytes on the links
        return int(azdist.gather_samples(mine)["z"].shape[0])

    # Untimed, declared pre-roll: all slots start together on empty boards, so right after the
    # start the games that end are the short ones of ONE cohort.  Steps run until every slot has
    # turned over at least twice and three mean game lengths have passed; from then on games end at
    # the steady rate slots / mean length per step, whatever --warmup and --steps are.
    preroll = 0
    while preroll < args.preroll_max:
        run.step()
        preroll += 1
        if run.games_delivered >= 2 * args.games and preroll >= 3.0 * run.plies_done / run.games_delivered:
            break
    for _ in range(args.warmup):
        run.step()
    if use_dist:
        gather_all(drained)          # warm-up of the exchange too: RCCL sets its channels up on first use
    drained.clear()
    eng.profile()                                   # reset the kernel timers
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    g0 = run.games_done
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run.step()
    gathered = 0
    if use_dist:
        gathered = gather_all(drained)                      # the tuples of the timed region, inside the timed region
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    dt = time.perf_counter() - t0
    run.finish()
    games = run.games_done - g0
    # mean game length over every game of the run (pre-roll included): whole games only
    all_games, plies = run.games_delivered, run.plies_done
    launches, kernel_ms, evals = eng.profile()
    host = eng.host_timing()

    stats = torch.tensor([dt, games, plies, evals, kernel_ms, launches, all_games, preroll, 1.0], dtype=torch.float64,
                         device="cpu" if args.rehearse_on_one_gpu else dev)
    ranks_seen, dt_min, dt_max = 1, dt, dt
    if use_dist:
        mx = stats.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        sm = stats.clone()
        dist.all_reduce(sm, op=dist.ReduceOp.SUM)
        mn = stats.clone()
        dist.all_reduce(mn, op=dist.ReduceOp.MIN)
        ranks_seen, dt_min, dt_max = int(round(float(sm[8]))), float(mn[0]), float(mx[0])   # a one from every rank of the group
        dt = float(mx[0])
        games, plies, evals, all_games = float(sm[1]), float(sm[2]), float(sm[3]), float(sm[6])
        kernel_ms, launches = float(sm[4]) / world, float(sm[5]) / world
        preroll = int(mx[7])
    if rank == 0:
        flop = FLOP_PER_EVAL.get(n, 188416.0 * n * n + 8.0 * n ** 4 + 128)
        # per-GPU dominant kernel: leaf-eval flops of one launch / its HIP-event duration
        evals_per_gpu = evals / world
        achieved = flop * evals_per_gpu / (kernel_ms * 1e-3) / 1e12 if kernel_ms > 0 else 0.0
        plies_per_game = plies / all_games if all_games else None
        traffic, traffic_src = pmc_traffic(args.cnn, n, args.playouts, args.games)
        out = {
            "metric": "completed self-play games/sec, %dx%d board n_playout=%d" % (n, n, args.playouts),
            "value": games / dt, "unit": "games/s", "n_gpus": world, "ranks_seen": ranks_seen, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "ms_per_step_rank_min_max": [dt_min / args.steps * 1e3, dt_max / args.steps * 1e3],
            "backend": ("gloo" if args.rehearse_on_one_gpu else "nccl (RCCL)") if use_dist else None,
            "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None,
            "dtype": "f16x3 (f32 operands split into two halves, three f16 MFMAs per product, f32 accumulate; tree in f64)" if args.cnn == "f16x3" else "f32",
            "data": "synthetic" + (" (REHEARSAL: all ranks share GPU 0, exchange over gloo -- not a measurement)" if args.rehearse_on_one_gpu else ""),
            "config": {"workload": "%dx%d board %d-in-row, n_playout=%d, %d parallel self-play games per MI355X, "
                                   "%s, empty-board starts, "
                                   "c_puct=5, temp=1, Dirichlet(0.1) mix 0.25, tree reuse, finished slots refilled"
                                   % (n, n, k, args.playouts, args.games,
                                      "TRAINED net %s (not the headline configuration)" % args.weights if args.weights
                                      else "random-init net (torch.manual_seed(0) default init)"),
                       "games_in_flight_per_gpu": args.games, "sharding": "game id mod n_gpus",
                       "step": "one move (n_playout playouts + sample/advance/re-root) for every slot"},
            "games_completed": games, "plies_per_game": plies_per_game,
            "preroll_steps": preroll, "games_before_timed_region": all_games - games,
            # self-check of `value`: in steady state games end at slots / (mean length) per step
            "value_from_plies": (args.games * world / (plies_per_game * dt / args.steps)) if plies_per_game else None,
            "playouts_per_s": evals / dt, "samples_gathered": gathered,
            "host_noise_ms_per_step": host[0] / host[1] if host[1] else None,
            "roofline": {"bound": "mfma", "kernel": ("search_dual_kernel (net waves: conv2 + conv3 + heads; tree waves: select, conv1, FC, expand, backup)"
                                                 if 9 <= n <= 15 and args.cnn == "f16x3" else
                                                 "search_kernel (fused select + leaf CNN + expand + backup)"),
                         "achieved": achieved, "peak": PEAK_TFLOPS[args.cnn], "unit": "TFLOP/s",
                         "frac": achieved / PEAK_TFLOPS[args.cnn],
                         "traffic": traffic, "traffic_source": traffic_src,
                         "traffic_note": "HBM bytes per launch, (FETCH_SIZE + WRITE_SIZE) * 1024 of separate rocprofv3 --pmc "
                                         "passes; the tree arrays move as 4-8 B per lane accesses, for which gfx950's FETCH_SIZE "
                                         "is uncalibrated (it halves 16 B per lane streams): the read share may be up to 2x larger",
                         "note": "achieved counts the ALGORITHMIC flops of the leaf CNN (flop_per_leaf_eval, 12.09 MFLOP at 8x8); "
                                 "the f16x3 path issues ~2.9x that many MFMA flops",
                         "launches": launches, "avg_launch_ms": kernel_ms / launches if launches else None,
                         "kernel_launches_per_search": eng.search_parts(),
                         "avg_kernel_launch_ms": (kernel_ms / launches / eng.search_parts()) if launches else None,
                         "launch_note": "`launches` counts searches (one per move step); a search is kernel_launches_per_search "
                                        "launches of the kernel back to back, timed together by one pair of HIP events; "
                                        "avg_kernel_launch_ms is what rocprofv3's kernel statistics show per call",
                         "flop_per_leaf_eval": flop, "leaf_evals_per_launch": evals_per_gpu / launches if launches else None},
        }
        # the tree / board side of the same launch against the HBM roof: algorithmic bytes per
        # playout (select reads, expand writes, backup, board copy, planes, net output) x playouts
        tree_bytes = {8: 3.7e3, 15: 14.4e3}.get(n)
        if tree_bytes and kernel_ms > 0:
            hbm = tree_bytes * (args.playouts * args.games) / (kernel_ms / launches * 1e-3) / 1e9
            out["roofline_tree_side"] = {"bound": "hbm", "achieved": hbm, "peak": 8000.0, "unit": "GB/s", "frac": hbm / 8000.0,
                                         "note": "same kernel; %.1f KB of tree/board traffic per playout, latency-bound "
                                                 "pointer chasing inside the fused launch" % (tree_bytes / 1e3)}
        # The same workload with a TRAINED net: sharp priors make deep, narrow trees, the regime in which
        # the descent (not the leaf CNN) bounds a launch.  Own pre-roll, own timed region, one GPU only.
        model = args.model or os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests", "golden", "g10_trained.npz")
        if world == 1 and model != "none" and not args.weights and os.path.exists(model):
            tparams = trained_params(model, n, dev)
            if tparams is not None:
                eng.set_weights(tparams)
                run2 = SelfPlayRunner(eng, temp=1.0, seed0=1 << 20, first_game_id=0, game_id_stride=1, total=None,
                                      on_samples=lambda chunk: None, device_samples=True, pipelined=not args.host_refill)
                run2.start()
                pre2 = 0
                while pre2 < args.preroll_max:
                    run2.step()
                    pre2 += 1
                    if run2.games_delivered >= 2 * args.games and pre2 >= 3.0 * run2.plies_done / run2.games_delivered:
                        break
                eng.profile()
                torch.cuda.synchronize()
                g2, t2 = run2.games_done, time.perf_counter()
                for _ in range(args.trained_steps):
                    run2.step()
                torch.cuda.synchronize()
                dt2 = time.perf_counter() - t2
                run2.finish()
                l2, k2, e2 = eng.profile()
                ach2 = flop * e2 / (k2 * 1e-3) / 1e12 if k2 > 0 else 0.0
                out["trained_net"] = {"model": os.path.relpath(model, os.path.dirname(os.path.abspath(__file__))),
                                      "steps": args.trained_steps, "preroll_steps": pre2,
                                      "ms_per_step": dt2 / args.trained_steps * 1e3, "value": (run2.games_done - g2) / dt2,
                                      "unit": "games/s", "avg_launch_ms": k2 / l2 if l2 else None,
                                      "plies_per_game": run2.plies_done / run2.games_delivered if run2.games_delivered else None,
                                      "roofline_frac": ach2 / PEAK_TFLOPS[args.cnn]}
                eng.set_weights(params)
        if world == 1 and not args.no_cpu_baseline:
            params_np = [p.cpu().numpy() for p in params]
            out["cpu_baseline"] = cpu_baseline(n, k, args.playouts, params_np, plies_per_game)
            if drained:
                from oracle import az_oracle
                out["visit_l1_vs_cpu"] = visit_l1_vs_cpu(az, az_oracle, n, k, args.playouts, params, params_np, drained[-1],
                                                         local_rank, az._lib.CNN_F16X3 if args.cnn == "f16x3" else az._lib.CNN_F32)
        # BASELINE configuration C4 (15x15 / 5 in a row / 800 playouts / 512 games, random-init net) beside the
        # headline: the two-wave-group kernel of the larger boards.  Own engine (the headline one is closed
        # first: its trees hold 20 GB), own pre-roll; few games END in a handful of 46 ms steps, so the
        # steady-state rate slots / (plies per game x seconds per step) is the figure, as value_from_plies above.
        default_workload = (n, k, args.playouts, args.games, args.cnn) == (8, 5, 400, 1024, "f16x3") and not args.weights
        if world == 1 and default_workload and args.c4_steps > 0:
            eng.close()
            eng = None
            n4, k4, po4, b4 = 15, 5, 800, 512
            eng4 = az.SelfPlayEngine(n4, k4, po4, b4, 5.0, device=local_rank)
            eng4.set_weights(random_init_params(n4, dev))
            run4 = SelfPlayRunner(eng4, temp=1.0, seed0=0, first_game_id=0, game_id_stride=1, total=None,
                                  on_samples=lambda chunk: None, device_samples=True, pipelined=True)
            run4.start()
            pre4 = 0
            while pre4 < 400:
                run4.step()
                pre4 += 1
                if run4.games_delivered >= 2 * b4 and pre4 >= 3.0 * run4.plies_done / run4.games_delivered:
                    break
            eng4.profile()
            torch.cuda.synchronize()
            g4, t4 = run4.games_done, time.perf_counter()
            for _ in range(args.c4_steps):
                run4.step()
            torch.cuda.synchronize()
            dt4 = time.perf_counter() - t4
            run4.finish()
            l4, k4ms, e4 = eng4.profile()
            ppg4 = run4.plies_done / run4.games_delivered if run4.games_delivered else None
            flop4 = FLOP_PER_EVAL[15]
            out["c4"] = {"workload": "15x15 board 5-in-row, n_playout=800, 512 parallel self-play games, random-init net",
                         "steps": args.c4_steps, "preroll_steps": pre4, "ms_per_step": dt4 / args.c4_steps * 1e3,
                         "avg_search_ms": k4ms / l4 if l4 else None, "kernel_launches_per_search": eng4.search_parts(),
                         "games_completed": run4.games_done - g4, "plies_per_game": ppg4,
                         "value_from_plies": b4 / (ppg4 * dt4 / args.c4_steps) if ppg4 else None, "unit": "games/s",
                         "roofline_frac": (flop4 * e4 / (k4ms * 1e-3) / 1e12 / PEAK_TFLOPS["f16x3"]) if k4ms > 0 else None}
            eng4.close()
        print(json.dumps(out))
    if eng is not None:
        eng.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
