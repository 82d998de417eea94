#!/usr/bin/env python3
"""Self-play throughput of the HIP engine: completed 8x8 / 5-in-row / n_playout=400 games per
second with 1024 games in flight per GPU, random-init net, synthetic start positions.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

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
# HBM bytes per search launch from rocprofv3 PMC passes (FETCH_SIZE + WRITE_SIZE, separate runs,
# profiles/r1_n_pmc_search_kernel.json); only valid for the default workload and build
PMC_TRAFFIC_BYTES = {("f16x3", 8, 400, 1024): (269291.4 + 570463.6) * 1024.0}


def random_init_params(n, device):
    """PyTorch default initialisation of the reference topology under manual_seed(0)."""
    import torch
    from alphazero_gomoku_amd.net import Net, PARAM_ORDER
    torch.manual_seed(0)
    net = Net(n, n)
    sd = net.state_dict()
    return [sd[k].to(device) for k in PARAM_ORDER]


def cpu_baseline(n, k, n_playout, params_np, seconds_hint=25.0):
    """The CPU oracle (C port of the reference's mcts_alphaZero + numpy-net path) timed on the
    host cores: one whole game per thread, float32 net."""
    from oracle import az_oracle
    az_oracle.build()
    threads = max(1, min(16, os.cpu_count() or 1))
    net = az_oracle.OracleNet(n, n, params_np, precise=False)
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
            "single_core": {"value": 1.0 / dt1, "unit": "games/s", "cores": 1,
                            "sample": "1 game (seed 0), %d plies in %.1f s" % (plies1, dt1)}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=96)
    ap.add_argument("--warmup", type=int, default=64)
    ap.add_argument("--games", type=int, default=1024, help="game slots per GPU")
    ap.add_argument("--board", type=int, default=8)
    ap.add_argument("--n-in-row", type=int, default=5)
    ap.add_argument("--playouts", type=int, default=400)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--host-refill", action="store_true",
                    help="restart finished slots from the host between steps instead of the pipelined azg_selfplay_* calls")
    ap.add_argument("--cnn", choices=["f16x3", "f32"], default="f16x3",
                    help="conv arithmetic: split-f16 MFMA (three products, f32 accumulate) or f32-input MFMA")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    use_dist = world > 1 or "RANK" in os.environ
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
    eng = az.SelfPlayEngine(n, k, args.playouts, args.games, 5.0, device=local_rank,
                            cnn_mode=az._lib.CNN_F16X3 if args.cnn == "f16x3" else az._lib.CNN_F32)
    eng.set_weights(params)
    drained = []
    run = SelfPlayRunner(eng, temp=1.0, seed0=0, first_game_id=rank, game_id_stride=world,
                         total=None, on_samples=drained.append, device_samples=True, pipelined=not args.host_refill)
    run.start()

    def gather_all(chunks):
        # the (state, pi, z) gather of the sharded configuration: every rank ends up with every
        # rank's tuples (RCCL all_gather over xGMI)
        if chunks:
            mine = {key: torch.cat([d[key] for d in chunks]) for key in ("states", "pi", "z", "game_id", "ply")}
        else:
            mine = {"states": torch.empty((0, 4, n, n), dtype=torch.float32, device=dev), "pi": torch.empty((0, n * n), dtype=torch.float64, device=dev),
                    "z": torch.empty(0, device=dev), "game_id": torch.empty(0, dtype=torch.int64, device=dev),
                    "ply": torch.empty(0, dtype=torch.int32, device=dev)}
        mine["states"] = mine["states"].to(torch.uint8)   # the planes are 0/1: exact, a quarter of the bytes on the links
        return int(azdist.gather_samples(mine)["z"].shape[0])

    for _ in range(args.warmup):
        run.step()
    if use_dist:
        gather_all(drained)          # warm-up of the exchange too: RCCL sets its channels up on first use
    drained.clear()
    eng.profile()                                   # reset the kernel timers
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    g0, p0 = run.games_done, run.plies_done
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
    games, plies = run.games_done - g0, run.plies_done - p0
    launches, kernel_ms, evals = eng.profile()

    stats = torch.tensor([dt, games, plies, evals, kernel_ms, launches], dtype=torch.float64, device=dev)
    if use_dist:
        mx = stats.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        sm = stats.clone()
        dist.all_reduce(sm, op=dist.ReduceOp.SUM)
        dt = float(mx[0])
        games, plies, evals = float(sm[1]), float(sm[2]), float(sm[3])
        kernel_ms, launches = float(sm[4]) / world, float(sm[5]) / world
    if rank == 0:
        flop = FLOP_PER_EVAL.get(n, 188416.0 * n * n + 8.0 * n ** 4 + 128)
        # per-GPU dominant kernel: leaf-eval flops of one launch / its HIP-event duration
        evals_per_gpu = evals / world
        achieved = flop * evals_per_gpu / (kernel_ms * 1e-3) / 1e12 if kernel_ms > 0 else 0.0
        out = {
            "metric": "completed self-play games/sec, %dx%d board n_playout=%d" % (n, n, args.playouts),
            "value": games / dt, "unit": "games/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None,
            "dtype": "f16x3 (f32 operands split into two halves, three f16 MFMAs per product, f32 accumulate; tree in f64)" if args.cnn == "f16x3" else "f32",
            "data": "synthetic",
            "config": {"workload": "%dx%d board %d-in-row, n_playout=%d, %d parallel self-play games per MI355X, "
                                   "random-init net (torch.manual_seed(0) default init), empty-board starts, "
                                   "c_puct=5, temp=1, Dirichlet(0.1) mix 0.25, tree reuse, finished slots refilled"
                                   % (n, n, k, args.playouts, args.games),
                       "games_in_flight_per_gpu": args.games, "sharding": "game id mod n_gpus",
                       "step": "one move (n_playout playouts + sample/advance/re-root) for every slot"},
            "games_completed": games, "plies_per_game": plies / games if games else None,
            "playouts_per_s": evals / dt, "samples_gathered": gathered,
            "roofline": {"bound": "mfma", "kernel": "search_kernel (fused select + leaf CNN + expand + backup)",
                         "achieved": achieved, "peak": PEAK_TFLOPS[args.cnn], "unit": "TFLOP/s",
                         "frac": achieved / PEAK_TFLOPS[args.cnn],
                         "traffic": PMC_TRAFFIC_BYTES.get((args.cnn, n, args.playouts, args.games)),
                         "traffic_note": "HBM bytes per launch, FETCH_SIZE + WRITE_SIZE of a separate rocprofv3 --pmc run "
                                         "(tree arrays; 4-8 B per lane accesses, no width correction applied)",
                         "note": "achieved counts the ALGORITHMIC flops of the leaf CNN (12.09 MFLOP per evaluation at 8x8); "
                                 "the f16x3 path issues ~2.9x that many MFMA flops",
                         "launches": launches, "avg_launch_ms": kernel_ms / launches if launches else None,
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
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(n, k, args.playouts, [p.cpu().numpy() for p in params])
        print(json.dumps(out))
    eng.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
