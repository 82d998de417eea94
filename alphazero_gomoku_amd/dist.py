"""Multi-GPU self-play: games shard over ranks, (state, pi, z) tuples are gathered.

One process per GPU (torch.distributed; backend "nccl" is RCCL on ROCm, "gloo" on CPU).
Games are independent, so there is no collective on the search path: rank r plays the
games with id = r, r + world, r + 2*world, ... and game id g always uses seed seed0 + g,
which makes the set of games independent of the number of ranks.  The only exchange is
the variable-length gather of training tuples: an all_gather of the per-rank counts,
then one padded all_gather per tensor."""
import torch
import torch.distributed as dist


def shard_game_ids(rank, world, n_games_total):
    """(first_game_id, stride, count) of the games rank `rank` plays."""
    count = (n_games_total - rank + world - 1) // world if n_games_total > rank else 0
    return rank, world, count


def gather_samples(tensors, group=None):
    """tensors: dict name -> tensor with the same leading length n (this rank's samples),
    all on this rank's device.  Returns dict name -> tensor holding every rank's samples,
    rank after rank.  Collective: every rank must call it with the same keys."""
    world = dist.get_world_size(group)
    names = sorted(tensors)
    n = int(tensors[names[0]].shape[0]) if names else 0
    dev = tensors[names[0]].device
    counts = torch.zeros(world, dtype=torch.int64, device=dev)
    mine = torch.tensor([n], dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(counts, mine, group=group)
    counts_h = counts.cpu().tolist()
    nmax = max(counts_h) if counts_h else 0
    out = {}
    for name in names:
        t = tensors[name].contiguous()
        pad = torch.zeros((nmax,) + tuple(t.shape[1:]), dtype=t.dtype, device=dev)
        pad[:n] = t
        buf = torch.empty((world * nmax,) + tuple(t.shape[1:]), dtype=t.dtype, device=dev)
        dist.all_gather_into_tensor(buf, pad, group=group)
        buf = buf.view((world, nmax) + tuple(t.shape[1:]))
        out[name] = torch.cat([buf[r, :counts_h[r]] for r in range(world)], dim=0)
    return out


def broadcast_weights(net, src=0, group=None):
    """Replicate rank `src`'s parameters (476 KB at 8x8) after an optimiser step."""
    for t in net.param_tensors():
        dist.broadcast(t, src=src, group=group)
    net.version += 1
