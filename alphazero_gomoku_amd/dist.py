"""Multi-GPU self-play: games shard over ranks, (state, pi, z) tuples are gathered.

One process per GPU (torch.distributed; backend "nccl" is RCCL on ROCm, "gloo" on CPU).
Games are independent, so there is no collective on the search path: rank r plays the
games with id = r, r + world, r + 2*world, ... and game id g always uses seed seed0 + g,
which makes the set of games independent of the number of ranks.  The only exchange is
the variable-length gather of training tuples: an all_gather of the per-rank counts,
then one padded all_gather per tensor."""
import os

import torch
import torch.distributed as dist


def shard_game_ids(rank, world, n_games_total):
    """(first_game_id, stride, count) of the games rank `rank` plays."""
    count = (n_games_total - rank + world - 1) // world if n_games_total > rank else 0
    return rank, world, count


def _comm_device(dev, group=None):
    """Device the collectives' buffers live on: the tensors' own (RCCL moves GPU memory directly),
    the host under the gloo backend (CPU tests; rehearsals of several ranks on one GPU)."""
    if dist.get_backend(group) == "gloo" and os.environ.get("AZG_GLOO_DEVICE_TENSORS") != "1":   # (=1: diagnostic only)
        return torch.device("cpu")
    return dev


def gather_samples(tensors, group=None):
    """tensors: dict name -> tensor with the same leading length n (this rank's samples),
    all on this rank's device.  Returns dict name -> tensor holding every rank's samples,
    rank after rank.  Collective: every rank must call it with the same keys."""
    world = dist.get_world_size(group)
    names = sorted(tensors)
    n = int(tensors[names[0]].shape[0]) if names else 0
    own = tensors[names[0]].device
    dev = _comm_device(own, group)
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
        out[name] = torch.cat([buf[r, :counts_h[r]] for r in range(world)], dim=0).to(own)
    return out


PACKED_FIELDS = (("stones", torch.int64, 8), ("last", torch.int32, 1), ("ply", torch.int32, 1),
                 ("z", torch.float32, 1), ("game_id", torch.int64, 1))


def pack_samples(t):
    """Training tuples in their packed form -- stones int64 [n,2,4] (the two colours' bit sets),
    last i32 [n], ply i32 [n], z f32 [n], game_id i64 [n], pi f64 [n,S] -- as ONE uint8 matrix
    [n, 84 + 8*S]: a row is a whole tuple (596 bytes at 8x8 against 1.3 KB with float planes)."""
    n = int(t["z"].shape[0])
    cols = [t[name].to(dtype).contiguous().view(torch.uint8).reshape(n, -1) for name, dtype, _ in PACKED_FIELDS]
    cols.append(t["pi"].to(torch.float64).contiguous().view(torch.uint8).reshape(n, -1))
    return torch.cat(cols, dim=1)


def unpack_samples(rows, n_cells):
    """Inverse of pack_samples."""
    n = int(rows.shape[0])
    out, off = {}, 0
    for name, dtype, count in PACKED_FIELDS:
        width = count * torch.empty(0, dtype=dtype).element_size()
        out[name] = rows[:, off:off + width].contiguous().view(dtype).reshape((n, 2, 4) if name == "stones" else (n,))
        off += width
    out["pi"] = rows[:, off:off + 8 * n_cells].contiguous().view(torch.float64).reshape(n, n_cells)
    return out


def gather_packed(t, n_cells, group=None):
    """All ranks' training tuples, rank after rank, with two collectives in all: an all_gather of
    the per-rank counts and ONE padded all_gather of the packed rows (pack_samples).  The feature
    planes are not sent; the receiver rebuilds them from the stones (SelfPlayEngine.unpack_states)."""
    world = dist.get_world_size(group)
    rows = pack_samples(t)
    n, own = int(rows.shape[0]), rows.device
    dev = _comm_device(own, group)
    counts = torch.zeros(world, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(counts, torch.tensor([n], dtype=torch.int64, device=dev), group=group)
    counts_h = counts.cpu().tolist()
    nmax = max(counts_h)
    pad = torch.zeros((nmax, rows.shape[1]), dtype=torch.uint8, device=dev)
    pad[:n] = rows
    buf = torch.empty((world * nmax, rows.shape[1]), dtype=torch.uint8, device=dev)
    dist.all_gather_into_tensor(buf, pad, group=group)
    buf = buf.view(world, nmax, rows.shape[1])
    return unpack_samples(torch.cat([buf[r, :counts_h[r]] for r in range(world)], dim=0).to(own), n_cells)


def broadcast_weights(net, src=0, group=None):
    """Replicate rank `src`'s parameters (476 KB at 8x8) after an optimiser step."""
    for t in net.param_tensors():
        c = t.to(_comm_device(t.device, group))
        dist.broadcast(c, src=src, group=group)
        if c is not t:
            t.copy_(c)
    net.version += 1


def broadcast_indices(idx, device, src=0, group=None):
    """The mini-batch positions rank `src` drew, as a list, on every rank."""
    t = torch.tensor(idx, dtype=torch.int64, device=_comm_device(device, group))
    dist.broadcast(t, src=src, group=group)
    return t.tolist()


def max_over_ranks(x, device, group=None):
    """The largest of the ranks' values of the float x."""
    t = torch.tensor([x], dtype=torch.float64, device=_comm_device(device, group))
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t[0])


def allreduce_flat(flat, group=None):
    """Average the flat gradient vector `flat` over the ranks in place (one all_reduce)."""
    world = dist.get_world_size(group)
    if world == 1:
        return
    own = flat.device
    buf = flat.to(_comm_device(own, group))                  # gloo: through the host
    dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
    flat.copy_(buf.to(own))
    flat /= world


def allreduce_gradients(module, group=None):
    """Data-parallel training step (one rank per GPU, each on its share of the mini-batch):
    average the gradients of `module` over the ranks.  The whole network is 119 k parameters at
    8x8 (476 KB), so all gradients travel as ONE flat buffer in one all_reduce: the exchange is
    latency-bound and a single collective beats per-tensor or bucketed ones."""
    world = dist.get_world_size(group)
    params = [p for p in module.parameters() if p.grad is not None]
    if world == 1 or not params:
        return
    flat = torch.cat([p.grad.reshape(-1) for p in params])
    own = flat.device
    flat = flat.to(_comm_device(own, group))                 # gloo: through the host
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    flat = flat.to(own)
    flat /= world
    off = 0
    for p in params:
        n = p.grad.numel()
        p.grad.copy_(flat[off:off + n].view_as(p.grad))
        off += n
