"""Shared helpers for tests and tools/gen_golden.py (no reference, no product imports)."""
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden")
M64 = (1 << 64) - 1


def mix64(z):
    z &= M64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M64
    return z ^ (z >> 31)


def state_key(states, current_player, n_cells):
    """Position key: stones of player 1 then player 2 as 64-bit words, plus side to move.

    `states` maps move -> player (the reference Board.states dict)."""
    words = (n_cells + 63) // 64
    k = (0x9E3779B97F4A7C15 ^ int(current_player)) & M64
    for c in (1, 2):
        for w in range(words):
            word = 0
            for m, p in states.items():
                m = int(m)
                if p == c and 64 * w <= m < 64 * (w + 1):
                    word |= 1 << (m - 64 * w)
            k = mix64(k ^ word)
    return k


def hash_policy(states, current_player, n_cells):
    """Exact synthetic priors (dyadic rationals, float32-representable) and value."""
    k = state_key(states, current_player, n_cells)
    probs = np.empty(n_cells, dtype=np.float64)
    for a in range(n_cells):
        r = (mix64(k + 0x9E3779B97F4A7C15 * (a + 1)) >> 32) & 0xFFF
        probs[a] = float(np.float32(r + 1) / np.float32(131072.0))
    rv = (mix64(k ^ 0xD1B54A32D192ED03) >> 32) & 0xFFF
    return probs, (float(rv) - 2048.0) / 2048.0


def uniform32_policy(n_avail, n_cells):
    p = float(np.float32(1.0) / np.float32(n_avail)) if n_avail > 0 else 0.0
    return np.full(n_cells, p, dtype=np.float64), 0.0


PARAM_SHAPES = lambda n: [  # noqa: E731  PyTorch layout, policy_value_net_pytorch.py:31-41
    (32, 4, 3, 3), (32,), (64, 32, 3, 3), (64,), (128, 64, 3, 3), (128,),
    (4, 128, 1, 1), (4,), (n * n, 4 * n * n), (n * n,),
    (2, 128, 1, 1), (2,), (64, 2 * n * n), (64,), (1, 64), (1,)]


def make_params(n, seed=0, scale=1.0):
    """Deterministic random-init weights (PyTorch-default-style uniform(+-1/sqrt(fan_in)))
    in the PyTorch tensor layout; float32.  `scale` > 1 sharpens the outputs."""
    rs = np.random.RandomState(seed)
    shapes = PARAM_SHAPES(n)
    out = []
    for i in range(0, 16, 2):
        ws, bs = shapes[i], shapes[i + 1]
        fan_in = int(np.prod(ws[1:]))
        bound = scale / np.sqrt(fan_in)
        out.append(rs.uniform(-bound, bound, size=ws).astype(np.float32))
        out.append(rs.uniform(-bound, bound, size=bs).astype(np.float32))
    return out


def torch_to_theano_layout(params):
    """PyTorch-layout tensors -> the 16-tensor list policy_value_net_numpy.py:103-113
    consumes (conv kernels rotated 180 degrees, FC weights transposed)."""
    out = []
    for i, p in enumerate(params):
        if p.ndim == 4:
            out.append(np.ascontiguousarray(p[:, :, ::-1, ::-1]))
        elif p.ndim == 2:
            out.append(np.ascontiguousarray(p.T))
        else:
            out.append(p.copy())
    return out


def random_positions(n, n_in_row, count, seed):
    """Move prefixes of random legal non-terminal play (lists of ints)."""
    rs = np.random.RandomState(seed)
    out = []
    for _ in range(count):
        length = int(rs.randint(0, max(1, n * n - n_in_row)))
        out.append([int(x) for x in rs.permutation(n * n)[:length]])
    return out


def choice_index(probs, u):
    """np.random.choice(acts, p=probs) for one draw, given its uniform u: cumulative sums,
    normalised by the last one, searchsorted(side='right') (numpy/random/mtrand.pyx)."""
    cdf = np.cumsum(np.asarray(probs, dtype=np.float64))
    cdf /= cdf[-1]
    return int(min(np.searchsorted(cdf, u, side="right"), len(cdf) - 1))


# ---- fixtures made of records (lists of flat dicts) stored as plain arrays -----------------------
def pack_records(name, records):
    """-> {"name/count": n, "name/i/key": array}: scalars, strings, lists and arrays of a list of
    flat dicts as plain numeric / unicode arrays, so the archive loads with allow_pickle=False."""
    out = {name + "/count": np.int64(len(records))}
    for i, r in enumerate(records):
        for key, v in r.items():
            a = np.asarray(v)
            assert a.dtype != object, (name, i, key)
            out["%s/%d/%s" % (name, i, key)] = a
    return out


def load_golden(fname):
    """An .npz fixture without unpickling anything: plain entries as they are, record lists
    (see pack_records) rebuilt as lists of dicts with Python scalars for 0-d entries."""
    out, recs = {}, {}
    with np.load(os.path.join(GOLDEN, fname), allow_pickle=False) as z:
        for key in z.files:
            parts = key.split("/")
            if len(parts) == 2 and parts[1] == "count":
                recs.setdefault(parts[0], {})
                out[parts[0]] = [dict() for _ in range(int(z[key]))]
            elif len(parts) == 3:
                recs.setdefault(parts[0], {})[(int(parts[1]), parts[2])] = z[key]
            else:
                out[key] = z[key]
    for name, items in recs.items():
        for (i, key), a in items.items():
            out[name][i][key] = a.item() if a.ndim == 0 else a
    return out
