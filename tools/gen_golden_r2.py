#!/usr/bin/env python3
"""Second batch of fixtures generated from the reference (build container only, needs
/root/reference):

  g6_equi.npz     TrainPipeline.get_equi_data run by the reference's own function body
  g9_train.npz    PolicyValueNet.train_step (parameters after 1 and 3 Adam steps, loss, entropy,
                  outputs of the updated net) and TrainPipeline.policy_update's KL early stop and
                  learning-rate multiplier on scripted KL sequences
  g10_trained.npz the bundled trained 6x6 and 8x8 nets (mode/best_policy_*.model): parameters,
                  outputs of the reference's numpy net, one seeded self-play game each
  g11_ties.npz    seeded 3x3 self-play games of the reference, ties among them

train.py cannot be imported (module-level Keras import, yaml.load without Loader), so single
methods of TrainPipeline are taken out of its syntax tree and compiled on their own, against a
stand-in `self`; nothing of the reference's text is written anywhere.

    PYTHONDONTWRITEBYTECODE=1 python tools/gen_golden_r2.py
"""
import ast
import collections
import os
import pickletools
import random
import sys

sys.dont_write_bytecode = True
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import numpy as np  # noqa: E402

import gen_golden as g1  # noqa: E402  (imports the reference's game / mcts / numpy-net modules)
from gen_golden import OUT, REF, helpers, ref_np_net  # noqa: E402


# ---------------------------------------------------------------- reference methods by syntax tree
def reference_method(path, cls, name, glob):
    """Compile ONE method of a class in a reference file that cannot be imported as a whole."""
    with open(path, encoding="utf-8") as f:
        tree = ast.parse(f.read())
    for node in tree.body:
        if isinstance(node, ast.ClassDef) and node.name == cls:
            for item in node.body:
                if isinstance(item, ast.FunctionDef) and item.name == name:
                    item.decorator_list = []
                    mod = ast.Module(body=[item], type_ignores=[])
                    ns = dict(glob)
                    exec(compile(mod, path, "exec"), ns)
                    return ns[name]
    raise KeyError((cls, name))


class _Quiet:
    def info(self, *a, **k):
        pass
    warning = error = debug = info


class _Self:
    pass


# ---------------------------------------------------------------- G6 augmentation, pinned
def gen_equi():
    fn = reference_method(os.path.join(REF, "train.py"), "TrainPipeline", "get_equi_data", {"np": np})
    out = {}
    rng = np.random.RandomState(7)
    for n in (6, 8, 15):
        cnt = 5
        states = (rng.rand(cnt, 4, n, n) < 0.3).astype(np.float32)
        pi = rng.rand(cnt, n * n)
        pi /= pi.sum(axis=1, keepdims=True)
        z = rng.choice([-1.0, 0.0, 1.0], size=cnt).astype(np.float32)
        me = _Self()
        me.board_height = me.board_width = n
        ext = fn(me, [(states[i], pi[i], z[i]) for i in range(cnt)])
        assert len(ext) == 8 * cnt
        out["states_%d" % n], out["pi_%d" % n], out["z_%d" % n] = states, pi, z
        out["ext_states_%d" % n] = np.stack([e[0] for e in ext]).astype(np.float32)
        out["ext_pi_%d" % n] = np.stack([e[1] for e in ext])
        out["ext_z_%d" % n] = np.asarray([e[2] for e in ext], dtype=np.float32)
    # the earlier fixture (explicit index loops) must be what the reference computes
    old = np.load(os.path.join(OUT, "g6_equi.npz"))
    for key in old.files:
        assert np.array_equal(old[key], out[key]), key
    out["source"] = np.array("train.py:116-135 executed")
    np.savez_compressed(os.path.join(OUT, "g6_equi.npz"), **out)
    print("g6_equi: the reference's get_equi_data reproduces the earlier fixture; rewritten")


# ---------------------------------------------------------------- G9 train_step / policy_update
def gen_train():
    import torch
    import policy_value_net_pytorch as ref_t
    out = {}
    n, k, B = 8, 5, 64
    params = helpers.make_params(n, seed=108)
    torch.manual_seed(0)
    pvn = ref_t.PolicyValueNet(n, n)
    sd = pvn.policy_value_net.state_dict()
    for key, p in zip(sd.keys(), params):
        sd[key].copy_(torch.from_numpy(p.copy()))
    rs = np.random.RandomState(11)
    states = np.stack([g1.new_board(n, k, mv).current_state() for mv in helpers.random_positions(n, k, B, seed=5)])
    pi = rs.dirichlet(0.3 * np.ones(n * n), size=B)
    z = rs.choice([-1.0, 1.0], size=B)
    lrs = [2e-3, 2e-3 * 1.5, 2e-3 / 1.5]
    losses, ents = [], []
    for step, lr in enumerate(lrs):
        # loss and entropy as policy_value_net_pytorch.py:135-147 defines them, from the net BEFORE the step
        with torch.no_grad():
            logp, v = pvn.policy_value_net(torch.FloatTensor(states))
            loss = torch.nn.functional.mse_loss(v.view(-1), torch.FloatTensor(z)) - torch.mean(torch.sum(torch.FloatTensor(pi) * logp, 1))
            ent = -torch.mean(torch.sum(torch.exp(logp) * logp, 1))
        losses.append(float(loss))
        ents.append(float(ent))
        try:
            pvn.train_step(list(states), list(pi), list(z), lr)
            raise AssertionError("the reference's train_step returned: expected its IndexError at loss.data[0]")
        except IndexError:
            pass                                   # raised at :148, after optimizer.step() at :143
        if step in (0, 2):
            for i, key in enumerate(sd.keys()):
                out["params_after_%d_%d" % (step + 1, i)] = pvn.policy_value_net.state_dict()[key].numpy().copy()
    with torch.no_grad():
        probs, vals = pvn.policy_value(states[:16])
    out.update(states=states.astype(np.uint8), pi=pi, z=z, lrs=np.array(lrs), loss=np.array(losses), entropy=np.array(ents),
               probs_after_3=probs.astype(np.float32), values_after_3=vals.astype(np.float32).reshape(-1),
               spec=np.array([n, k, 108]))
    print("g9 train_step: loss %s entropy %s" % (np.round(losses, 5), np.round(ents, 5)))

    # ---- policy_update (train.py:161-214) on scripted nets: the sequence of KL values decides how
    # many train steps run and what happens to lr_multiplier
    fn = reference_method(os.path.join(REF, "train.py"), "TrainPipeline", "policy_update",
                          {"np": np, "random": random, "_logger": _Quiet()})
    S, nb = 9, 12

    class Scripted:
        """policy_value returns table[calls]; train_step records its learning rate."""

        def __init__(self, tables):
            self.tables, self.calls, self.lrs = tables, 0, []

        def policy_value(self, state_batch):
            t = self.tables[min(self.calls, len(self.tables) - 1)]
            self.calls += 1
            return t, np.linspace(-0.5, 0.5, len(state_batch)).reshape(-1, 1)

        def train_step(self, s, p, w, lr):
            self.lrs.append(lr)
            return 1.0 + len(self.lrs), 2.0

    def table(shift):
        base = np.tile(np.linspace(1.0, 2.0, S), (nb, 1))
        base[:, 0] += shift
        return base / base.sum(axis=1, keepdims=True)
    scen = []
    for name, shifts, mult0 in [("kl_small_raises_multiplier", [0.0, 0.01, 0.02, 0.03, 0.03, 0.03], 1.0),
                                ("kl_medium_keeps_multiplier", [0.0, 0.3, 0.5, 0.8, 1.0, 1.3], 1.0),
                                ("kl_large_lowers_multiplier", [0.0, 0.6, 1.0, 1.5, 1.9, 2.2], 1.0),
                                ("kl_huge_stops_early", [0.0, 0.5, 3.0, 9.0, 9.0, 9.0], 1.0),
                                ("multiplier_floor", [0.0, 0.9, 1.5, 1.9, 2.2, 2.5], 0.04),
                                ("multiplier_ceiling", [0.0, 0.0, 0.0, 0.0, 0.0, 0.0], 11.0)]:
        me = _Self()
        me.batch_size, me.epochs, me.learn_rate, me.lr_multiplier, me.kl_targ = nb, 5, 2e-3, mult0, 0.02
        me.data_buffer = collections.deque([(np.zeros((4, 3, 3)), np.ones(S) / S, float((-1) ** i)) for i in range(nb)], maxlen=100)
        tabs = [table(s) for s in shifts]
        me.policy_value_net = Scripted(tabs)
        random.seed(3)
        loss, ent = fn(me)
        kls = [float(np.mean(np.sum(tabs[0] * (np.log(tabs[0] + 1e-10) - np.log(t + 1e-10)), axis=1))) for t in tabs[1:]]
        scen.append(dict(name=name, tables=np.stack(tabs), lr_multiplier_before=mult0, lr_multiplier_after=me.lr_multiplier,
                         train_steps=len(me.policy_value_net.lrs), lrs=np.array(me.policy_value_net.lrs), kls=np.array(kls),
                         loss=loss, entropy=ent))
        print("  policy_update %-28s kl %s -> %d steps, multiplier %.4f -> %.4f"
              % (name, np.round(kls, 4), len(me.policy_value_net.lrs), mult0, me.lr_multiplier))
    assert {s["train_steps"] for s in scen} >= {5, 2} and any(s["lr_multiplier_after"] > 1 for s in scen) \
        and any(s["lr_multiplier_after"] < 1 for s in scen)
    out.update(helpers.pack_records("policy_update", scen))
    np.savez_compressed(os.path.join(OUT, "g9_train.npz"), **out)
    print("g9_train written")


# ---------------------------------------------------------------- bundled trained nets
class _Inert:
    """What an opcode that would build an object leaves on the stack: a record of it."""

    def __init__(self, fn, args):
        self.fn, self.args, self.state = fn, args, None


def read_array_list(path):
    """The list of float32 arrays in a protocol-2 pickle written by the reference's Theano net
    (policy_value_net.py:118-121), WITHOUT unpickling: the file is only disassembled
    (pickletools.genops) and a handful of data opcodes are replayed on a stack of plain Python
    values.  GLOBAL and REDUCE never import or call anything: they leave inert records whose names
    are checked afterwards.  Any other opcode is refused."""
    with open(path, "rb") as f:
        data = f.read()
    MARK = object()
    stack, memo = [], {}

    def pop_mark():
        i = len(stack) - 1
        while stack[i] is not MARK:
            i -= 1
        items = stack[i + 1:]
        del stack[i:]
        return items
    for op, arg, _ in pickletools.genops(data):
        name = op.name
        if name == "PROTO":
            pass
        elif name == "EMPTY_LIST":
            stack.append([])
        elif name == "MARK":
            stack.append(MARK)
        elif name in ("BINPUT", "LONG_BINPUT"):
            memo[arg] = stack[-1]
        elif name in ("BINGET", "LONG_BINGET"):
            stack.append(memo[arg])
        elif name == "GLOBAL":
            stack.append(("global", arg))
        elif name in ("BININT", "BININT1", "BININT2", "LONG1"):
            stack.append(int(arg))
        elif name in ("SHORT_BINSTRING", "BINSTRING"):
            stack.append(arg.encode("latin-1") if isinstance(arg, str) else bytes(arg))
        elif name == "NONE":
            stack.append(None)
        elif name in ("NEWFALSE", "NEWTRUE"):
            stack.append(name == "NEWTRUE")
        elif name in ("TUPLE1", "TUPLE2", "TUPLE3"):
            k = int(name[-1])
            items = tuple(stack[-k:])
            del stack[-k:]
            stack.append(items)
        elif name == "TUPLE":
            stack.append(tuple(pop_mark()))
        elif name == "REDUCE":
            args = stack.pop()
            fn = stack.pop()
            stack.append(_Inert(fn, args))
        elif name == "BUILD":
            state = stack.pop()
            stack[-1].state = state
        elif name == "APPENDS":
            items = pop_mark()
            stack[-1].extend(items)
        elif name == "APPEND":
            item = stack.pop()
            stack[-1].append(item)
        elif name == "STOP":
            break
        else:
            raise ValueError("opcode %s is not a plain-data opcode: file refused" % name)
    top = stack.pop()
    arrays = []
    for rec in top:
        assert isinstance(rec, _Inert) and rec.fn == ("global", "numpy.core.multiarray _reconstruct"), rec.fn
        version, shape, dt, fortran, raw = rec.state
        assert isinstance(dt, _Inert) and dt.fn == ("global", "numpy dtype") and dt.args[0] == b"f4" and dt.state[1] == b"<"
        assert version == 1 and fortran is False and len(raw) == 4 * int(np.prod(shape))
        arrays.append(np.frombuffer(raw, dtype="<f4").reshape(shape).copy())
    return arrays


def all_probs(nnet, board, n):
    """The numpy net's probabilities for all S cells and its value (policy_value_fn returns only
    the legal entries): the same calls as policy_value_net_numpy.py:92-115."""
    X = board.current_state().reshape(-1, 4, n, n)
    for i in [0, 2, 4]:
        X = ref_np_net.relu(ref_np_net.conv_forward(X, nnet.params[i], nnet.params[i + 1]))
    Xp = ref_np_net.relu(ref_np_net.conv_forward(X, nnet.params[6], nnet.params[7], padding=0))
    pn = ref_np_net.softmax(ref_np_net.fc_forward(Xp.flatten(), nnet.params[8], nnet.params[9]))
    ap, v = nnet.policy_value_fn(board)
    for a, p in ap:
        assert abs(pn[a] - p) < 1e-15
    return pn, float(v)


def gen_trained():
    out = {}
    games = []
    for (n, k, npo, fname, seed) in [(6, 4, 200, "best_policy_6_6_4.model", 0), (8, 5, 400, "best_policy_8_8_5.model", 1)]:
        params = read_array_list(os.path.join(REF, "mode", fname))
        want = [tuple(np.asarray(p).shape) for p in helpers.torch_to_theano_layout(helpers.make_params(n))]
        assert [p.shape for p in params] == want, [p.shape for p in params]
        nnet = ref_np_net.PolicyValueNetNumpy(n, n, params)
        states, probs, vals = [], [], []
        for mv in helpers.random_positions(n, k, 64, seed=13 * n):
            b = g1.new_board(n, k, mv)
            if b.game_end()[0]:
                continue
            pn, v = all_probs(nnet, b, n)
            states.append(b.current_state().astype(np.uint8))
            probs.append(pn)
            vals.append(v)
        # positions of real play too (the nets were trained on those)
        def cb_net(board, nnet=nnet):
            ap, v = nnet.policy_value_fn(board)
            return [(a, float(np.float32(p))) for a, p in ap], float(np.float32(v))
        g = g1.record_game(n, k, cb_net, 5, npo, 1.0, seed)
        g["cb"] = "numpy_net_f32"
        g["model"] = fname
        games.append(g)
        b = g1.new_board(n, k)
        for m in g["moves"][:-1]:
            b.do_move(int(m))
            pn, v = all_probs(nnet, b, n)
            states.append(b.current_state().astype(np.uint8))
            probs.append(pn)
            vals.append(v)
        for i, p in enumerate(params):
            out["params_%d_%d" % (n, i)] = p
        out["states_%d" % n], out["probs_%d" % n], out["values_%d" % n] = np.stack(states), np.stack(probs), np.array(vals)
        amax = max(float(np.abs(p).max()) for p in params)
        print("  %s: %d positions, max|w| %.3f, max prob %.4f, |v| up to %.4f; game %d plies winner %d"
              % (fname, len(states), amax, np.stack(probs).max(), np.abs(vals).max(), len(g["moves"]), g["winner"]))
    out.update(helpers.pack_records("games", games))
    np.savez_compressed(os.path.join(OUT, "g10_trained.npz"), **out)
    print("g10_trained written")


# ---------------------------------------------------------------- ties
def gen_ties():
    games = []
    for seed in range(40):
        g = g1.record_game(3, 3, g1.cb_hash, 5, 30, 1.0, 900 + seed)
        g["cb"] = "hash"
        games.append(g)
    ties = sum(1 for g in games if g["winner"] == -1)
    print("g11_ties: %d games on 3x3, %d ties, lengths %s" % (len(games), ties, sorted(set(len(g["moves"]) for g in games))))
    assert ties >= 2 and any(g["winner"] == 1 for g in games)
    for g in games:
        if g["winner"] == -1:
            assert len(g["moves"]) == 9 and not g["z"].any()
    np.savez_compressed(os.path.join(OUT, "g11_ties.npz"), **helpers.pack_records("games", games))


if __name__ == "__main__":
    which = sys.argv[1:] or ["equi", "train", "trained", "ties"]
    for name in which:
        {"equi": gen_equi, "train": gen_train, "trained": gen_trained, "ties": gen_ties}[name]()
