"""ctypes binding of the CPU oracle (oracle/az_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The product package never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libaz_oracle.so")
MAX_CELLS = 256


def build(force=False):
    """Compile the oracle with gcc (no-op when the library is up to date)."""
    src = os.path.join(_HERE, "az_oracle.c")
    hdr = os.path.join(_HERE, "az_oracle.h")
    if (not force and os.path.exists(_LIB_PATH)
            and os.path.getmtime(_LIB_PATH) >= max(os.path.getmtime(src), os.path.getmtime(hdr))):
        return _LIB_PATH
    subprocess.check_call(["make", "-C", _HERE, "-B"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


class Board(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("n_in_row", C.c_int),
                ("cell", C.c_int * MAX_CELLS), ("avail", C.c_int * MAX_CELLS),
                ("n_avail", C.c_int), ("current_player", C.c_int),
                ("last_move", C.c_int), ("n_stones", C.c_int)]


POLICY_FN = C.CFUNCTYPE(None, C.POINTER(Board), C.POINTER(C.c_double),
                        C.POINTER(C.c_double), C.c_void_p)


class Net(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int),
                ("p", C.POINTER(C.c_float) * 16), ("precise", C.c_int)]


class GameRecord(C.Structure):
    _fields_ = [("n_plies", C.c_int), ("winner", C.c_int),
                ("moves", C.c_int * MAX_CELLS), ("players", C.c_int * MAX_CELLS),
                ("z", C.c_double * MAX_CELLS),
                ("pi", C.POINTER(C.c_double)), ("visits", C.POINTER(C.c_int)),
                ("states", C.POINTER(C.c_double))]


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        build()
    L = C.CDLL(_LIB_PATH)
    L.azo_board_init.argtypes = [C.POINTER(Board), C.c_int, C.c_int, C.c_int, C.c_int]
    L.azo_board_do_move.argtypes = [C.POINTER(Board), C.c_int]
    L.azo_board_game_end.argtypes = [C.POINTER(Board), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.azo_board_has_a_winner.argtypes = [C.POINTER(Board), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.azo_board_current_state.argtypes = [C.POINTER(Board), C.POINTER(C.c_double)]
    L.azo_state_key.argtypes = [C.POINTER(Board)]
    L.azo_state_key.restype = C.c_uint64
    L.azo_net_forward.argtypes = [C.POINTER(Net), C.POINTER(C.c_double), C.POINTER(C.c_double),
                                  C.POINTER(C.c_double)]
    L.azo_rng_seed.argtypes = [C.c_void_p, C.c_uint32]
    L.azo_rng_double.argtypes = [C.c_void_p]
    L.azo_rng_double.restype = C.c_double
    L.azo_rng_dirichlet.argtypes = [C.c_void_p, C.c_double, C.c_int, C.POINTER(C.c_double)]
    L.azo_mcts_new.argtypes = [C.c_void_p, C.c_void_p, C.c_double, C.c_int]
    L.azo_mcts_new.restype = C.c_void_p
    L.azo_mcts_free.argtypes = [C.c_void_p]
    L.azo_mcts_get_move_probs.argtypes = [C.c_void_p, C.POINTER(Board), C.c_double,
                                          C.POINTER(C.c_int), C.POINTER(C.c_int),
                                          C.POINTER(C.c_double), C.POINTER(C.c_double),
                                          C.POINTER(C.c_double)]
    L.azo_mcts_update_with_move.argtypes = [C.c_void_p, C.c_int]
    L.azo_mcts_root_children.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int),
                                         C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.azo_mcts_root_visits.argtypes = [C.c_void_p]
    L.azo_self_play.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_double,
                                C.c_int, C.c_double, C.c_uint32, C.POINTER(GameRecord)]
    L.azo_self_play_many.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(Net), C.c_double,
                                     C.c_int, C.c_double, C.c_uint32, C.c_int, C.c_int,
                                     C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.azo_self_play_many.restype = C.c_long
    L.azo_board_move_target.argtypes = [C.POINTER(Board), C.POINTER(C.c_int)]
    L.azo_pure_get_move.argtypes = [C.POINTER(Board), C.c_double, C.c_int, C.c_void_p, C.POINTER(C.c_int),
                                    C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_double),
                                    C.POINTER(C.c_int)]
    _lib = L
    return L


def policy_ptr(kind):
    """Address of a built-in policy provider ('uniform32', 'hash', 'net')."""
    f = getattr(lib(), "azo_policy_" + kind)
    return C.cast(f, C.c_void_p)


class OracleBoard:
    """Thin wrapper with the reference Board's method names."""

    def __init__(self, width=8, height=8, n_in_row=5):
        self.b = Board()
        self.width, self.height, self.n_in_row = width, height, n_in_row

    def init_board(self, start_player=0):
        rc = lib().azo_board_init(C.byref(self.b), self.width, self.height, self.n_in_row,
                                  start_player)
        if rc != 0:
            raise Exception("board width and height can not be less than n_in_row")

    def do_move(self, move):
        if lib().azo_board_do_move(C.byref(self.b), int(move)) != 0:
            raise ValueError("move not available")

    def game_end(self):
        e, w = C.c_int(), C.c_int()
        lib().azo_board_game_end(C.byref(self.b), C.byref(e), C.byref(w))
        return bool(e.value), w.value

    def has_a_winner(self):
        e, w = C.c_int(), C.c_int()
        lib().azo_board_has_a_winner(C.byref(self.b), C.byref(e), C.byref(w))
        return bool(e.value), w.value

    def current_state(self):
        out = np.zeros((4, self.width, self.height))
        lib().azo_board_current_state(C.byref(self.b), out.ctypes.data_as(C.POINTER(C.c_double)))
        return out

    @property
    def availables(self):
        return [self.b.avail[i] for i in range(self.b.n_avail)]

    @property
    def current_player(self):
        return self.b.current_player

    @property
    def last_move(self):
        return self.b.last_move

    def state_key(self):
        return int(lib().azo_state_key(C.byref(self.b)))

    def move_target(self):
        """Board.get_move_target (game.py:35-58): the cells the rollout opponent considers."""
        out = (C.c_int * MAX_CELLS)()
        k = lib().azo_board_move_target(C.byref(self.b), out)
        return [out[i] for i in range(k)]


class OracleNet:
    """16 float32 tensors in the PyTorch layout (policy_value_net_pytorch.py:31-41)."""

    def __init__(self, width, height, params, precise=True):
        assert len(params) == 16
        self.params = [np.ascontiguousarray(p, dtype=np.float32) for p in params]
        self.net = Net()
        self.net.width, self.net.height = width, height
        for i, p in enumerate(self.params):
            self.net.p[i] = p.ctypes.data_as(C.POINTER(C.c_float))
        self.net.precise = 1 if precise else 0
        self.S = width * height

    def forward(self, state):
        st = np.ascontiguousarray(state, dtype=np.float64)
        probs = np.zeros(self.S)
        val = C.c_double()
        lib().azo_net_forward(C.byref(self.net), st.ctypes.data_as(C.POINTER(C.c_double)),
                              probs.ctypes.data_as(C.POINTER(C.c_double)), C.byref(val))
        return probs, val.value

    def ctx(self):
        return C.cast(C.byref(self.net), C.c_void_p)


def _resolve_policy(policy, net=None):
    """-> (function pointer as c_void_p, ctx as c_void_p, keepalive)."""
    if callable(policy):
        def tramp(bptr, probs, value, _ctx):
            b = bptr.contents
            S = b.width * b.height
            p, v = policy(b)
            for i in range(S):
                probs[i] = p[i]
            value[0] = v
        cb = POLICY_FN(tramp)
        return C.cast(cb, C.c_void_p), C.c_void_p(0), cb
    if policy == "net":
        return policy_ptr("net"), net.ctx(), net
    return policy_ptr(policy), C.c_void_p(0), None


class OracleMCTS:
    def __init__(self, policy, c_puct=5, n_playout=400, net=None):
        fn, ctx, self._keep = _resolve_policy(policy, net)
        self.m = lib().azo_mcts_new(fn, ctx, float(c_puct), int(n_playout))

    def get_move_probs(self, board, temp=1e-3):
        acts = (C.c_int * MAX_CELLS)()
        visits = (C.c_int * MAX_CELLS)()
        q = (C.c_double * MAX_CELLS)()
        pr = (C.c_double * MAX_CELLS)()
        probs = (C.c_double * MAX_CELLS)()
        k = lib().azo_mcts_get_move_probs(self.m, C.byref(board.b), float(temp), acts, visits,
                                          q, pr, probs)
        return (np.array(acts[:k]), np.array(visits[:k]), np.array(q[:k]), np.array(pr[:k]),
                np.array(probs[:k]))

    def update_with_move(self, move):
        lib().azo_mcts_update_with_move(self.m, int(move))

    def root_children(self):
        acts = (C.c_int * MAX_CELLS)()
        visits = (C.c_int * MAX_CELLS)()
        q = (C.c_double * MAX_CELLS)()
        pr = (C.c_double * MAX_CELLS)()
        k = lib().azo_mcts_root_children(self.m, acts, visits, q, pr)
        return np.array(acts[:k]), np.array(visits[:k]), np.array(q[:k]), np.array(pr[:k])

    def root_visits(self):
        return lib().azo_mcts_root_visits(self.m)

    def __del__(self):
        try:
            lib().azo_mcts_free(self.m)
        except Exception:
            pass


def self_play(width, n_in_row, policy, c_puct, n_playout, temp, seed, net=None, want_states=True):
    """One seeded self-play game -> dict(moves, players, pi, visits, z, winner, states)."""
    S = width * width
    fn, ctx, keep = _resolve_policy(policy, net)
    rec = GameRecord()
    pi = np.zeros((S, S))
    visits = np.zeros((S, S), dtype=np.int32)
    states = np.zeros((S, 4, width, width)) if want_states else None
    rec.pi = pi.ctypes.data_as(C.POINTER(C.c_double))
    rec.visits = visits.ctypes.data_as(C.POINTER(C.c_int))
    rec.states = states.ctypes.data_as(C.POINTER(C.c_double)) if want_states else None
    rc = lib().azo_self_play(width, width, n_in_row, fn, ctx, float(c_puct), int(n_playout),
                             float(temp), int(seed), C.byref(rec))
    assert rc == 0
    n = rec.n_plies
    del keep
    return dict(moves=np.array(rec.moves[:n]), players=np.array(rec.players[:n]),
                pi=pi[:n].copy(), visits=visits[:n].copy(), z=np.array(rec.z[:n]),
                winner=rec.winner, states=None if states is None else states[:n].copy())


def self_play_many(width, n_in_row, net, c_puct, n_playout, temp, seed0, n_games, n_threads):
    plies = (C.c_int * n_games)()
    winners = (C.c_int * n_games)()
    total = lib().azo_self_play_many(width, width, n_in_row, C.byref(net.net), float(c_puct),
                                     int(n_playout), float(temp), int(seed0), int(n_games),
                                     int(n_threads), plies, winners)
    return int(total), np.array(plies[:]), np.array(winners[:])


class OracleRng:
    def __init__(self, seed):
        self.buf = C.create_string_buffer(624 * 4 + 8)
        lib().azo_rng_seed(self.buf, int(seed))

    def get_state(self):
        """(key[624] uint32, pos) like np.random.get_state()[1:3]."""
        raw = np.frombuffer(self.buf, dtype=np.uint32, count=625).copy()
        return raw[:624], int(raw[624])

    def set_state(self, key, pos):
        raw = np.frombuffer(self.buf, dtype=np.uint32, count=625)
        raw[:624] = np.asarray(key, dtype=np.uint32)
        raw[624] = int(pos)

    def random_sample(self):
        return lib().azo_rng_double(self.buf)

    def dirichlet(self, alpha, k):
        out = np.zeros(k)
        lib().azo_rng_dirichlet(self.buf, float(alpha), int(k), out.ctypes.data_as(C.POINTER(C.c_double)))
        return out


def pure_get_move(board, c_puct, n_playout, rng):
    """mcts_pure.MCTSPlayer.get_action from a fresh root on an OracleBoard; rng: OracleRng (the
    global numpy stream of the reference).  Returns (move, visits by cell, root visits)."""
    S = board.width * board.height
    k, rv = C.c_int(), C.c_int()
    acts, vis = (C.c_int * MAX_CELLS)(), (C.c_int * MAX_CELLS)()
    q = (C.c_double * MAX_CELLS)()
    move = lib().azo_pure_get_move(C.byref(board.b), float(c_puct), int(n_playout), rng.buf, C.byref(k), acts, vis, q,
                                   C.byref(rv))
    visits = np.zeros(S, dtype=np.int32)
    for i in range(k.value):
        visits[acts[i]] = vis[i]
    return move, visits, rv.value


def equi_data(play_data, board_height, board_width):
    """TrainPipeline.get_equi_data (train.py:116-135) restated with the same numpy calls.
    train.py itself cannot be imported (it imports the Keras net at module level); the fixture
    tests/golden/g6_equi.npz was produced by the reference's own method body, compiled out of
    train.py's syntax tree, and tests/test_oracle.py checks this restatement against it."""
    extend_data = []
    for state, mcts_prob, winner in play_data:
        for i in [1, 2, 3, 4]:
            equi_state = np.array([np.rot90(s, i) for s in state])
            equi_mcts_prob = np.rot90(np.flipud(mcts_prob.reshape(board_height, board_width)), i)
            extend_data.append((equi_state, np.flipud(equi_mcts_prob).flatten(), winner))
            equi_state = np.array([np.fliplr(s) for s in equi_state])
            equi_mcts_prob = np.fliplr(equi_mcts_prob)
            extend_data.append((equi_state, np.flipud(equi_mcts_prob).flatten(), winner))
    return extend_data
