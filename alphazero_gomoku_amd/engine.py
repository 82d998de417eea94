"""SelfPlayEngine: numpy-facing wrapper of one libazgomoku handle (one GPU)."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import PRIOR_HASH, PRIOR_NET, PRIOR_PEAKED, PRIOR_UNIFORM32  # noqa: F401


def _ptr(a, typ=C.c_void_p):
    return None if a is None else C.cast(a.ctypes.data, typ) if typ is C.c_void_p else a.ctypes.data_as(typ)


class SelfPlayEngine:
    """B concurrent Gomoku self-play games searched by the HIP engine.

    Restates, batched over games, MCTSPlayer.get_action (mcts_alphaZero.py:212-260)
    and Game.start_self_play (game.py:360-416) of the reference."""

    def __init__(self, board_n=8, n_in_row=5, n_playout=400, n_games=1, c_puct=5.0, device=0,
                 prior_mode=PRIOR_NET, arena_nodes=0, cnn_mode=_lib.CNN_F16X3, big_arenas=0):
        self.L = _lib.load()
        self.N, self.S, self.B = int(board_n), int(board_n) ** 2, int(n_games)
        self.n_in_row, self.n_playout, self.c_puct = int(n_in_row), int(n_playout), float(c_puct)
        self.device, self.prior_mode = int(device), int(prior_mode)
        cfg = _lib.Config(self.N, self.n_in_row, self.n_playout, self.B, self.c_puct, self.device,
                          self.prior_mode, int(arena_nodes), int(cnn_mode), int(big_arenas))
        h = C.c_void_p()
        rc = self.L.azg_create(C.byref(cfg), C.byref(h))
        if rc != 0:
            raise _lib.AzgError(rc, (self.L.azg_last_error(None) or b"").decode())
        self.h = h
        self._weights_keepalive = None

    def close(self):
        if getattr(self, "h", None):
            self.L.azg_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        _lib.check(self.h, rc)

    # ---- weights -------------------------------------------------------------
    def set_weights(self, params, layout=_lib.LAYOUT_TORCH):
        """params: 16 float32 arrays (numpy) or torch tensors (CPU or this GPU)."""
        arr = (C.c_void_p * 16)()
        keep = []
        for i, p in enumerate(params):
            if hasattr(p, "data_ptr"):                       # torch tensor
                t = p.detach().contiguous().float()
                keep.append(t)
                arr[i] = t.data_ptr()
            else:
                a = np.ascontiguousarray(p, dtype=np.float32)
                keep.append(a)
                arr[i] = a.ctypes.data
        self._chk(self.L.azg_set_weights(self.h, arr, layout))
        self._weights_keepalive = None                        # the engine holds its own copy

    # ---- game control ----------------------------------------------------------
    def reset_games(self, seeds=None, mask=None, game_ids=None):
        seeds_a = None if seeds is None else np.ascontiguousarray(seeds, dtype=np.uint32)
        mask_a = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
        ids_a = None if game_ids is None else np.ascontiguousarray(game_ids, dtype=np.int64)
        for a in (seeds_a, mask_a, ids_a):
            assert a is None or a.shape == (self.B,)
        self._chk(self.L.azg_reset_games(
            self.h, _ptr(mask_a, C.POINTER(C.c_uint8)), _ptr(seeds_a, C.POINTER(C.c_uint32)),
            _ptr(ids_a, C.POINTER(C.c_int64))))

    def set_position(self, slot, states, last_move, current_player):
        """states: dict move -> player (the reference Board.states)."""
        w = [np.zeros(4, dtype=np.uint64), np.zeros(4, dtype=np.uint64)]
        for m, p in states.items():
            m = int(m)
            w[int(p) - 1][m >> 6] |= np.uint64(1) << np.uint64(m & 63)
        self._chk(self.L.azg_set_position(self.h, int(slot), w[0].ctypes.data_as(C.POINTER(C.c_uint64)),
                                          w[1].ctypes.data_as(C.POINTER(C.c_uint64)), int(last_move),
                                          int(current_player)))

    def search(self):
        self._chk(self.L.azg_search(self.h))

    def playout_begin(self, slot=0):
        """First half of one playout (select down to a leaf) -> (stones1, stones2, ply, last_move,
        current_player, terminal): the two stone sets as Python ints (bit m = cell m), terminal
        0 / 1 / 2 / -1 (game goes on, winner, tie)."""
        s1, s2 = (C.c_uint64 * 4)(), (C.c_uint64 * 4)()
        ply, last, cur, term = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        self._chk(self.L.azg_playout_begin(self.h, int(slot), s1, s2, C.byref(ply), C.byref(last), C.byref(cur),
                                           C.byref(term)))
        as_int = lambda w: sum(int(w[i]) << (64 * i) for i in range(4))
        return as_int(s1), as_int(s2), ply.value, last.value, cur.value, term.value

    def playout_end(self, priors, value, slot=0):
        """Second half: expand the leaf with priors (float32 [S] by cell, None for a terminal leaf)
        and back `value` up."""
        if priors is None:
            self._chk(self.L.azg_playout_end(self.h, int(slot), None, float(value)))
            return
        p = np.ascontiguousarray(priors, dtype=np.float32)
        assert p.shape == (self.S,)
        if not np.isfinite(p).all():
            raise ValueError("policy_value_fn returned a prior that is not finite")
        self._chk(self.L.azg_playout_end(self.h, int(slot), C.c_void_p(p.ctypes.data), float(value)))

    def pure_search(self, n_playout, mt_state):
        """The rollout opponent's move for every live slot (mcts_pure.py:214-231, 261-268).
        mt_state: uint32 [B, 625], per slot the 624 MT19937 key words and the position of
        np.random.get_state(); updated in place.  Returns the moves (int32 [B], -1 = not live)."""
        assert mt_state.dtype == np.uint32 and mt_state.shape == (self.B, 625) and mt_state.flags.c_contiguous
        moves = np.zeros(self.B, dtype=np.int32)
        self._chk(self.L.azg_pure_search(self.h, int(n_playout), mt_state.ctypes.data_as(C.POINTER(C.c_uint32)),
                                         moves.ctypes.data_as(C.POINTER(C.c_int32))))
        return moves

    def root_stats(self):
        visits = np.zeros((self.B, self.S), dtype=np.int32)
        q = np.zeros((self.B, self.S))
        p = np.zeros((self.B, self.S))
        rn = np.zeros(self.B, dtype=np.int32)
        self._chk(self.L.azg_root_stats(self.h, _ptr(visits), _ptr(q), _ptr(p), _ptr(rn)))
        return visits, q, p, rn

    def sample_and_advance(self, temp=1.0, want_pi=True):
        moves = np.zeros(self.B, dtype=np.int32)
        pi = np.zeros((self.B, self.S)) if want_pi else None
        self._chk(self.L.azg_sample_and_advance(
            self.h, float(temp), moves.ctypes.data_as(C.POINTER(C.c_int32)),
            None if pi is None else pi.ctypes.data_as(C.POINTER(C.c_double))))
        return moves, pi

    def advance(self, moves, record=False, temp=1.0):
        mv = np.ascontiguousarray(moves, dtype=np.int32)
        assert mv.shape == (self.B,)
        self._chk(self.L.azg_advance(self.h, mv.ctypes.data_as(C.POINTER(C.c_int32)), int(record), float(temp)))

    def update_with_move(self, moves):
        mv = np.ascontiguousarray(moves, dtype=np.int32)
        assert mv.shape == (self.B,)
        self._chk(self.L.azg_update_with_move(self.h, mv.ctypes.data_as(C.POINTER(C.c_int32))))

    def status(self):
        ended = np.zeros(self.B, dtype=np.uint8)
        winner = np.zeros(self.B, dtype=np.int8)
        plies = np.zeros(self.B, dtype=np.int32)
        self._chk(self.L.azg_status(self.h, ended.ctypes.data_as(C.POINTER(C.c_uint8)),
                                    winner.ctypes.data_as(C.POINTER(C.c_int8)),
                                    plies.ctypes.data_as(C.POINTER(C.c_int32))))
        return ended.astype(bool), winner, plies

    def states(self):
        out = np.zeros((self.B, 4, self.N, self.N), dtype=np.float32)
        self._chk(self.L.azg_states(self.h, _ptr(out)))
        return out

    def pending_samples(self, ended=None, plies=None):
        """Upper bound of the samples azg_drain_samples would return now."""
        if ended is None:
            ended, _, plies = self.status()
        return int(plies[ended].sum())

    def drain_samples(self, max_samples=None):
        """-> dict(states f32 [n,4,N,N], pi f64 [n,S], z f32 [n], game_id i64 [n], ply i32 [n])."""
        if max_samples is None:
            max_samples = max(1, self.pending_samples())
        st = np.zeros((max_samples, 4, self.N, self.N), dtype=np.float32)
        pi = np.zeros((max_samples, self.S))
        z = np.zeros(max_samples, dtype=np.float32)
        gid = np.zeros(max_samples, dtype=np.int64)
        ply = np.zeros(max_samples, dtype=np.int32)
        n = C.c_int(0)
        self._chk(self.L.azg_drain_samples(self.h, int(max_samples), _ptr(st), _ptr(pi), _ptr(z), _ptr(gid),
                                           _ptr(ply), C.byref(n)))
        k = n.value
        return dict(states=st[:k], pi=pi[:k], z=z[:k], game_id=gid[:k], ply=ply[:k])

    def drain_samples_device(self, states, pi, z, game_id, ply):
        """Same, into preallocated torch CUDA tensors; returns the sample count."""
        n = C.c_int(0)
        self._chk(self.L.azg_drain_samples(self.h, int(z.shape[0]), C.c_void_p(states.data_ptr()),
                                           C.c_void_p(pi.data_ptr()), C.c_void_p(z.data_ptr()),
                                           C.c_void_p(game_id.data_ptr()), C.c_void_p(ply.data_ptr()),
                                           C.byref(n)))
        return n.value

    # ---- pipelined self-play (azg_selfplay_*) --------------------------------------
    def selfplay_begin(self, temp=1.0, seed0=0, first_game_id=0, game_id_stride=1, total=None):
        self._chk(self.L.azg_selfplay_begin(self.h, float(temp), int(seed0) & 0xFFFFFFFF, int(first_game_id),
                                            int(game_id_stride), -1 if total is None else int(total)))

    def selfplay_step(self):
        """One move for every live slot -> (games finished in this step, samples ready to fetch)."""
        fin, ready = C.c_int(0), C.c_int(0)
        self._chk(self.L.azg_selfplay_step(self.h, C.byref(fin), C.byref(ready)))
        return fin.value, ready.value

    def selfplay_end(self):
        ready = C.c_int(0)
        self._chk(self.L.azg_selfplay_end(self.h, C.byref(ready)))
        return ready.value

    def selfplay_fetch(self, n, device=False, packed=False):
        """The n ready samples as a dict of numpy arrays (or torch CUDA tensors with device=True).
        packed=True (device only): `stones` u64-as-int64 [n,2,4] and `last` i32 [n] instead of the
        feature planes `states` (see unpack_states)."""
        if device:
            import torch
            dev = torch.device("cuda", self.device)
            t = dict(pi=torch.empty((n, self.S), dtype=torch.float64, device=dev),
                     z=torch.empty(n, dtype=torch.float32, device=dev),
                     game_id=torch.empty(n, dtype=torch.int64, device=dev),
                     ply=torch.empty(n, dtype=torch.int32, device=dev))
            if packed:
                t["stones"] = torch.empty((n, 2, 4), dtype=torch.int64, device=dev)
                t["last"] = torch.empty(n, dtype=torch.int32, device=dev)
                keys = ("stones", "last", "pi", "z", "game_id", "ply")
                fn = self.L.azg_selfplay_fetch_packed
            else:
                t["states"] = torch.empty((n, 4, self.N, self.N), dtype=torch.float32, device=dev)
                keys = ("states", "pi", "z", "game_id", "ply")
                fn = self.L.azg_selfplay_fetch
            if n:
                torch.cuda.current_stream(dev).synchronize()          # the buffers exist before the engine's stream writes them
                self._chk(fn(self.h, *[C.c_void_p(t[k].data_ptr()) for k in keys]))
            return t
        t = dict(states=np.zeros((n, 4, self.N, self.N), dtype=np.float32), pi=np.zeros((n, self.S)),
                 z=np.zeros(n, dtype=np.float32), game_id=np.zeros(n, dtype=np.int64), ply=np.zeros(n, dtype=np.int32))
        if n:
            self._chk(self.L.azg_selfplay_fetch(self.h, *[_ptr(t[k]) for k in ("states", "pi", "z", "game_id", "ply")]))
        return t

    def eval_batch(self, states):
        st = np.ascontiguousarray(states, dtype=np.float32).reshape(-1, 4, self.N, self.N)
        n = st.shape[0]
        probs = np.zeros((n, self.S), dtype=np.float32)
        vals = np.zeros(n, dtype=np.float32)
        self._chk(self.L.azg_eval_batch(self.h, _ptr(st), n, _ptr(probs), _ptr(vals)))
        return probs, vals

    def unpack_states(self, stones, last, ply):
        """Board.current_state planes (torch CUDA f32 [n,4,N,N]) of packed self-play positions:
        stones int64 [n,2,4] (bit sets of stone 1 / stone 2), last i32 [n], ply i32 [n]."""
        import torch
        n = int(ply.shape[0])
        out = torch.empty((n, 4, self.N, self.N), dtype=torch.float32, device=stones.device)
        if n:
            torch.cuda.current_stream(stones.device).synchronize()
            self._chk(self.L.azg_unpack_states(self.h, n, C.c_void_p(stones.contiguous().data_ptr()),
                                               C.c_void_p(last.contiguous().data_ptr()), C.c_void_p(ply.contiguous().data_ptr()),
                                               C.c_void_p(out.data_ptr())))
        return out

    def eval_batch_device(self, states):
        """eval_batch on a torch CUDA float32 tensor [n,4,N,N] of the engine's device, results as
        torch tensors on that device (no host copy, no staging)."""
        import torch
        st = states.contiguous()
        n = int(st.shape[0])
        probs = torch.empty((n, self.S), dtype=torch.float32, device=st.device)
        vals = torch.empty(n, dtype=torch.float32, device=st.device)
        if n:
            torch.cuda.current_stream(st.device).synchronize()
            self._chk(self.L.azg_eval_batch(self.h, C.c_void_p(st.data_ptr()), n, C.c_void_p(probs.data_ptr()),
                                            C.c_void_p(vals.data_ptr())))
        return probs, vals

    def board_step_batch(self, stones, ply, last_move, moves, player, want_planes=True, static=False):
        """In-place Board.do_move + game_end (+ current_state) on n boards.  static=True goes
        through azg_board_step_batch_static (no device allocation, at most 1024 boards)."""
        n = int(ply.shape[0])
        assert stones.dtype == np.uint64 and stones.shape == (n, 2, 4) and stones.flags.c_contiguous
        assert ply.dtype == np.int32 and last_move.dtype == np.int32
        mv = np.ascontiguousarray(moves, dtype=np.int32)
        pl = np.ascontiguousarray(player, dtype=np.int32)
        ended = np.zeros(n, dtype=np.uint8)
        winner = np.zeros(n, dtype=np.int8)
        planes = np.zeros((n, 4, self.N, self.N), dtype=np.float32) if want_planes else None
        i32 = C.POINTER(C.c_int32)
        fn = self.L.azg_board_step_batch_static if static else self.L.azg_board_step_batch
        self._chk(fn(
            self.h, n, stones.ctypes.data_as(C.POINTER(C.c_uint64)), ply.ctypes.data_as(i32),
            last_move.ctypes.data_as(i32), mv.ctypes.data_as(i32), pl.ctypes.data_as(i32),
            ended.ctypes.data_as(C.POINTER(C.c_uint8)), winner.ctypes.data_as(C.POINTER(C.c_int8)),
            None if planes is None else planes.ctypes.data_as(C.POINTER(C.c_float))))
        return ended.astype(bool), winner, planes

    def search_parts(self):
        """Kernel launches a move's playouts are spread over (azg_search_parts)."""
        return int(self.L.azg_search_parts(self.h))

    def set_search_parts(self, parts):
        """Spread a move's playouts over `parts` launches (0: the library's default); same trees."""
        self._chk(self.L.azg_set_search_parts(self.h, int(parts)))

    def profile(self):
        """(search launches, summed ms on the engine's stream, summed leaf evaluations) since last call."""
        a, b, c = C.c_int64(0), C.c_double(0.0), C.c_int64(0)
        self._chk(self.L.azg_profile(self.h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def arena_stats(self):
        """(games moved into a large tree arena, trees dropped because none was free) since creation."""
        a, b = C.c_int64(0), C.c_int64(0)
        self._chk(self.L.azg_arena_stats(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def reroot_stats(self):
        """(re-roots that left the kept subtree in place, re-roots that compacted it) since creation."""
        a, b = C.c_int64(0), C.c_int64(0)
        self._chk(self.L.azg_reroot_stats(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def gate_stats(self):
        """(waits for a CU's matrix gate that ran out since the process started on this GPU, gate
        words set right now).  The gate only schedules (see azg_gate_stats); a count > 0 means slower
        launches, never different results."""
        a, b = C.c_int64(0), C.c_int64(0)
        self._chk(self.L.azg_gate_stats(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def gate_poke(self, value):
        """Diagnostic: store `value` in every word of this GPU's gate table."""
        self._chk(self.L.azg_gate_poke(self.h, int(value)))

    def host_timing(self):
        """(host milliseconds spent drawing exploration noise, move steps) since the last call."""
        a, b = C.c_double(0.0), C.c_int64(0)
        self._chk(self.L.azg_host_timing(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def stream(self):
        return self.L.azg_stream(self.h)

    # ---- batched self-play -----------------------------------------------------
    def play_games(self, n_games_total, temp=1.0, seed0=0, first_game_id=0, game_id_stride=1,
                   on_samples=None, max_moves=None, pipelined=True):
        """Play `n_games_total` self-play games through the B slots, refilling a slot as
        soon as its game ends (game k has id = first_game_id + k * game_id_stride and draws
        its noise from numpy's legacy RandomState seeded with seed0 + id).  Returns
        dict(games, plies, move_steps, samples); samples is the concatenated drain output
        (None if on_samples consumed it)."""
        collected = []
        sink = on_samples if on_samples is not None else collected.append
        run = SelfPlayRunner(self, temp, seed0, first_game_id, game_id_stride, n_games_total, sink,
                             pipelined=pipelined)
        run.start()
        try:
            while run.games_done < n_games_total and (max_moves is None or run.steps < max_moves):
                run.step()
        finally:
            run.finish()                                     # leaves the handle usable after an error, too
        samples = None
        if collected:
            samples = {k: np.concatenate([c[k] for c in collected]) for k in collected[0]}
        return dict(games=run.games_done, plies=run.plies_done, move_steps=run.steps, samples=samples)


class SelfPlayRunner:
    """Keeps the B slots of an engine busy: every step() is one move for every live game
    (MCTSPlayer.get_action + the per-move part of Game.start_self_play); finished games are
    drained and their slots restarted with the next game id until `total` games have been
    started (total=None: without limit).

    pipelined=True uses the engine's azg_selfplay_* calls: finished slots restart on the device
    and the next search is already running while the host handles the finished games, whose
    samples reach on_samples one step later (call finish() for the last ones).  The games
    played are the same either way; which slot plays which game may differ."""

    def __init__(self, engine, temp=1.0, seed0=0, first_game_id=0, game_id_stride=1, total=None,
                 on_samples=None, device_samples=False, pipelined=False, packed_samples=False):
        self.e, self.temp, self.seed0 = engine, float(temp), int(seed0)
        self.first, self.stride, self.total = int(first_game_id), int(game_id_stride), total
        self.on_samples, self.device_samples = on_samples, device_samples
        self.pipelined, self._running = bool(pipelined), False
        self.packed_samples = bool(packed_samples)          # pipelined + device_samples only
        B = engine.B
        self.ids = np.zeros(B, dtype=np.int64)
        self.seeds = np.zeros(B, dtype=np.uint32)
        self.live = np.zeros(B, dtype=bool)
        self.next_k = 0
        # games_done counts a game in the step it ends; games_delivered and plies_done count it
        # when its samples reach on_samples (pipelined: one step later), so plies_done /
        # games_delivered is a mean over whole games
        self.games_done = self.games_delivered = self.plies_done = self.steps = 0
        self._undelivered = 0

    def _assign(self, slots):
        mask = np.zeros(self.e.B, dtype=np.uint8)
        for g in slots:
            if self.total is not None and self.next_k >= self.total:
                self.live[g] = False
                continue
            self.ids[g] = self.first + self.next_k * self.stride
            self.seeds[g] = (self.seed0 + int(self.ids[g])) & 0xFFFFFFFF
            mask[g] = 1
            self.live[g] = True
            self.next_k += 1
        if mask.any():
            self.e.reset_games(seeds=self.seeds, mask=mask, game_ids=self.ids)

    def start(self):
        if self.pipelined:
            self.e.selfplay_begin(self.temp, self.seed0, self.first, self.stride, self.total)
            self._running = True
        else:
            self._assign(range(self.e.B))

    def _deliver(self, n):
        self.games_delivered += self._undelivered
        self._undelivered = 0
        if n:
            out = self.e.selfplay_fetch(n, device=self.device_samples, packed=self.packed_samples)
            self.plies_done += n
            if self.on_samples is not None:
                self.on_samples(out)

    def finish(self):
        """Pipelined mode: wait for the work in flight and deliver the last samples."""
        if self._running:
            self._running = False
            self._deliver(self.e.selfplay_end())

    def step(self):
        """One move for every live slot; returns (games finished, samples drained)."""
        e = self.e
        if self.pipelined:
            nfin, ready = e.selfplay_step()
            self.steps += 1
            self.games_done += nfin
            self._deliver(ready)
            self._undelivered = nfin
            return nfin, ready
        e.search()
        e.sample_and_advance(self.temp, want_pi=False)
        self.steps += 1
        ended, _, plies = e.status()
        fin = ended & self.live
        nfin = int(fin.sum())
        nsamp = 0
        if nfin:
            nsamp = int(plies[fin].sum())
            if self.device_samples:
                import torch
                dev = torch.device("cuda", e.device)
                t = dict(states=torch.empty((nsamp, 4, e.N, e.N), dtype=torch.float32, device=dev),
                         pi=torch.empty((nsamp, e.S), dtype=torch.float64, device=dev),
                         z=torch.empty(nsamp, dtype=torch.float32, device=dev),
                         game_id=torch.empty(nsamp, dtype=torch.int64, device=dev),
                         ply=torch.empty(nsamp, dtype=torch.int32, device=dev))
                k = e.drain_samples_device(t["states"], t["pi"], t["z"], t["game_id"], t["ply"])
                out = {name: v[:k] for name, v in t.items()}
            else:
                out = e.drain_samples(max(1, nsamp))
            if self.on_samples is not None:
                self.on_samples(out)
            self.games_done += nfin
            self.games_delivered += nfin
            self.plies_done += nsamp
            self._assign(np.nonzero(fin)[0])
        return nfin, nsamp
