"""Training-data side of the self-play loop: dihedral augmentation and the replay buffer
(TrainPipeline.get_equi_data / data_buffer / policy_update's sampling, train.py:82,116-135,167)."""
import ctypes as C
import random

import numpy as np

from . import _lib


def equi_data_arrays(engine, states, pi, z):
    """8-fold augmentation of n tuples on the engine's GPU: states f32 [n,4,N,N], pi f64 [n,S],
    z f32 [n] (numpy arrays, or torch CUDA tensors of the engine's device) -> the same three with
    8n rows, tuple i giving rows 8i..8i+7 in the order of train.py:121-134."""
    n = int(z.shape[0])
    if hasattr(z, "data_ptr"):
        import torch
        so = torch.empty((8 * n,) + tuple(states.shape[1:]), dtype=torch.float32, device=states.device)
        po = torch.empty((8 * n, pi.shape[1]), dtype=torch.float64, device=pi.device)
        zo = torch.empty(8 * n, dtype=torch.float32, device=z.device)
        ptr = lambda t: C.c_void_p(t.data_ptr())
        args = [states.contiguous(), pi.contiguous(), z.contiguous(), so, po, zo]
    else:
        states = np.ascontiguousarray(states, dtype=np.float32)
        pi = np.ascontiguousarray(pi, dtype=np.float64)
        z = np.ascontiguousarray(z, dtype=np.float32)
        so = np.zeros((8 * n,) + states.shape[1:], dtype=np.float32)
        po = np.zeros((8 * n, pi.shape[1]), dtype=np.float64)
        zo = np.zeros(8 * n, dtype=np.float32)
        ptr = lambda a: C.c_void_p(a.ctypes.data)
        args = [states, pi, z, so, po, zo]
    if hasattr(z, "data_ptr"):
        torch.cuda.current_stream(z.device).synchronize()    # inputs are complete before the engine's stream reads them
    if n:
        _lib.check(engine.h, engine.L.azg_equi_data(engine.h, n, *[ptr(a) for a in args]))
    return so, po, zo


def get_equi_data(engine, play_data):
    """Drop-in for TrainPipeline.get_equi_data (train.py:116-135): list of (state [4,H,W],
    mcts_prob [S], winner) -> list of 8x as many tuples, same order and values."""
    play_data = list(play_data)
    if not play_data:
        return []
    states = np.stack([np.asarray(d[0], dtype=np.float32) for d in play_data])
    pi = np.stack([np.asarray(d[1], dtype=np.float64) for d in play_data])
    z = np.asarray([d[2] for d in play_data], dtype=np.float32)
    so, po, zo = equi_data_arrays(engine, states, pi, z)
    return [(so[i], po[i], zo[i]) for i in range(so.shape[0])]


class DeviceReplayBuffer:
    """The same ring (deque(maxlen) + random.sample over it, train.py:82,155,167) kept in GPU
    memory: tuples arrive as torch CUDA tensors from the augmentation kernel and leave as the
    mini-batch train_step consumes, without touching the host.  Only the mini-batch INDICES are
    drawn on the host, by random.sample on `rng` (a random.Random, or the `random` module for the
    reference's global stream), so a seeded run picks the same buffer positions as ReplayBuffer."""

    def __init__(self, maxlen, board_n, device, rng=random):
        import torch
        self.maxlen, self.N, self.rng = int(maxlen), int(board_n), rng
        self.states = torch.zeros((self.maxlen, 4, self.N, self.N), dtype=torch.uint8, device=device)   # planes are 0/1
        self.pi = torch.zeros((self.maxlen, self.N * self.N), dtype=torch.float64, device=device)
        self.z = torch.zeros(self.maxlen, dtype=torch.float32, device=device)
        self.start = self.count = 0

    def __len__(self):
        return self.count

    def extend(self, states, pi, z):
        import torch
        k = int(z.shape[0])
        if k > self.maxlen:                                  # only the newest maxlen survive a deque.extend
            states, pi, z = states[k - self.maxlen:], pi[k - self.maxlen:], z[k - self.maxlen:]
            self.start = (self.start + self.count + k - self.maxlen) % self.maxlen
            self.count, k = 0, self.maxlen
        first = (self.start + self.count) % self.maxlen
        j = (first + torch.arange(k, device=self.z.device)) % self.maxlen
        self.states[j] = states.to(torch.uint8)
        self.pi[j] = pi.to(torch.float64)
        self.z[j] = z.to(torch.float32)
        over = max(0, self.count + k - self.maxlen)          # that many of the oldest fall out
        self.start = (self.start + over) % self.maxlen
        self.count = min(self.maxlen, self.count + k)

    def draw(self, batch_size):
        """Logical positions (0 = oldest) of a mini-batch: random.sample(range(len), batch_size)."""
        return self.rng.sample(range(self.count), batch_size)

    def take(self, idx):
        """(states f32, pi f32, z f32) CUDA tensors of the tuples at logical positions idx."""
        import torch
        phys = (torch.as_tensor(idx, dtype=torch.int64, device=self.z.device) + self.start) % self.maxlen
        return self.states[phys].to(torch.float32), self.pi[phys].to(torch.float32), self.z[phys]

    def sample(self, batch_size):
        return self.take(self.draw(batch_size))


class ReplayBuffer:
    """deque(maxlen=buffer_size) of training tuples (train.py:82) kept as three ring arrays, with
    policy_update's uniform mini-batch draw (train.py:167): random.sample over the buffer uses
    the global `random` stream exactly as the reference does, so a seeded run picks the same
    positions of the buffer."""

    def __init__(self, maxlen, board_n, rng=random):
        self.maxlen, self.N, self.rng = int(maxlen), int(board_n), rng
        self.states = np.zeros((self.maxlen, 4, self.N, self.N), dtype=np.float32)
        self.pi = np.zeros((self.maxlen, self.N * self.N), dtype=np.float64)
        self.z = np.zeros(self.maxlen, dtype=np.float32)
        self.start = self.count = 0

    def __len__(self):
        return self.count

    def extend(self, states, pi, z):
        """deque.extend: the tuples are appended in order and the oldest fall out."""
        states, pi, z = np.asarray(states), np.asarray(pi), np.asarray(z)
        k = int(z.shape[0])
        if k > self.maxlen:
            states, pi, z = states[k - self.maxlen:], pi[k - self.maxlen:], z[k - self.maxlen:]
            self.start = (self.start + self.count + k - self.maxlen) % self.maxlen
            self.count, k = 0, self.maxlen
        j = (self.start + self.count + np.arange(k)) % self.maxlen
        self.states[j], self.pi[j], self.z[j] = states, pi, z
        over = max(0, self.count + k - self.maxlen)
        self.start = (self.start + over) % self.maxlen
        self.count = min(self.maxlen, self.count + k)

    def draw(self, batch_size):
        return self.rng.sample(range(self.count), batch_size)

    def take(self, idx):
        phys = [(self.start + i) % self.maxlen for i in idx]
        return self.states[phys], self.pi[phys], self.z[phys]

    def sample(self, batch_size):
        return self.take(self.draw(batch_size))
