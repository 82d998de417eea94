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


class ReplayBuffer:
    """deque(maxlen=buffer_size) of training tuples (train.py:82) kept as three ring arrays, with
    policy_update's uniform mini-batch draw (train.py:167): random.sample over the buffer uses
    the global `random` stream exactly as the reference does, so a seeded run picks the same
    positions of the buffer."""

    def __init__(self, maxlen, board_n):
        self.maxlen, self.N = int(maxlen), int(board_n)
        self.states = np.zeros((self.maxlen, 4, self.N, self.N), dtype=np.float32)
        self.pi = np.zeros((self.maxlen, self.N * self.N), dtype=np.float64)
        self.z = np.zeros(self.maxlen, dtype=np.float32)
        self.start = self.count = 0

    def __len__(self):
        return self.count

    def extend(self, states, pi, z):
        for i in range(int(np.shape(z)[0])):              # deque.extend: append one by one, oldest falls out
            j = (self.start + self.count) % self.maxlen
            if self.count == self.maxlen:
                self.start = (self.start + 1) % self.maxlen
            else:
                self.count += 1
            self.states[j], self.pi[j], self.z[j] = states[i], pi[i], z[i]

    def sample(self, batch_size):
        idx = random.sample(range(self.count), batch_size)
        phys = [(self.start + i) % self.maxlen for i in idx]
        return self.states[phys], self.pi[phys], self.z[phys]
