"""PolicyValueNet with the reference's surface (policy_value_net_pytorch.py:60-157).

PyTorch only holds the parameters: every forward used by search or by policy_value /
policy_value_fn runs in the HIP engine's CNN kernel from a packed copy of them, and train_step is
the library's own float32 forward / backward / Adam (azg_train_*, csrc/azg_train.hpp) on the
parameter tensors in place.  The PyTorch autograd version of the same step (train_step_tensors)
is kept as the float32 reference the tests compare against and for CPU tensors."""
import ctypes as C

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib, model_files
from .engine import SelfPlayEngine

PARAM_ORDER = ["conv1.weight", "conv1.bias", "conv2.weight", "conv2.bias", "conv3.weight", "conv3.bias",
               "act_conv1.weight", "act_conv1.bias", "act_fc1.weight", "act_fc1.bias",
               "val_conv1.weight", "val_conv1.bias", "val_fc1.weight", "val_fc1.bias",
               "val_fc2.weight", "val_fc2.bias"]


def theano_to_torch_layout(params):
    """The 16-tensor list of the reference's Theano/Lasagne and numpy nets
    (policy_value_net_numpy.py:103-113: convolution kernels stored for a true convolution, i.e.
    rotated by 180 degrees, fully connected matrices as (in, out)) -> the PyTorch state_dict
    layout of PARAM_ORDER.  The mapping is its own inverse up to the transposes."""
    out = []
    for p in params:
        p = np.asarray(p, dtype=np.float32)
        if p.ndim == 4:
            p = p[:, :, ::-1, ::-1]
        elif p.ndim == 2:
            p = p.T
        out.append(np.array(p, dtype=np.float32, order="C", copy=True))
    return out


torch_to_theano_layout = theano_to_torch_layout          # flips and transposes undo themselves


class Net(nn.Module):
    """Parameter container with the reference's module names, so state_dicts interchange
    (policy_value_net_pytorch.py:23-41).  forward() exists for autograd in train_step."""

    def __init__(self, board_width, board_height):
        super().__init__()
        s = board_width * board_height
        self.board_width, self.board_height = board_width, board_height
        self.conv1 = nn.Conv2d(4, 32, 3, padding=1)
        self.conv2 = nn.Conv2d(32, 64, 3, padding=1)
        self.conv3 = nn.Conv2d(64, 128, 3, padding=1)
        self.act_conv1 = nn.Conv2d(128, 4, 1)
        self.act_fc1 = nn.Linear(4 * s, s)
        self.val_conv1 = nn.Conv2d(128, 2, 1)
        self.val_fc1 = nn.Linear(2 * s, 64)
        self.val_fc2 = nn.Linear(64, 1)

    def forward(self, x):
        s = self.board_width * self.board_height
        x = F.relu(self.conv3(F.relu(self.conv2(F.relu(self.conv1(x))))))
        a = F.relu(self.act_conv1(x)).reshape(-1, 4 * s)
        a = F.log_softmax(self.act_fc1(a), dim=1)
        v = F.relu(self.val_conv1(x)).reshape(-1, 2 * s)
        v = torch.tanh(self.val_fc2(F.relu(self.val_fc1(v))))
        return a, v


def train_step_tensors(module, optimizer, s, p, z, lr, reduce_fn=None):
    """policy_value_net_pytorch.py:117-148 on torch tensors of any device: loss = MSE(v, z) -
    mean(sum(pi * log p)), one optimiser step at learning rate lr; reduce_fn(module), if given, runs
    between backward and the step (gradient averaging over ranks).  Returns (loss, entropy) as
    0-dim tensors."""
    optimizer.zero_grad()
    for group in optimizer.param_groups:
        group['lr'] = lr
    logp, v = module(s)
    value_loss = F.mse_loss(v.view(-1), z)
    policy_loss = -torch.mean(torch.sum(p * logp, 1))
    loss = value_loss + policy_loss
    loss.backward()
    if reduce_fn is not None:
        reduce_fn(module)
    optimizer.step()
    entropy = -torch.mean(torch.sum(torch.exp(logp) * logp, 1))
    return loss.detach(), entropy.detach()


class PolicyValueNet(object):
    def __init__(self, board_width, board_height, model_file=None, use_gpu=True, device=0, native_train=True):
        if board_width != board_height:
            raise ValueError("only square boards are supported by the HIP engine")
        if not torch.cuda.is_available():
            raise RuntimeError("PolicyValueNet needs a GPU: the forward pass runs in the HIP engine")
        self.use_gpu = True
        self.board_width, self.board_height = board_width, board_height
        self.l2_const = 1e-4
        self.device = torch.device("cuda", device)
        self.policy_value_net = Net(board_width, board_height).to(self.device)
        self.optimizer = torch.optim.Adam(self.policy_value_net.parameters(), weight_decay=self.l2_const)
        self.version = 0          # bumped whenever the parameters change
        if model_file:
            self.load_model(model_file)
        self.data_parallel = False   # True: train_step averages gradients over torch.distributed ranks
        # native_train: train_step runs in the HIP library; False: PyTorch autograd + torch.optim.Adam
        # (the two keep separate optimiser states: use one or the other on a net)
        self.native_train = bool(native_train)
        self._adam = None            # (grad, m, v) flat float32 vectors in state_dict order, step count
        self._ptr_cache = None
        self._adam_step = 0
        self._eval_engine = None
        self._eval_version = -1

    # ---- parameters ----------------------------------------------------------
    def param_tensors(self):
        sd = self.policy_value_net.state_dict()
        return [sd[k] for k in PARAM_ORDER]

    def load_numpy_params(self, params):
        """16 float32 arrays in the state_dict order above."""
        sd = self.policy_value_net.state_dict()
        for k, p in zip(PARAM_ORDER, params):
            # a real copy: a flipped 1x1 kernel counts as contiguous for numpy but keeps negative strides
            sd[k].copy_(torch.from_numpy(np.array(p, dtype=np.float32, order="C", copy=True)).reshape(sd[k].shape))
        self.version += 1

    def zero_head_biases(self):
        """Set the biases of the two 1x1 head convolutions (4 policy channels, 2 value channels) to
        zero.  A freshly initialised trunk puts out |a3| of about 0.02 while PyTorch draws these
        biases from U(-0.088, 0.088), so at the start every head channel is switched on or off for
        ALL positions by the sign of its bias alone: one net in four starts with both value channels
        dead, and a few large early optimiser steps can push the remaining ones under as well, after
        which the heads pass no gradient and the net never learns.  With zero biases the sign follows
        the position and about half of every channel is active."""
        with torch.no_grad():
            self.policy_value_net.act_conv1.bias.zero_()
            self.policy_value_net.val_conv1.bias.zero_()
        self.version += 1

    def load_model(self, model_file, layout=None):
        """A PyTorch state_dict file (what save_model and the reference's PyTorch net write,
        policy_value_net_pytorch.py:150-157), an .npz archive of the 16 parameter arrays
        (arr_0..arr_15) in the PyTorch layout or, with layout='theano' (or a 'layout' entry in
        the archive saying so), in the layout of the reference's Theano/numpy model files, or such
        a model file itself: the pickled list of 16 arrays that policy_value_net.py:113-121 writes
        and policy_value_net_numpy.py:87-90 reads (Theano layout).
        Files are read with loaders that execute nothing from them: weights_only=True,
        allow_pickle=False, and for the pickled lists model_files.read_pickled_arrays, which only
        disassembles the file and refuses anything but plain float arrays."""
        if not str(model_file).endswith(".npz") and model_files.looks_like_pickle(model_file):
            try:
                params = model_files.read_pickled_arrays(model_file)
            except ValueError:
                params = None                      # e.g. a state_dict in torch's legacy pickle format: below
            if params is not None:
                if len(params) != 16:
                    raise ValueError("%s holds %d arrays, a model of this net has 16" % (model_file, len(params)))
                self.load_numpy_params(theano_to_torch_layout(params) if layout in (None, "theano") else params)
                return
        if str(model_file).endswith(".npz"):
            with np.load(model_file, allow_pickle=False) as z:
                if layout is None and "layout" in z.files:
                    layout = str(z["layout"])
                params = [z["arr_%d" % i] for i in range(16)]
            if layout == "theano":
                params = theano_to_torch_layout(params)
            elif layout not in (None, "torch"):
                raise ValueError("layout must be 'torch' or 'theano'")
            self.load_numpy_params(params)
        else:
            self.policy_value_net.load_state_dict(torch.load(model_file, map_location=self.device, weights_only=True))
            self.version += 1

    def save_numpy(self, path, layout="torch"):
        """The 16 parameter arrays as an .npz archive (see load_model); layout='theano' writes what
        policy_value_net_numpy.PolicyValueNetNumpy consumes."""
        params = [t.detach().cpu().numpy() for t in self.param_tensors()]
        if layout == "theano":
            params = torch_to_theano_layout(params)
        elif layout != "torch":
            raise ValueError("layout must be 'torch' or 'theano'")
        np.savez(path, *params, layout=layout)

    def sync_engine(self, engine, have_version):
        """Push the parameters into `engine` if it holds an older copy; returns the version."""
        if have_version != self.version:
            engine.set_weights(self.param_tensors(), _lib.LAYOUT_TORCH)
        return self.version

    def _engine_handle(self):
        """The net's own library handle (training calls work on the parameter tensors themselves and
        need no packed copy)."""
        if self._eval_engine is None:
            self._eval_engine = SelfPlayEngine(self.board_width, min(5, self.board_width), 1, 1,
                                               device=self.device.index or 0, arena_nodes=8)
        return self._eval_engine

    def _engine(self):
        """The same handle with its packed copy of the parameters brought up to date (policy_value, policy_value_fn)."""
        eng = self._engine_handle()
        self._eval_version = self.sync_engine(eng, self._eval_version)
        return eng

    # ---- reference surface -----------------------------------------------------
    def policy_value(self, state_batch):
        """policy_value_net_pytorch.py:80-94: (probs [B,S], value [B,1]) as numpy; a torch CUDA
        tensor stays on the device and gives torch tensors."""
        if torch.is_tensor(state_batch) and state_batch.is_cuda:
            probs, vals = self._engine().eval_batch_device(state_batch.to(torch.float32))
            return probs, vals.reshape(-1, 1)
        probs, vals = self._engine().eval_batch(np.asarray(state_batch, dtype=np.float32))
        return probs, vals.reshape(-1, 1)

    def _param_ptrs(self):
        """The 16 parameter tensors' device pointers (PyTorch layout) for the azg_train_* calls; the
        tensors are the module's own storage, so an update in place is the module's update."""
        if self._ptr_cache is None:                          # load_* copy into the same storage: the pointers stay
            ts = [p.data for p in (dict(self.policy_value_net.named_parameters())[k] for k in PARAM_ORDER)]
            self._ptr_cache = ((C.c_void_p * 16)(*[t.data_ptr() for t in ts]), ts)
        return self._ptr_cache

    def policy_value_in_training(self, state_batch, old_probs=None):
        """(probs [B,S], value [B,1]) of a CUDA batch for the KL monitor inside policy_update, from the
        parameters as they are NOW (between two train steps of one update the search engine's packed
        copy is not refreshed): the library's float32 forward pass, azg_train_forward.
        With old_probs (CUDA f32 [B,S]) a third value: the KL test's number of train.py:189-193,
        mean(sum(old * (log(old + 1e-10) - log(new + 1e-10)), 1)), summed in double."""
        st = state_batch.to(device=self.device, dtype=torch.float32).contiguous()
        if not self.native_train:
            with torch.no_grad():
                logp, v = self.policy_value_net(st)
            probs = torch.exp(logp)
            if old_probs is None:
                return probs, v.reshape(-1, 1)
            o = old_probs.double()
            kl = float(torch.mean(torch.sum(o * (torch.log(o + 1e-10) - torch.log(probs.double() + 1e-10)), dim=1)))
            return probs, v.reshape(-1, 1), kl
        n, s = int(st.shape[0]), self.board_width * self.board_height
        probs = torch.empty((n, s), dtype=torch.float32, device=self.device)
        vals = torch.empty(n, dtype=torch.float32, device=self.device)
        eng = self._engine_handle()
        arr, keep = self._param_ptrs()
        old = None if old_probs is None else old_probs.to(device=self.device, dtype=torch.float32).contiguous()
        kl = C.c_double(0.0)
        torch.cuda.current_stream(self.device).synchronize()
        eng._chk(eng.L.azg_train_forward(eng.h, n, C.c_void_p(st.data_ptr()), arr, C.c_void_p(probs.data_ptr()),
                                         C.c_void_p(vals.data_ptr()), C.c_void_p(old.data_ptr()) if old is not None else None,
                                         C.byref(kl) if old is not None else None))
        if old is None:
            return probs, vals.reshape(-1, 1)
        return probs, vals.reshape(-1, 1), float(kl.value)

    def policy_value_fn(self, board):
        """policy_value_net_pytorch.py:96-115: (zip(legal moves, probs), value)."""
        legal = board.availables
        st = np.ascontiguousarray(board.current_state().reshape(-1, 4, self.board_width, self.board_height))
        probs, vals = self._engine().eval_batch(st.astype(np.float32))
        return zip(legal, probs[0][legal]), float(vals[0])

    def native_gradients(self, state_batch, mcts_probs, winner_batch):
        """(flat gradient in state_dict order as a new CUDA tensor, loss, entropy) of the library's
        backward pass on a batch, without touching the parameters (tests; gradient inspection)."""
        dev = self.device
        s = torch.as_tensor(np.asarray(state_batch), dtype=torch.float32, device=dev).contiguous() if not torch.is_tensor(state_batch) \
            else state_batch.to(device=dev, dtype=torch.float32).contiguous()
        p = torch.as_tensor(np.asarray(mcts_probs), dtype=torch.float32, device=dev).contiguous() if not torch.is_tensor(mcts_probs) \
            else mcts_probs.to(device=dev, dtype=torch.float32).contiguous()
        z = torch.as_tensor(np.asarray(winner_batch), dtype=torch.float32, device=dev).contiguous().reshape(-1) if not torch.is_tensor(winner_batch) \
            else winner_batch.to(device=dev, dtype=torch.float32).contiguous().reshape(-1)
        eng = self._engine_handle()
        grad = torch.zeros(int(eng.L.azg_train_param_count(eng.h)), dtype=torch.float32, device=dev)
        arr, keep = self._param_ptrs()
        loss, entropy = C.c_float(0.0), C.c_float(0.0)
        torch.cuda.current_stream(dev).synchronize()
        eng._chk(eng.L.azg_train_backward(eng.h, int(s.shape[0]), C.c_void_p(s.data_ptr()), C.c_void_p(p.data_ptr()),
                                          C.c_void_p(z.data_ptr()), arr, C.c_void_p(grad.data_ptr()), C.byref(loss), C.byref(entropy)))
        return grad, float(loss.value), float(entropy.value)

    def train_step(self, state_batch, mcts_probs, winner_batch, lr):
        """policy_value_net_pytorch.py:117-148 (loss = MSE(v,z) - mean(pi . log p), Adam with
        weight decay 1e-4); returns Python floats (the reference's loss.data[0] indexing
        no longer works on current PyTorch).  With data_parallel every rank passes its share of the
        mini-batch and the gradients are averaged over the ranks before the update."""
        dev = self.device

        def tensor(a):
            if torch.is_tensor(a):
                return a.to(device=dev, dtype=torch.float32)
            return torch.as_tensor(np.asarray(a), dtype=torch.float32, device=dev)
        dp = self.data_parallel and torch.distributed.is_available() and torch.distributed.is_initialized()
        if not self.native_train:
            reduce_fn = None
            if dp:
                from .dist import allreduce_gradients
                reduce_fn = allreduce_gradients                  # every rank holds an equal share of the batch
            loss, entropy = train_step_tensors(self.policy_value_net, self.optimizer, tensor(state_batch), tensor(mcts_probs),
                                               tensor(winner_batch), lr, reduce_fn)
            self.version += 1
            return float(loss.item()), float(entropy.item())
        s, p, z = tensor(state_batch).contiguous(), tensor(mcts_probs).contiguous(), tensor(winner_batch).contiguous().reshape(-1)
        n = int(s.shape[0])
        eng = self._engine_handle()
        if self._adam is None:
            total = int(eng.L.azg_train_param_count(eng.h))
            self._adam = tuple(torch.zeros(total, dtype=torch.float32, device=dev) for _ in range(3))
        grad, m, v = self._adam
        arr, keep = self._param_ptrs()
        loss, entropy = C.c_float(0.0), C.c_float(0.0)
        torch.cuda.current_stream(dev).synchronize()           # the batch exists before the library's stream reads it
        self._adam_step += 1
        if dp:
            from .dist import allreduce_flat
            eng._chk(eng.L.azg_train_backward(eng.h, n, C.c_void_p(s.data_ptr()), C.c_void_p(p.data_ptr()), C.c_void_p(z.data_ptr()),
                                              arr, C.c_void_p(grad.data_ptr()), C.byref(loss), C.byref(entropy)))
            allreduce_flat(grad)                                 # one collective: the whole gradient is one vector
            torch.cuda.current_stream(dev).synchronize()
            eng._chk(eng.L.azg_train_adam(eng.h, arr, C.c_void_p(grad.data_ptr()), C.c_void_p(m.data_ptr()), C.c_void_p(v.data_ptr()),
                                          self._adam_step, float(lr), 0.9, 0.999, 1e-8, self.l2_const))
        else:
            eng._chk(eng.L.azg_train_step(eng.h, n, C.c_void_p(s.data_ptr()), C.c_void_p(p.data_ptr()), C.c_void_p(z.data_ptr()), arr,
                                          C.c_void_p(grad.data_ptr()), C.c_void_p(m.data_ptr()), C.c_void_p(v.data_ptr()),
                                          self._adam_step, float(lr), 0.9, 0.999, 1e-8, self.l2_const, C.byref(loss), C.byref(entropy)))
        self.version += 1
        return float(loss.value), float(entropy.value)

    def train_epoch(self, state_batch, mcts_probs, winner_batch, lr, kl_states, old_probs):
        """One epoch of policy_update's loop (train.py:176-201) in one library call and one wait:
        train_step on the batch, then the KL test's forward pass of kl_states with the updated
        parameters.  Returns (loss, entropy, new_probs [B,S], new_values [B,1], kl).  CUDA tensors;
        single process (data-parallel training exchanges gradients between the two halves: train_step +
        policy_value_in_training)."""
        dev = self.device
        if not self.native_train or self.data_parallel:
            raise RuntimeError("train_epoch is the library's single-process path")

        def tensor(a):
            return a.to(device=dev, dtype=torch.float32).contiguous()
        s, p, z = tensor(state_batch), tensor(mcts_probs), tensor(winner_batch).reshape(-1)
        ks, old = tensor(kl_states), tensor(old_probs)
        n, n_all, cells = int(s.shape[0]), int(ks.shape[0]), self.board_width * self.board_height
        eng = self._engine_handle()
        if self._adam is None:
            total = int(eng.L.azg_train_param_count(eng.h))
            self._adam = tuple(torch.zeros(total, dtype=torch.float32, device=dev) for _ in range(3))
        grad, m, v = self._adam
        arr, keep = self._param_ptrs()
        probs = torch.empty((n_all, cells), dtype=torch.float32, device=dev)
        vals = torch.empty(n_all, dtype=torch.float32, device=dev)
        loss, entropy, kl = C.c_float(0.0), C.c_float(0.0), C.c_double(0.0)
        torch.cuda.current_stream(dev).synchronize()           # the batch exists before the library's stream reads it
        self._adam_step += 1
        eng._chk(eng.L.azg_train_epoch(eng.h, n, C.c_void_p(s.data_ptr()), C.c_void_p(p.data_ptr()), C.c_void_p(z.data_ptr()), arr,
                                       C.c_void_p(grad.data_ptr()), C.c_void_p(m.data_ptr()), C.c_void_p(v.data_ptr()),
                                       self._adam_step, float(lr), 0.9, 0.999, 1e-8, self.l2_const, C.byref(loss), C.byref(entropy),
                                       n_all, C.c_void_p(ks.data_ptr()), C.c_void_p(probs.data_ptr()), C.c_void_p(vals.data_ptr()),
                                       C.c_void_p(old.data_ptr()), C.byref(kl)))
        self.version += 1
        return float(loss.value), float(entropy.value), probs, vals.reshape(-1, 1), float(kl.value)

    def get_policy_param(self):
        return self.policy_value_net.state_dict()

    def save_model(self, model_file):
        torch.save(self.get_policy_param(), model_file)
