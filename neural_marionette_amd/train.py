"""Training steps of the reference's two regimes.

Detector mode (`pretrained_mode = 0`, train.py:270-276): `DetectorTrainer` — the keypoint detector trains on its 11 losses
with the AIST weights (train.py:177-181); forward = nm_detector_forward_train, backward = nm_detector_backward (HIP kernels),
one flat-bucket gradient all-reduce (8.56 M floats = 34 MB), fused Adam.

Learner mode:

The reference's `pretrained_mode = 1` regime (train.py:146, 270-276; the mode stored in
pretrained/aist/opt.pickle): the keypoint detector is frozen and only the dynamics module
trains, on `keypoints.detach()` (neural_marionette.py:53).  One step of train.py:376-412 is

    log  = network(voxel, {'detector': False, 'learner': True})
    loss = sum_k weight[k] * log[k]          (train.py:389-398; only kypt_recon_loss and kl_kypt carry gradients)
    loss.backward();  optimizer.step()       (Adam, lr 4e-4, defaults; re-created every epoch, train.py:366-374)

Here the forward and the back-propagation through time run in libnm355.so, the gradient
all-reduce (clip-sharded data parallelism) goes through `torch.distributed` — backend "nccl" is
RCCL over xGMI on the GPU box — as ONE flat bucket of 1.53 M floats (6.1 MB), and Adam is the
library's fused kernel.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch
import torch.distributed as dist

from . import _lib

# AIST loss weights of the reference (train.py:177-181 / opt.pickle)
LEARNER_LOSS_WEIGHTS = {"kypt_recon_loss": 1.0, "kl_kypt": 0.003}


def allreduce_mean_(tensors, world: Optional[int] = None) -> None:
    """Average a list of gradient tensors across ranks through one flat bucket (in place)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return
    world = world or dist.get_world_size()
    flat = torch.cat([t.reshape(-1) for t in tensors])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    flat /= world
    off = 0
    for t in tensors:
        n = t.numel()
        t.copy_(flat[off:off + n].view_as(t))
        off += n


def adam_step_(eng, params, grads, exp_avg, exp_avg_sq, step, lr, betas, eps) -> None:
    """torch.optim.Adam's update for a list of tensors in one library launch (nm_adam_step_multi); bumps the version counters so
    that the engine re-uploads / re-packs the weights before the next forward."""
    import ctypes as C
    n = len(params)
    arr = lambda ts: (C.c_void_p * n)(*[_lib.ptr(t) for t in ts])
    gs = [g.contiguous() for g in grads]
    with torch.no_grad():
        eng.call("nm_adam_step_multi", arr([p.data for p in params]), arr(gs), arr(exp_avg), arr(exp_avg_sq),
                 (C.c_int64 * n)(*[p.numel() for p in params]), n, step, lr, betas[0], betas[1], eps)
        torch._foreach_add_(list(params), 0.0)


class LearnerTrainer:
    def __init__(self, net, lr: float = 4e-4, weights: Optional[Dict[str, float]] = None,
                 betas=(0.9, 0.999), eps: float = 1e-8):
        self.net = net
        self.lr, self.betas, self.eps = lr, betas, eps
        self.weights = dict(LEARNER_LOSS_WEIGHTS if weights is None else weights)
        net.control_active({"detector": False, "learner": True})
        self.params = [p for p in net.dyna_module.parameters() if p.requires_grad]
        self.reset_optimizer()

    def reset_optimizer(self) -> None:
        """The reference re-instantiates Adam at every epoch (train.py:366-374): state starts from zero."""
        self.t = 0
        self.m = [torch.zeros_like(p) for p in self.params]
        self.v = [torch.zeros_like(p) for p in self.params]

    def step(self, vox, eps=None, sync: bool = True):
        """One training step.  sync=True returns python floats (waits for the device); sync=False returns 0-dim device tensors."""
        net = self.net
        for p in self.params:
            p.grad = None
        log = net(vox, {"detector": False, "learner": True}, eps=eps)
        loss = sum(w * log[k] for k, w in self.weights.items())
        loss.backward()
        grads = [p.grad for p in self.params]
        allreduce_mean_(grads)
        eng = net._engine
        eng.ready()
        self.t += 1
        adam_step_(eng, self.params, grads, self.m, self.v, self.t, self.lr, self.betas, self.eps)
        if not sync:
            return {"loss": loss.detach(), **{k: log[k].detach() for k in self.weights}}
        return {"loss": float(loss.detach()), **{k: float(log[k].detach()) for k in self.weights}}


# AIST loss weights of the reference for the detector losses (train.py:177-181 / opt.pickle)
DETECTOR_LOSS_WEIGHTS = {"recon_loss": 100.0, "sparsity_loss": 5.0, "separation_loss": 0.1, "vol_fit_reg": 10.0,
                         "kypt_const_loss": 0.0, "local_const_loss": 1e-3, "time_const_loss": 1.0, "sparsity_const_loss": 0.01,
                         "intensity_const_loss": 0.01, "graph_traj_loss": 1.0, "graph_vol_loss": 0.0}


class DetectorTrainer:
    """One step of train.py:376-412 in detector mode: log = network(voxel, {'detector': True, 'learner': False});
    loss = sum_k weight[k] * log[k]; loss.backward(); Adam(lr 4e-4).  `nepoch` drives KyptDetector.anneal (affinity start)."""

    def __init__(self, net, lr: float = 4e-4, weights: Optional[Dict[str, float]] = None, betas=(0.9, 0.999), eps: float = 1e-8):
        self.net = net
        self.lr, self.betas, self.eps = lr, betas, eps
        self.weights = dict(DETECTOR_LOSS_WEIGHTS if weights is None else weights)
        self.acts = {"detector": True, "learner": False}
        net.control_active(self.acts)
        self.reset_optimizer()

    def _params(self):
        return [p for p in self.net.kypt_detector.parameters() if p.requires_grad]

    def reset_optimizer(self) -> None:
        """The reference re-instantiates Adam at every epoch (train.py:366-374): state starts from zero."""
        self.t = 0
        self.state = {}

    def step(self, vox, sync: bool = True):
        """One training step.  sync=True returns python floats (waits for the device); sync=False returns 0-dim device tensors and
        lets the host run ahead into the next step."""
        net = self.net
        params = self._params()
        for p in net.kypt_detector.parameters():
            p.grad = None
        log = net(vox, self.acts)
        loss = sum(w * log[k] for k, w in self.weights.items())
        loss.backward()
        params = [p for p in params if p.grad is not None]
        grads = [p.grad for p in params]
        allreduce_mean_(grads)
        eng = net._engine
        eng.ready()
        self.t += 1
        for p in params:
            if id(p) not in self.state:
                self.state[id(p)] = (torch.zeros_like(p), torch.zeros_like(p))
        adam_step_(eng, params, grads, [self.state[id(p)][0] for p in params], [self.state[id(p)][1] for p in params], self.t,
                   self.lr, self.betas, self.eps)
        if not sync:
            return {"loss": loss.detach(), **{k: log[k].detach() for k in self.weights}}
        return {"loss": float(loss.detach()), **{k: float(log[k].detach()) for k in self.weights}}
