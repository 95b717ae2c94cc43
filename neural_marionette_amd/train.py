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
    """Average a list of gradient tensors across ranks through one flat bucket (in place).  Generic helper (copies in and out);
    the trainers below keep their gradients IN a persistent bucket (GradBucket) and skip both copies."""
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


class GradBucket:
    """One pre-allocated flat gradient buffer; every parameter's gradient is a view into it, so the backward kernels write
    straight into the buffer the collective reduces (no torch.cat, no copy back).  ``chunks`` are contiguous element ranges
    in the order the backward pass completes them: a chunk's all-reduce (RCCL over xGMI: ``torch.distributed`` backend
    'nccl'; gloo in the CPU tests) is issued on a side stream as soon as its gradients exist and overlaps the rest of the
    backward pass.  One process per GPU, sum then 1/world."""

    always_reduce = False      # True: issue the collectives in a 1-rank group too (the GPU test of the RCCL path on one device)

    def __init__(self, named_params, chunk_of, nchunks: int):
        """named_params: [(name, parameter)]; chunk_of(name) -> chunk index in completion order."""
        order = sorted(range(len(named_params)), key=lambda i: (chunk_of(named_params[i][0]), i))
        dev, total = named_params[0][1].device, sum(p.numel() for _, p in named_params)
        self.flat = torch.zeros(total, device=dev, dtype=torch.float32)
        self.views, self.chunks = {}, []
        off, cur, start = 0, 0, 0
        for i in order:
            name, p = named_params[i]
            ch = chunk_of(name)
            while cur < ch:
                self.chunks.append((start, off)); start = off; cur += 1
            self.views[name] = self.flat[off:off + p.numel()].view(p.shape)
            off += p.numel()
        while cur < nchunks:
            self.chunks.append((start, off)); start = off; cur += 1
        self.comm_stream = torch.cuda.Stream(dev) if self.flat.is_cuda else None
        self._pending = []
        # bench.py: event pair around a step's collectives on the side stream (first chunk's start .. last chunk's end)
        self.time_collectives = False
        self._ev_t0 = self._ev_t1 = None
        self._t0_set = False

    def chunk(self, i) -> torch.Tensor:
        a, b = self.chunks[i]
        return self.flat[a:b]

    def reduce_chunk(self, i, after_event=None) -> None:
        """Start the all-reduce of chunk i (asynchronous; on the GPU: on the bucket's side stream, behind ``after_event`` or
        behind everything queued on the current stream so far)."""
        if not dist.is_initialized() or (dist.get_world_size() == 1 and not self.always_reduce) or self.chunks[i][0] == self.chunks[i][1]:
            return
        if self.comm_stream is None:
            self._pending.append(dist.all_reduce(self.chunk(i), op=dist.ReduceOp.SUM, async_op=True))
            return
        if after_event is not None:
            self.comm_stream.wait_event(after_event)
        else:
            self.comm_stream.wait_stream(torch.cuda.current_stream(self.flat.device))
        with torch.cuda.stream(self.comm_stream):
            if self.time_collectives and not self._t0_set:
                if self._ev_t0 is None:
                    self._ev_t0, self._ev_t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                self._ev_t0.record(self.comm_stream); self._t0_set = True
            self._pending.append(dist.all_reduce(self.chunk(i), op=dist.ReduceOp.SUM, async_op=True))
            if self.time_collectives:
                self._ev_t1.record(self.comm_stream)

    def finish(self) -> None:
        """Wait for the started collectives (stream-ordered on the GPU) and turn the sums into means."""
        if not self._pending:
            return
        for w in self._pending:
            w.wait()
        self._pending = []
        self._t0_set = False
        if self.comm_stream is not None:
            torch.cuda.current_stream(self.flat.device).wait_stream(self.comm_stream)
        self.flat.mul_(1.0 / dist.get_world_size())

    def last_allreduce_ms(self):
        """Side-stream time from the start of the last step's first collective to the end of its last one (None when the step
        issued none: no process group, or one rank without ``always_reduce``).  Synchronises on the end event."""
        if not self.time_collectives or self._ev_t0 is None:
            return None
        self._ev_t1.synchronize()
        return float(self._ev_t0.elapsed_time(self._ev_t1))


def _bucket_ok(flat: torch.Tensor, check: bool, what: str) -> torch.Tensor:
    """The library's range guard is deferred: a conv that overflowed the fp16 range in step n (conv modes 'split16' / 'f16' / 'bf16')
    is reported at the first library call of step n + 1 (include/nm355.h "Range / finiteness status") - after this step's Adam
    update.  So that such a step cannot destroy the master weights and the Adam moments, the finiteness of the whole gradient bucket
    (after the collective: NaN spreads through the sum, every rank sees the same verdict) is reduced to ONE device float - one read
    pass over <= 34 MB, no synchronisation - and handed to the Adam launch (nm_adam_step_multi_ok), which does nothing at all when
    it is 0: parameters, both moments untouched.  The host-side step counter still advances (bias corrections one step ahead for the
    rest of the epoch - the price of not synchronising); step(check=True) reads the flag on the host instead, raises BEFORE the
    optimizer and leaves the counter alone."""
    ok = torch.isfinite(flat).all()
    if check and not bool(ok):
        raise _lib.NmError(f"{what}: the step's gradient bucket holds a non-finite entry - no optimizer update was applied "
                           "(an activation left the split-fp16 range, or the input was not finite; set_conv_mode('auto') re-runs such a "
                           "forward in exact fp32)")
    return ok.to(torch.float32)


def adam_step_(eng, params, grads, exp_avg, exp_avg_sq, step, lr, betas, eps, ok: Optional[torch.Tensor] = None) -> None:
    """torch.optim.Adam's update for a list of tensors in one library launch (nm_adam_step_multi_ok; ok: 0-dim device float, 0 = skip
    the whole update on the device); bumps the version counters so that the engine re-uploads / re-packs the weights before the
    next forward."""
    import ctypes as C
    n = len(params)
    arr = lambda ts: (C.c_void_p * n)(*[_lib.ptr(t) for t in ts])
    gs = [g.contiguous() for g in grads]
    with torch.no_grad():
        eng.call("nm_adam_step_multi_ok", arr([p.data for p in params]), arr(gs), arr(exp_avg), arr(exp_avg_sq),
                 (C.c_int64 * n)(*[p.numel() for p in params]), n, step, lr, betas[0], betas[1], eps, _lib.ptr(ok) if ok is not None else None)
        torch._foreach_add_(list(params), 0.0)


class LearnerTrainer:
    def __init__(self, net, lr: float = 4e-4, weights: Optional[Dict[str, float]] = None,
                 betas=(0.9, 0.999), eps: float = 1e-8, lean: bool = False):
        """lean=True: the frozen detector runs without its voxel decoder and losses (KyptDetector.detect) - the learner's loss reads only
        the keypoints and the affinity; losses, gradients and updated weights are bit-identical to lean=False, which executes the whole
        detector forward as the reference's step does (neural_marionette.py:45-47)."""
        self.net = net
        self.lean = bool(lean)
        if self.lean and weights is not None:
            from .spec import DETECTOR_LOSS_KEYS
            bad = sorted(set(weights) & set(DETECTOR_LOSS_KEYS))
            if bad:
                raise ValueError(f"LearnerTrainer(lean=True) does not compute the detector losses {bad}: use lean=False to weight them")
        self.lr, self.betas, self.eps = lr, betas, eps
        self.weights = dict(LEARNER_LOSS_WEIGHTS if weights is None else weights)
        net.control_active({"detector": False, "learner": True})
        self.named = [("dyna_module." + n, p) for n, p in net.dyna_module.named_parameters() if p.requires_grad]
        self.params = [p for _, p in self.named]
        self.bucket: Optional[GradBucket] = None
        self._bucket_key = None
        self.reset_optimizer()

    def reset_optimizer(self) -> None:
        """The reference re-instantiates Adam at every epoch (train.py:366-374): state starts from zero."""
        self.t = 0
        self.m = [torch.zeros_like(p) for p in self.params]
        self.v = [torch.zeros_like(p) for p in self.params]

    def step(self, vox, eps=None, sync: bool = True, global_clips: Optional[int] = None, check: bool = False):
        """One training step.  sync=True returns python floats (waits for the device); sync=False returns 0-dim device tensors.
        check=True: the finiteness of the gradient bucket is read on the host (one synchronisation) and a non-finite step raises NmError
        BEFORE the optimizer runs; default: the Adam launch skips itself on the device (_bucket_ok).
        global_clips: as DetectorTrainer.step (uneven clip split: the loss that is back-propagated is scaled by this rank's share).
        Gradients land in a persistent GradBucket (every p.grad is a view of the flat buffer autograd accumulates into), which is
        what the collective reduces: one all-reduce of 1.53 M floats, no torch.cat, no copy back - the DetectorTrainer's scheme."""
        net = self.net
        key = tuple((n, p.data_ptr()) for n, p in self.named)
        if self.bucket is None or key != self._bucket_key:
            self.bucket = GradBucket(self.named, lambda n: 0, 1)
            self._bucket_key = key
        bucket = self.bucket
        bucket.flat.zero_()
        for n, p in self.named:
            p.grad = bucket.views[n]                  # autograd accumulates in place into an existing .grad
        # (no synchronous range probe per step - the weights change every step - unless conv mode 'auto' was asked for explicitly: the
        #  deferred guard reports an overflow at the next call and the Adam launch skips itself on the device when the bucket is not finite: _bucket_ok)
        net._engine.suppress_probe = True
        try:
            log = net(vox, {"detector": False, "learner": True}, eps=eps, detector_outputs="keypoints" if self.lean else "all")
        finally:
            net._engine.suppress_probe = False
        loss = sum(w * log[k] for k, w in self.weights.items())
        world = dist.get_world_size() if dist.is_initialized() else 1
        scale = 1.0 if global_clips is None else float(vox.shape[0]) * world / float(global_clips)
        (loss * scale if scale != 1.0 else loss).backward()
        for n, p in self.named:                       # (autograd replaces .grad instead of accumulating in some modes: keep the bucket authoritative)
            if p.grad is not None and p.grad.data_ptr() != bucket.views[n].data_ptr():
                bucket.views[n].copy_(p.grad); p.grad = bucket.views[n]
        bucket.reduce_chunk(0)
        bucket.finish()
        ok = _bucket_ok(bucket.flat, check, "LearnerTrainer.step")
        grads = [bucket.views[n] for n, _ in self.named]
        eng = net._engine
        eng.ready()
        self.t += 1
        adam_step_(eng, self.params, grads, self.m, self.v, self.t, self.lr, self.betas, self.eps, ok=ok)
        if not sync:
            return {"loss": loss.detach(), **{k: log[k].detach() for k in self.weights}}
        return {"loss": float(loss.detach()), **{k: float(log[k].detach()) for k in self.weights}}


# AIST loss weights of the reference for the detector losses (train.py:177-181 / opt.pickle)
DETECTOR_LOSS_WEIGHTS = {"recon_loss": 100.0, "sparsity_loss": 5.0, "separation_loss": 0.1, "vol_fit_reg": 10.0,
                         "kypt_const_loss": 0.0, "local_const_loss": 1e-3, "time_const_loss": 1.0, "sparsity_const_loss": 0.01,
                         "intensity_const_loss": 0.01, "graph_traj_loss": 1.0, "graph_vol_loss": 0.0}


class DetectorTrainer:
    """One step of train.py:376-412 in detector mode: log = network(voxel, {'detector': True, 'learner': False});
    loss = sum_k weight[k] * log[k]; loss.backward(); Adam(lr 4e-4).  `nepoch` drives KyptDetector.anneal (affinity start).

    The step calls the library's training forward / backward directly (same kernels as the autograd bridge of
    KyptDetector.forward) with the gradients landing in a GradBucket: chunk 0 = the decoder's parameters
    (kypt_detector.kypt_to_vox.*, complete when the decoder's backward is - the library records an event there), chunk 1 = the
    rest; chunk 0's all-reduce runs beside the backward of the heads and both feature nets."""

    def __init__(self, net, lr: float = 4e-4, weights: Optional[Dict[str, float]] = None, betas=(0.9, 0.999), eps: float = 1e-8):
        from .spec import DETECTOR_LOSS_KEYS
        self.net = net
        self.lr, self.betas, self.eps = lr, betas, eps
        self.weights = dict(DETECTOR_LOSS_WEIGHTS if weights is None else weights)
        self.loss_keys = DETECTOR_LOSS_KEYS
        unknown = sorted(set(self.weights) - set(DETECTOR_LOSS_KEYS))
        if unknown:
            raise KeyError(f"DetectorTrainer: not detector losses: {unknown} (known: {list(DETECTOR_LOSS_KEYS)})")
        self.acts = {"detector": True, "learner": False}
        net.control_active(self.acts)
        self.bucket: Optional[GradBucket] = None
        self._bucket_key = None
        self._wvec = None
        self._wvec_key = None
        self._ev = None
        self._rank_scale = 1.0
        self.reset_optimizer()

    def _named(self):
        """Every kypt_detector parameter: nm_detector_backward writes a gradient for each of them (frozen ones included - e.g.
        affinity_params while KyptDetector.anneal keeps it frozen, kypt_detector.py:71-78); `requires_grad` only decides which
        ones Adam updates and which get a `.grad`."""
        return [("kypt_detector." + n, p) for n, p in self.net.kypt_detector.named_parameters()]

    def _weight_vector(self, dev):
        """dL/dloss_k as a device vector, rebuilt whenever `self.weights` is edited (e.g. a per-epoch schedule)."""
        key = (str(dev), tuple(sorted(self.weights.items())))
        if key != self._wvec_key:
            unknown = sorted(set(self.weights) - set(self.loss_keys))
            if unknown:
                raise KeyError(f"DetectorTrainer: not detector losses: {unknown}")
            self._wvec = torch.tensor([float(self.weights.get(k, 0.0)) for k in self.loss_keys], device=dev)
            self._wvec_key = key
        return self._wvec

    def reset_optimizer(self) -> None:
        """The reference re-instantiates Adam at every epoch (train.py:366-374): state starts from zero."""
        self.t = 0
        self.state = {}

    # -- the three device operations (replaced by CPU stand-ins in tests/test_sharding_cpu.py) ---------------------------
    def _forward_backward(self, vox, named, bucket: GradBucket):
        """Training forward + backward of L = sum_k w_k loss_k; gradients into bucket.views; returns the 11 losses (device).
        Starts chunk 0's all-reduce as soon as the decoder's gradients are complete."""
        import ctypes as C
        from .spec import FEAT_DIM
        det = self.net.kypt_detector
        eng = self.net._engine
        eng.set_training(True)
        c = eng.ready()
        dev = c.device
        G, K, g = det.grid_size, det.nkeypoints, det.grid_size // 4
        if vox.dim() != 6 or tuple(vox.shape[2:]) != (1, G, G, G):
            raise ValueError(f"expected seq of shape (B,T,1,{G},{G},{G}), got {tuple(vox.shape)}")
        vox = vox.detach().to(device=dev, dtype=torch.float32).contiguous()
        B, T = int(vox.shape[0]), int(vox.shape[1])
        kp = torch.empty(B, T, K, 4, device=dev); hm = torch.empty(B, T, K, g, g, g, device=dev)
        ff = torch.empty(B, FEAT_DIM, g, g, g, device=dev); recon = torch.empty(B, T, 1, G, G, G, device=dev)
        aff = torch.empty(det.nneighbor, K, K, 1, device=dev) if det.affinity_start else None
        losses = torch.empty(len(self.loss_keys), device=dev)
        eng.suppress_probe = True                   # (see LearnerTrainer.step)
        try:
            eng.call_conv("nm_detector_forward_train", _lib.ptr(vox), B, T, int(det.affinity_start), _lib.ptr(kp), _lib.ptr(hm), _lib.ptr(ff),
                          _lib.ptr(recon), _lib.ptr(aff), _lib.ptr(losses))
        finally:
            eng.suppress_probe = False
        wvec = self._weight_vector(dev)
        if self._rank_scale != 1.0:                 # uneven clip split: this rank's share of the global clip mean (see step())
            wvec = wvec * self._rank_scale
        if self._ev is None:
            self._ev = torch.cuda.Event()
            self._ev.record()                       # (creates the HIP event; the library re-records it)
        arr = (_lib.NmNamedTensor * len(named))()
        keep = []
        for i, (n, _) in enumerate(named):
            v = bucket.views[n]
            keep.append(n.encode())
            arr[i].name, arr[i].data, arr[i].numel = keep[-1], v.data_ptr(), v.numel()
        eng.call("nm_ctx_set_backward_event", C.c_void_p(self._ev.cuda_event))
        try:
            eng.call("nm_detector_backward", _lib.ptr(wvec), arr, len(named))
        finally:                                    # never leave the raw event handle in the context (a failed backward included)
            eng.call("nm_ctx_set_backward_event", None)
        bucket.reduce_chunk(0, after_event=self._ev)
        self._keep = (vox, kp, recon, aff)          # read by the backward kernels (stream-ordered: freed no earlier than the next step)
        return losses

    def _adam(self, params, grads, m, v, ok=None):
        eng = self.net._engine
        eng.ready()
        adam_step_(eng, params, grads, m, v, self.t, self.lr, self.betas, self.eps, ok=ok)

    def step(self, vox, sync: bool = True, global_clips: Optional[int] = None, check: bool = False):
        """One training step.  sync=True returns python floats (waits for the device); sync=False returns 0-dim device tensors and
        lets the host run ahead into the next step.  check=True: a non-finite gradient bucket raises NmError before the optimizer
        (one host synchronisation); default: the Adam launch skips itself on the device (_bucket_ok).
        global_clips: total number of clips of the step over all ranks when the ranks hold DIFFERENT numbers of clips (a batch that
        does not divide by the world size, dist.clip_shard).  Every loss is a mean over clips, so the single-process gradient of the
        whole batch is the clip-weighted mean of the ranks' gradients: this rank's dL/dloss vector is scaled by
        clips_here * world / global_clips (one multiply of 11 numbers - every gradient the backward kernels write is linear in it),
        and the bucket's sum / world is then that weighted mean.  None: equal shares (the bench's 4 clips per GPU)."""
        world = dist.get_world_size() if dist.is_initialized() else 1
        self._rank_scale = 1.0 if global_clips is None else float(vox.shape[0]) * world / float(global_clips)
        named = self._named()
        key = tuple((n, p.data_ptr()) for n, p in named)
        if self.bucket is None or key != self._bucket_key:
            self.bucket = GradBucket(named, lambda n: 0 if n.startswith("kypt_detector.kypt_to_vox.") else 1, 2)
            self._bucket_key = key
        bucket = self.bucket
        losses = self._forward_backward(vox, named, bucket)
        bucket.reduce_chunk(1)
        bucket.finish()
        ok = _bucket_ok(bucket.flat, check, "DetectorTrainer.step")
        live = [(n, p) for n, p in named if p.requires_grad]          # frozen parameters: gradient computed, not applied
        params = [p for _, p in live]
        grads = [bucket.views[n] for n, _ in live]
        for p, g in zip(params, grads):
            p.grad = g                                # (a view of the bucket: what a caller inspecting .grad expects to find)
        self.t += 1
        for p in params:
            if id(p) not in self.state:
                self.state[id(p)] = (torch.zeros_like(p), torch.zeros_like(p))
        self._adam(params, grads, [self.state[id(p)][0] for p in params], [self.state[id(p)][1] for p in params], ok=ok)
        loss = (losses * self._weight_vector(losses.device)).sum()      # (this rank's clips; dist.mean_losses gives the global figure)
        log = {k: losses[i] for i, k in enumerate(self.loss_keys) if k in self.weights}
        if not sync:
            return {"loss": loss, **log}
        return {"loss": float(loss), **{k: float(v) for k, v in log.items()}}
