"""Stand-alone duration of the GroupNorm-backward passes (nm_op_gn_backward) at training sizes, fp32 and bfloat16 storage: run under
rocprofv3 --kernel-trace --stats.  usage: time_gnb.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from neural_marionette_amd import _lib
cfg = _lib.NmConfig(device=0, grid_size=64, nkeypoints=24, nlatent=128, nhidden=512, nneighbor=2, gaussian_sigma=1.5, sep_sigma=0.02, vol_fit_chamfer=1, use_graph_traj=1)
c = _lib.Context(cfg); c.bind_stream()
_lib.check(c.lib.nm_set_conv_mode(c.handle, 4), "mode")
for (N, size, C, groups) in ((64, 32, 64, 4), (64, 64, 32, 2)):
    V = size ** 3
    for h in (0, 1):
        dt = torch.bfloat16 if h else torch.float32
        y = (torch.randn(N, V, C, device="cuda") * 1.5 + 0.3).to(dt); dA = torch.randn(N, V, C, device="cuda").to(dt)
        dy = torch.empty(N, V, C, device="cuda", dtype=dt)
        gam = torch.rand(C, device="cuda") + 0.5; bet = torch.randn(C, device="cuda") * 0.2
        dg = torch.zeros(C, device="cuda"); db = torch.zeros(C, device="cuda"); dbias = torch.zeros(C, device="cuda")
        _lib.check(c.lib.nm_op_set_storage16(c.handle, h, h), "set16")
        for _ in range(4):
            _lib.check(c.lib.nm_op_gn_backward(c.handle, _lib.ptr(y), N, V, C, groups, _lib.ptr(gam), _lib.ptr(bet), 0.01, _lib.ptr(dA), _lib.ptr(dy), _lib.ptr(dg), _lib.ptr(db), _lib.ptr(dbias)), "gnb")
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        ev[0].record()
        for _ in range(10):
            _lib.check(c.lib.nm_op_gn_backward(c.handle, _lib.ptr(y), N, V, C, groups, _lib.ptr(gam), _lib.ptr(bet), 0.01, _lib.ptr(dA), _lib.ptr(dy), _lib.ptr(dg), _lib.ptr(db), _lib.ptr(dbias)), "gnb")
        ev[1].record(); torch.cuda.synchronize()
        elems = N * V * C
        print("N=%d %d^3 C=%d %s: whole gn_backward op %.1f us (%.2f GB of tensors: 2 forward-stat + 2 + 3 passes -> %.2f TB/s)" % (N, size, C, "bf16" if h else "fp32", ev[0].elapsed_time(ev[1]) * 100, elems * (2 if h else 4) * 6 / 1e9, elems * (2 if h else 4) * 6 / (ev[0].elapsed_time(ev[1]) * 1e-4) / 1e12))
        del y, dA, dy
_lib.check(c.lib.nm_op_set_storage16(c.handle, 0, 0), "set16")
