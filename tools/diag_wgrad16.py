"""Diagnostic: time the split-fp16 weight-gradient kernel on one decoder-sized layer (use under rocprofv3 --kernel-trace --stats)."""
import sys, os
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from neural_marionette_amd import _lib

cfg = _lib.NmConfig(device=0, grid_size=64, nkeypoints=24, nlatent=128, nhidden=512, nneighbor=2, gaussian_sigma=1.5, sep_sigma=0.02,
                    vol_fit_chamfer=1, use_graph_traj=1)
ctx = _lib.Context(cfg); ctx.bind_stream()
_lib.check(ctx.lib.nm_set_conv_mode(ctx.handle, 1), "mode")
Cin, Cout, size, N = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
x = torch.randn(N, size, size, size, Cin, device="cuda")
w = torch.randn(Cout, Cin, 3, 3, 3, device="cuda") * 0.05
dy = torch.randn(N, size, size, size, Cout, device="cuda")
d_w = torch.zeros_like(w); d_b = torch.zeros(Cout, device="cuda")
for _ in range(3):
    _lib.check(ctx.lib.nm_op_conv3d_backward(ctx.handle, _lib.ptr(x), N, size, size, size, Cin, None, None, 1.0, _lib.ptr(w), Cout, 3, 1, 1, 0,
                                             _lib.ptr(dy), None, 0, _lib.ptr(d_w), _lib.ptr(d_b)), "bwd")
torch.cuda.synchronize()
