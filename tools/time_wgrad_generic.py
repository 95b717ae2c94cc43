#!/usr/bin/env python3
"""Times nm_op_conv3d_backward (weight gradient only) for a k / stride given on the command line, alone on the GPU.
usage: time_wgrad_generic.py Cin Cout ks stride size N"""
import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from neural_marionette_amd import _lib
cfg = _lib.NmConfig(device=0, grid_size=64, nkeypoints=24, nlatent=128, nhidden=512, nneighbor=2, gaussian_sigma=1.5, sep_sigma=0.02, vol_fit_chamfer=1, use_graph_traj=1)
ctx = _lib.Context(cfg); ctx.bind_stream()
_lib.check(ctx.lib.nm_set_conv_mode(ctx.handle, 1), "mode")
Cin, Cout, ks, stride, size, N = [int(v) for v in sys.argv[1:7]]
pad = 1 if ks == 3 else 0
osz = (size + 2 * pad - ks) // stride + 1
x = torch.randn(N, size, size, size, Cin, device="cuda"); w = torch.randn(Cout, Cin, ks, ks, ks, device="cuda") * 0.05
dy = torch.randn(N, osz, osz, osz, Cout, device="cuda"); d_w = torch.zeros_like(w); d_b = torch.zeros(Cout, device="cuda")
def call():
    _lib.check(ctx.lib.nm_op_conv3d_backward(ctx.handle, _lib.ptr(x), N, size, size, size, Cin, None, None, 1.0, _lib.ptr(w), Cout, ks, stride, pad, 0,
                                             _lib.ptr(dy), None, 0, _lib.ptr(d_w), _lib.ptr(d_b)), "bwd")
for _ in range(3): call()
torch.cuda.synchronize(); ts = []
for _ in range(7):
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); call(); e.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(e))
ts.sort()
gb = (x.numel() + dy.numel()) * 4 / 1e9
print("wgrad %dx%d k%d s%d @%d^3 x%d: median %.3f ms  (inputs %.2f GB -> %.2f TB/s if read once)" % (Cin, Cout, ks, stride, size, N, ts[3], gb, gb / ts[3]))
