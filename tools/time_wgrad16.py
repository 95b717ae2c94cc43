#!/usr/bin/env python3
"""Times the weight-gradient path of one k3 layer (nm_op_conv3d_backward without data gradient): NM355_W16_DBG ablations.
usage: time_wgrad16.py Cin Cout size N"""
import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from neural_marionette_amd import _lib
cfg = _lib.NmConfig(device=0, grid_size=64, nkeypoints=24, nlatent=128, nhidden=512, nneighbor=2, gaussian_sigma=1.5, sep_sigma=0.02, vol_fit_chamfer=1, use_graph_traj=1)
ctx = _lib.Context(cfg); ctx.bind_stream()
_lib.check(ctx.lib.nm_set_conv_mode(ctx.handle, 1), "mode")
Cin, Cout, size, N = [int(v) for v in sys.argv[1:5]]
x = torch.randn(N, size, size, size, Cin, device="cuda"); w = torch.randn(Cout, Cin, 3, 3, 3, device="cuda") * 0.05
dy = torch.randn(N, size, size, size, Cout, device="cuda"); d_w = torch.zeros_like(w); d_b = torch.zeros(Cout, device="cuda")
def call():
    _lib.check(ctx.lib.nm_op_conv3d_backward(ctx.handle, _lib.ptr(x), N, size, size, size, Cin, None, None, 1.0, _lib.ptr(w), Cout, 3, 1, 1, 0,
                                             _lib.ptr(dy), None, 0, _lib.ptr(d_w), _lib.ptr(d_b)), "bwd")
for _ in range(3): call()
torch.cuda.synchronize(); ts = []
for _ in range(7):
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); call(); e.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(e))
ts.sort()
fl = 2.0 * N * size ** 3 * Cin * Cout * 27
print("W16_DBG=%s  %dx%d @%d^3 x%d: median %.3f ms  (%.0f TFLOP/s algorithmic, x3 issued = %.2f of f16 peak)" % (
    os.environ.get("NM355_W16_DBG", "0"), Cin, Cout, size, N, ts[3], fl / ts[3] / 1e9, 3 * fl / ts[3] / 1e9 / 2500))
