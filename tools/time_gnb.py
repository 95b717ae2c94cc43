#!/usr/bin/env python3
"""GroupNorm(+LeakyReLU) backward of one layer alone on the GPU (nm_op_gn_backward: forward partials + finalize, backward partials,
finalize, apply): kernel times come from rocprofv3 --stats around this script.  usage: time_gnb.py C size N"""
import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from neural_marionette_amd import _lib
cfg = _lib.NmConfig(device=0, grid_size=64, nkeypoints=24, nlatent=128, nhidden=512, nneighbor=2, gaussian_sigma=1.5, sep_sigma=0.02, vol_fit_chamfer=1, use_graph_traj=1)
ctx = _lib.Context(cfg); ctx.bind_stream()
C, size, N = [int(v) for v in sys.argv[1:4]]
V = size ** 3
y = torch.randn(N, size, size, size, C, device="cuda"); dA = torch.randn_like(y); dy = torch.empty_like(y)
gam = torch.ones(C, device="cuda"); bet = torch.zeros(C, device="cuda")
dg = torch.zeros(C, device="cuda"); db = torch.zeros(C, device="cuda"); dbias = torch.zeros(C, device="cuda")
def call():
    _lib.check(ctx.lib.nm_op_gn_backward(ctx.handle, _lib.ptr(y), N, V, C, C // 16, _lib.ptr(gam), _lib.ptr(bet), 0.01, _lib.ptr(dA), _lib.ptr(dy),
                                         _lib.ptr(dg), _lib.ptr(db), _lib.ptr(dbias)), "gnb")
for _ in range(3): call()
torch.cuda.synchronize()
for _ in range(10): call()
torch.cuda.synchronize()
print("tensor %.2f GB" % (y.numel() * 4 / 1e9))
