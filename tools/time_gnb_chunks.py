#!/usr/bin/env python3
"""Does the GroupNorm backward (two passes over (dA, y) + one write) run faster frame-chunk by frame-chunk, so that the second pass
finds its chunk in the 256 MB Infinity Cache?  nm_op_gn_backward on a 64^3 x 32 x 64-frame tensor: one call vs calls on chunks."""
import sys, os, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from neural_marionette_amd import _lib
cfg = _lib.NmConfig(device=0, grid_size=64, nkeypoints=24, nlatent=128, nhidden=512, nneighbor=2, gaussian_sigma=1.5, sep_sigma=0.02, vol_fit_chamfer=1, use_graph_traj=1)
ctx = _lib.Context(cfg); ctx.bind_stream()
C, size, N = 32, 64, 64
V = size ** 3
y = torch.randn(N, size, size, size, C, device="cuda"); dA = torch.randn_like(y); dy = torch.empty_like(y)
gam = torch.ones(C, device="cuda"); bet = torch.zeros(C, device="cuda")
dg = torch.zeros(C, device="cuda"); db = torch.zeros(C, device="cuda"); dbias = torch.zeros(C, device="cuda")
def call(n0, n):
    _lib.check(ctx.lib.nm_op_gn_backward(ctx.handle, _lib.ptr(y[n0:n0 + n]), n, V, C, C // 16, _lib.ptr(gam), _lib.ptr(bet), 0.01, _lib.ptr(dA[n0:n0 + n]), _lib.ptr(dy[n0:n0 + n]),
                                         _lib.ptr(dg), _lib.ptr(db), _lib.ptr(dbias)), "gnb")
for chunk in (64, 8, 4, 2, 1):
    def run():
        for n0 in range(0, N, chunk): call(n0, chunk)
    for _ in range(2): run()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): run()
    torch.cuda.synchronize()
    print("chunk of %2d frames: %.3f ms per 64 frames" % (chunk, (time.perf_counter() - t0) / 5 * 1e3))
