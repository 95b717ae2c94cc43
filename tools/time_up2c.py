#!/usr/bin/env python3
"""Times the fused-upsample 64 -> 32 layer at the bench shape (64 frames, 32^3 -> 64^3) through nm_op_conv3d.
usage: time_up2c.py [reps]   (NM355_UP2C / NM355_UP2C_DIAG select the kernel / ablations)"""
import sys, os, torch, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neural_marionette_amd import _lib
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
N, D, Cin, Cout = 64, 32, 64, 32
cfg = _lib.NmConfig(device=0, grid_size=64, nkeypoints=24, nlatent=128, nhidden=512, nneighbor=2, gaussian_sigma=1.5, sep_sigma=0.02, vol_fit_chamfer=1, use_graph_traj=1)
ctx = _lib.Context(cfg); ctx.bind_stream()
_lib.check(ctx.lib.nm_set_conv_mode(ctx.handle, int(os.environ.get("NM355_TIME_MODE", "1"))), "mode")
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.randn(N, D, D, D, Cin, device="cuda", generator=g)
w = torch.randn(Cout, Cin, 3, 3, 3, device="cuda", generator=g) / (Cin * 27) ** 0.5
b = torch.randn(Cout, device="cuda", generator=g) * 0.1
sc = torch.rand(N, Cin, device="cuda", generator=g) + 0.5; sh = torch.randn(N, Cin, device="cuda", generator=g) * 0.3
gam = torch.ones(Cout, device="cuda"); bet = torch.zeros(Cout, device="cuda")
out = torch.empty(N, 2 * D, 2 * D, 2 * D, Cout, device="cuda"); gsc = torch.zeros(N, Cout, device="cuda"); gsh = torch.zeros(N, Cout, device="cuda")
def call():
    _lib.check(ctx.lib.nm_op_conv3d(ctx.handle, _lib.ptr(x), N, D, D, D, Cin, _lib.ptr(sc), _lib.ptr(sh), 0.01, _lib.ptr(w), _lib.ptr(b), Cout, 3, 1, 1,
                                    _lib.ptr(out), 2, _lib.ptr(gam), _lib.ptr(bet), _lib.ptr(gsc), _lib.ptr(gsh), 1), "op")
for _ in range(3): call()
torch.cuda.synchronize()
ts = []
for _ in range(reps):
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); call(); e.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(e))
ts.sort()
print("UP2C=%s DIAG=%s  median %.3f ms  min %.3f ms  (whole op: compose + pack + conv [+ shell] + gn_finalize)" % (
    os.environ.get("NM355_UP2C", "1"), os.environ.get("NM355_UP2C_DIAG", "0"), ts[len(ts) // 2], ts[0]))
