#!/usr/bin/env python3
"""A/B of one conv layer between two builds of the library: usage time_conv_lib.py <lib.so> Cin Cout size N [reps]; prints the op's
time; run under rocprofv3 --kernel-trace --stats for the kernel's own duration."""
import ctypes as C, sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neural_marionette_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
lib = _lib.load()
Cin, Cout, size, N = [int(v) for v in sys.argv[2:6]]
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 10
cfg = _lib.NmConfig(device=0, grid_size=64, nkeypoints=24, nlatent=128, nhidden=512, nneighbor=2, gaussian_sigma=1.5, sep_sigma=0.02, vol_fit_chamfer=1, use_graph_traj=1)
ctx = _lib.Context(cfg); ctx.bind_stream()
x = torch.randn(N, size, size, size, Cin, device="cuda"); w = torch.randn(Cout, Cin, 3, 3, 3, device="cuda") * 0.05; b = torch.zeros(Cout, device="cuda")
sc = torch.ones(N, Cin, device="cuda"); sh = torch.zeros(N, Cin, device="cuda"); out = torch.empty(N, size, size, size, Cout, device="cuda")
gam = torch.ones(Cout, device="cuda"); bet = torch.zeros(Cout, device="cuda"); gsc = torch.zeros(N, Cout, device="cuda"); gsh = torch.zeros(N, Cout, device="cuda")
def run():
    _lib.check(lib.nm_op_conv3d(ctx.handle, x.data_ptr(), N, size, size, size, Cin, sc.data_ptr(), sh.data_ptr(), 0.01, w.data_ptr(), b.data_ptr(), Cout, 3, 1, 1,
                                out.data_ptr(), Cout // 16, gam.data_ptr(), bet.data_ptr(), gsc.data_ptr(), gsh.data_ptr(), 0), "conv")
for _ in range(3): run()
torch.cuda.synchronize()
ts = []
for _ in range(reps):
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); run(); e.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(e))
ts.sort(); print("%s Cin=%d Cout=%d %d^3 N=%d: op median %.3f ms min %.3f" % (os.path.basename(sys.argv[1]), Cin, Cout, size, N, ts[len(ts)//2], ts[0]))
