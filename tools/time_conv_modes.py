"""Stand-alone duration of one k3 conv layer (nm_op_conv3d, 64 frames) in conv modes split16 / f16 / bf16 storage: run under rocprofv3
--kernel-trace and summarise with tools/summarize_trace.py.  usage: time_conv_modes.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from neural_marionette_amd import _lib
cfg = _lib.NmConfig(device=0, grid_size=64, nkeypoints=24, nlatent=128, nhidden=512, nneighbor=2, gaussian_sigma=1.5, sep_sigma=0.02, vol_fit_chamfer=1, use_graph_traj=1)
c = _lib.Context(cfg); c.bind_stream()
lib = c.lib
N = 64
for (Cin, Cout, size) in ((64, 64, 32), (32, 32, 64)):
    for (mode, h, name) in ((1, 0, "split16"), (3, 0, "f16"), (4, 1, "bf16")):
        _lib.check(lib.nm_set_conv_mode(c.handle, mode), "mode")
        _lib.check(lib.nm_op_set_storage16(c.handle, h, h), "set16")
        dt = torch.bfloat16 if h else torch.float32
        x = torch.randn(N, size, size, size, Cin, device="cuda").to(dt); w = torch.randn(Cout, Cin, 3, 3, 3, device="cuda") * 0.05; b = torch.zeros(Cout, device="cuda")
        sc = torch.ones(N, Cin, device="cuda"); sh = torch.zeros(N, Cin, device="cuda"); out = torch.empty(N, size, size, size, Cout, device="cuda", dtype=dt)
        gam = torch.ones(Cout, device="cuda"); bet = torch.zeros(Cout, device="cuda"); gsc = torch.zeros(N, Cout, device="cuda"); gsh = torch.zeros(N, Cout, device="cuda")
        def run():
            _lib.check(lib.nm_op_conv3d(c.handle, x.data_ptr(), N, size, size, size, Cin, sc.data_ptr(), sh.data_ptr(), 0.01, w.data_ptr(), b.data_ptr(), Cout, 3, 1, 1,
                                        out.data_ptr(), Cout // 16, gam.data_ptr(), bet.data_ptr(), gsc.data_ptr(), gsh.data_ptr(), 0), "conv")
        for _ in range(3): run()
        torch.cuda.synchronize()
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10): run()
        e.record(); torch.cuda.synchronize()
        fl = 2.0 * N * size ** 3 * 27 * Cin * Cout
        ms = a.elapsed_time(e) / 10
        print("%-8s %d->%d @%d^3 x%d: op %.3f ms = %.0f TFLOP/s algorithmic" % (name, Cin, Cout, size, N, ms, fl / ms / 1e9))
        del x, out
_lib.check(lib.nm_set_conv_mode(c.handle, 1), "mode"); _lib.check(lib.nm_op_set_storage16(c.handle, 0, 0), "set16")
