"""A/B of the one-product conv kernels (conv modes 3 / 4): conv_f16q2 (NM355_F16Q2=1, default) against conv_f16p2<SINGLE> (=0) through
nm_op_conv3d - outputs and GroupNorm scale / shift bit for bit, then durations (events, 10 launches).  usage: ab_f16q2.py [frames]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from neural_marionette_amd import _lib
N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
cfg = dict(device=0, grid_size=64, nkeypoints=24, nlatent=128, nhidden=512, nneighbor=2, gaussian_sigma=1.5, sep_sigma=0.02, vol_fit_chamfer=1, use_graph_traj=1)
ctx = {}
for sw in ("0", "1"):
    os.environ["NM355_F16Q2"] = sw
    c = _lib.Context(_lib.NmConfig(**cfg)); c.bind_stream()
    ctx[sw] = c
del os.environ["NM355_F16Q2"]
torch.manual_seed(0)
for (Cin, Cout, size, n) in ((64, 64, 32, N), (32, 64, 32, N), (64, 128, 16, N), (128, 128, 16, N), (128, 256, 16, 4), (64, 64, 40, 3)):
    for (mode, h, name) in ((3, 0, "f16"), (4, 1, "bf16")):
        dt = torch.bfloat16 if h else torch.float32
        x = torch.randn(n, size, size, size, Cin, device="cuda").to(dt); w = torch.randn(Cout, Cin, 3, 3, 3, device="cuda") * 0.05; b = torch.randn(Cout, device="cuda")
        sc = 1 + 0.1 * torch.randn(n, Cin, device="cuda"); sh = 0.1 * torch.randn(n, Cin, device="cuda")
        gam = torch.ones(Cout, device="cuda"); bet = torch.zeros(Cout, device="cuda")
        res = {}
        for sw, c in ctx.items():
            lib = c.lib
            _lib.check(lib.nm_set_conv_mode(c.handle, mode), "mode"); _lib.check(lib.nm_op_set_storage16(c.handle, h, h), "set16")
            out = torch.zeros(n, size, size, size, Cout, device="cuda", dtype=dt); gsc = torch.zeros(n, Cout, device="cuda"); gsh = torch.zeros(n, Cout, device="cuda")
            def run():
                _lib.check(lib.nm_op_conv3d(c.handle, x.data_ptr(), n, size, size, size, Cin, sc.data_ptr(), sh.data_ptr(), 0.01, w.data_ptr(), b.data_ptr(), Cout, 3, 1, 1,
                                            out.data_ptr(), Cout // 16, gam.data_ptr(), bet.data_ptr(), gsc.data_ptr(), gsh.data_ptr(), 0), "conv")
            for _ in range(3): run()
            torch.cuda.synchronize()
            a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(10): run()
            e.record(); torch.cuda.synchronize()
            res[sw] = (out.clone(), gsc.clone(), gsh.clone(), a.elapsed_time(e) / 10)
            _lib.check(lib.nm_set_conv_mode(c.handle, 1), "mode"); _lib.check(lib.nm_op_set_storage16(c.handle, 0, 0), "set16")
        same = all(torch.equal(res["0"][i], res["1"][i]) for i in range(3))
        fl = 2.0 * n * size ** 3 * 27 * Cin * Cout
        d = (res["0"][0].float() - res["1"][0].float()).abs().max().item()
        print("%-5s %3d->%3d @%d^3 x%d: f16p2<SINGLE> %.3f ms (%.0f TFLOP/s)  f16q2 %.3f ms (%.0f TFLOP/s)  identical %s (max diff %.3e, finite %s)" % (
            name, Cin, Cout, size, n, res["0"][3], fl / res["0"][3] / 1e9, res["1"][3], fl / res["1"][3] / 1e9, same, d, bool(torch.isfinite(res["1"][0].float()).all())))
