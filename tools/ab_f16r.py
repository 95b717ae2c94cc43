"""A/B of the one-product modes' 32-output-channel conv (conv modes 3 / 4): conv_f16r (NM355_F16R=1, default: weights resident in LDS)
against conv_f16p<SINGLE> (=0) through nm_op_conv3d.  The two sum the k dimension in different orders, so the outputs are compared
with each other and BOTH with an fp64 convolution of the same fp16-rounded operands; then durations (events, 10 launches).
usage: ab_f16r.py [frames]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import torch.nn.functional as F
from neural_marionette_amd import _lib
N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
cfg = dict(device=0, grid_size=64, nkeypoints=24, nlatent=128, nhidden=512, nneighbor=2, gaussian_sigma=1.5, sep_sigma=0.02, vol_fit_chamfer=1, use_graph_traj=1)
ctx = {}
for sw in ("0", "1"):
    os.environ["NM355_F16R"] = sw
    c = _lib.Context(_lib.NmConfig(**cfg)); c.bind_stream()
    ctx[sw] = c
del os.environ["NM355_F16R"]
torch.manual_seed(0)
for (Cin, Cout, size, n, affine) in ((32, 32, 64, N, True), (64, 32, 64, max(N // 2, 1), False), (64, 32, 32, N, True), (32, 32, 32, N, False), (32, 32, 40, 3, True), (64, 32, 16, 5, True)):
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
                _lib.check(lib.nm_op_conv3d(c.handle, x.data_ptr(), n, size, size, size, Cin, sc.data_ptr() if affine else None, sh.data_ptr() if affine else None,
                                            0.01 if affine else 1.0, w.data_ptr(), b.data_ptr(), Cout, 3, 1, 1,
                                            out.data_ptr(), Cout // 16, gam.data_ptr(), bet.data_ptr(), gsc.data_ptr(), gsh.data_ptr(), 0), "conv")
            for _ in range(3): run()
            torch.cuda.synchronize()
            a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(10): run()
            e.record(); torch.cuda.synchronize()
            res[sw] = (out.clone(), gsc.clone(), gsh.clone(), a.elapsed_time(e) / 10)
            _lib.check(lib.nm_set_conv_mode(c.handle, 1), "mode"); _lib.check(lib.nm_op_set_storage16(c.handle, 0, 0), "set16")
        # fp64 reference on the first two frames: the operands the kernels multiply (activated input and weights rounded to fp16)
        nr = min(n, 2)
        xa = x[:nr].float()
        if affine:
            xa = torch.addcmul(sh[:nr, None, None, None, :], xa, sc[:nr, None, None, None, :])
            xa = torch.maximum(xa, xa * 0.01)
        xa = xa.half().double().permute(0, 4, 1, 2, 3)
        ref = F.conv3d(xa, w.half().double(), b.double(), padding=1).permute(0, 2, 3, 4, 1)
        if h: ref_cmp = ref.float().to(dt).double()
        else: ref_cmp = ref
        e0 = (res["0"][0][:nr].double() - ref_cmp).abs().max().item(); e1 = (res["1"][0][:nr].double() - ref_cmp).abs().max().item()
        d = (res["0"][0].float() - res["1"][0].float()).abs().max().item()
        dg = max((res["0"][i] - res["1"][i]).abs().max().item() for i in (1, 2))
        fl = 2.0 * n * size ** 3 * 27 * Cin * Cout
        print("%-5s %3d->%3d @%d^3 x%d%s: f16p<SINGLE> %.3f ms (%.0f TFLOP/s)  f16r %.3f ms (%.0f TFLOP/s)  |f16p - f16r| %.2e  GN scale/shift diff %.2e  "
              "err vs fp64 conv of the fp16 operands: f16p %.2e  f16r %.2e (max |ref| %.1f)  finite %s" % (
            name, Cin, Cout, size, n, "" if affine else " (no affine)", res["0"][3], fl / res["0"][3] / 1e9, res["1"][3], fl / res["1"][3] / 1e9, d, dg, e0, e1,
            ref.abs().max().item(), bool(torch.isfinite(res["1"][0].float()).all())))
