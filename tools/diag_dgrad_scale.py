"""Diagnostic: data-gradient conv accuracy vs the magnitude / smoothness of dy, both conv modes."""
import sys, os
import torch, torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from neural_marionette_amd import _lib
from test_ops_gpu import to_cl, from_cl, relerr, dev

cfg = _lib.NmConfig(device=0, grid_size=64, nkeypoints=24, nlatent=128, nhidden=512, nneighbor=2, gaussian_sigma=1.5, sep_sigma=0.02,
                    vol_fit_chamfer=1, use_graph_traj=1)
ctx = _lib.Context(cfg); ctx.bind_stream()
for (Cin, Cout, ks, pad, size, N) in [(128, 128, 3, 1, 8, 8), (64, 128, 3, 1, 8, 8), (128, 24, 1, 0, 8, 8), (32, 32, 3, 1, 16, 4)]:
    g = torch.Generator().manual_seed(1)
    x = torch.randn(N, Cin, size, size, size, generator=g)
    w = (torch.randn(Cout, Cin, ks, ks, ks, generator=g) / (Cin * ks ** 3) ** 0.5)
    for kind in ("random", "tiny", "smooth", "offset"):
        dy = torch.randn(N, Cout, size, size, size, generator=g)
        if kind == "tiny": dy = dy * 1e-9
        if kind == "smooth": dy = torch.ones_like(dy) * torch.randn(N, Cout, 1, 1, 1, generator=g) + 0.01 * dy
        if kind == "offset": dy = dy * 1e-3 + 1.0
        a = x.clone().double().requires_grad_(True)
        y = F.conv3d(a, w.double(), None, padding=pad)
        y.backward(dy.double())
        for mode in (0, 1):
            _lib.check(ctx.lib.nm_set_conv_mode(ctx.handle, mode), "mode")
            d_in = torch.zeros(N, size, size, size, Cin).cuda(); d_w = torch.zeros_like(w).cuda(); d_b = torch.zeros(Cout).cuda()
            xd, wd, dyd = to_cl(x), dev(w), to_cl(dy, Cout)
            _lib.check(ctx.lib.nm_op_conv3d_backward(ctx.handle, _lib.ptr(xd), N, size, size, size, Cin, None, None, 1.0, _lib.ptr(wd), Cout, ks, 1, pad, 0,
                                                     _lib.ptr(dyd), _lib.ptr(d_in), Cin, _lib.ptr(d_w), _lib.ptr(d_b)), "bwd")
            torch.cuda.synchronize()
            ref = a.grad
            got = from_cl(d_in, Cin).double()
            print("%3d->%3d k%d %-7s mode %d  d_in rel err %.2e   mean-channel-offset err %.2e" % (
                Cin, Cout, ks, kind, mode, (got - ref).abs().max().item() / ref.abs().max().item(),
                ((got - ref).mean(dim=(0, 2, 3, 4)).abs().max() / ref.abs().max()).item()))
