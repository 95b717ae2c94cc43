import sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from test_ops_gpu import to_cl, dev
from neural_marionette_amd import _lib
BF = torch.bfloat16
cfg = _lib.NmConfig(device=0, grid_size=64, nkeypoints=24, nlatent=128, nhidden=512, nneighbor=2, gaussian_sigma=1.5, sep_sigma=0.02, vol_fit_chamfer=1, use_graph_traj=1)
ctx = _lib.Context(cfg); ctx.bind_stream()
Cin, Cout, size, N = 64, 32, 16, 1
# x[c] = c + 1 everywhere (constant per channel): the upsampled value is c + 1; weight: centre tap picks channel co + off
x = torch.arange(1, Cin + 1).float().view(1, Cin, 1, 1, 1).expand(N, Cin, size, size, size).contiguous()
keep = []
def run(mode, ih, oh, off, affine):
    w = torch.zeros(Cout, Cin, 3, 3, 3)
    for co in range(Cout):
        w[co, co + off, 1, 1, 1] = 1.0
    _lib.check(ctx.lib.nm_set_conv_mode(ctx.handle, mode), "m"); _lib.check(ctx.lib.nm_op_set_storage16(ctx.handle, ih, oh), "s")
    xd = to_cl(x); xd = xd.to(BF) if ih else xd
    wd, bd = dev(w), dev(torch.zeros(Cout))
    scd = dev(torch.ones(N, Cin) * 2.0) if affine else None; shd = dev(torch.zeros(N, Cin)) if affine else None
    keep.extend([xd, wd, bd, scd, shd])
    out = torch.zeros(N, 32, 32, 32, Cout, dtype=BF if oh else torch.float32, device="cuda")
    _lib.check(ctx.lib.nm_op_conv3d(ctx.handle, _lib.ptr(xd), N, size, size, size, Cin, _lib.ptr(scd), _lib.ptr(shd), 0.01 if affine else 1.0, _lib.ptr(wd), _lib.ptr(bd), Cout, 3, 1, 1,
                                    _lib.ptr(out), 0, None, None, None, None, 1), "conv")
    torch.cuda.synchronize()
    return out.float().cpu()
for off in (0, 32):
    for affine in (False, True):
        ref = run(3, 0, 0, off, affine)
        got = run(4, 1, 0, off, affine)
        print("off", off, "affine", affine, "ref centre", ref[0, 8, 8, 8, :10].tolist())
        print("                     got centre", got[0, 8, 8, 8, :10].tolist(), "... ", got[0, 8, 8, 8, 24:].tolist())
