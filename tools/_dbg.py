import sys, os, torch, torch.nn.functional as F
sys.path.insert(0, '/root/repo' if os.path.exists('/root/repo/tests') else os.getcwd())
sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
from neural_marionette_amd import _lib
lib = _lib.load()
cfg = _lib.NmConfig(device=0, grid_size=64, nkeypoints=24, nlatent=128, nhidden=512, nneighbor=2, gaussian_sigma=1.5, sep_sigma=0.02, vol_fit_chamfer=1, use_graph_traj=1)
ctx = _lib.Context(cfg); ctx.bind_stream()
Cin, Cout, size, N = 32, 64, 32, 3
g = torch.Generator().manual_seed(1)
x = torch.randn(N, Cin, size, size, size, generator=g)
w = torch.randn(Cout, Cin, 3, 3, 3, generator=g) / (Cin * 27) ** 0.5
b = torch.randn(Cout, generator=g) * 0.1
ref = F.conv3d(x, w, b, padding=1)
xd = x.permute(0, 2, 3, 4, 1).contiguous().cuda(); wd = w.cuda(); bd = b.cuda()
out = torch.full((N, size, size, size, Cout), float('nan')).cuda()
gam = torch.ones(Cout).cuda(); bet = torch.zeros(Cout).cuda(); gsc = torch.zeros(N, Cout).cuda(); gsh = torch.zeros(N, Cout).cuda()
_lib.check(lib.nm_op_conv3d(ctx.handle, xd.data_ptr(), N, size, size, size, Cin, 0, 0, 1.0, wd.data_ptr(), bd.data_ptr(), Cout, 3, 1, 1, out.data_ptr(), 4, gam.data_ptr(), bet.data_ptr(), gsc.data_ptr(), gsh.data_ptr(), 0), "conv")
torch.cuda.synchronize()
got = out.cpu().permute(0, 4, 1, 2, 3)
err = (got - ref).abs()
print("max err", err.max().item(), "nan", torch.isnan(got).sum().item())
# per (n, cout group, brick) max error
e = err.view(N, 2, 32, 8, 4, 4, 8, 4, 8).amax(dim=(2, 4, 6, 8))   # n, cg, bz, by, bx
bad = (e > 1e-3).nonzero()
print("bad bricks", len(bad), "of", e.numel())
print(bad[:40].tolist())
# within a bad brick: which z rows / y / x / channels
if len(bad):
    n, cg, bz, by, bx = bad[0].tolist()
    blk = err[n, cg*32:(cg+1)*32, bz*4:(bz+1)*4, by*8:(by+1)*8, bx*8:(bx+1)*8]
    print("per channel", (blk.amax(dim=(1,2,3)) > 1e-3).int().tolist())
    print("per z", blk.amax(dim=(0,2,3)).tolist()); print("per y", blk.amax(dim=(0,1,3)).tolist()); print("per x", blk.amax(dim=(0,1,2)).tolist())
