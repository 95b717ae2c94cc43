"""Op-level parity: each HIP kernel family, called through the C ABI, against the same ATen
CPU op the oracle (and the reference) uses.  fp32 tolerance: 2e-5 relative to the output's
max magnitude (fp32 accumulation-order noise; the 1e-4 budget of BASELINE.json is for the
end-to-end keypoints / latents)."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

REL = 2e-5


@pytest.fixture(scope="module")
def ctx():
    from neural_marionette_amd import _lib
    cfg = _lib.NmConfig(device=0, grid_size=64, nkeypoints=24, nlatent=128, nhidden=512, nneighbor=2,
                        gaussian_sigma=1.5, sep_sigma=0.02, vol_fit_chamfer=1, use_graph_traj=1)
    c = _lib.Context(cfg)
    c.bind_stream()
    yield c
    c.close()


def to_cl(x, cpad=None):
    """NCDHW cpu -> channels-last device tensor (channels zero-padded to a multiple of 8)."""
    N, Cc = x.shape[:2]
    cp = cpad or ((Cc + 7) // 8 * 8)
    y = torch.zeros(N, *x.shape[2:], cp)
    y[..., :Cc] = x.permute(0, 2, 3, 4, 1)
    return y.contiguous().cuda()


def from_cl(y, Cc):
    return y[..., :Cc].permute(0, 4, 1, 2, 3).contiguous().cpu()


def relerr(a, b):
    return (a - b).abs().max().item() / max(b.abs().max().item(), 1e-30)


def dev(t):
    return None if t is None else t.contiguous().cuda()


CONV_CASES = [
    # Cin, Cout, ks, stride, pad, size, N, prologue, groups
    (32, 64, 3, 1, 1, 16, 2, False, 4),
    (64, 64, 3, 1, 1, 16, 1, True, 4),
    (4, 32, 5, 1, 2, 16, 2, False, 2),
    (32, 32, 2, 2, 0, 16, 2, True, 2),
    (64, 128, 1, 1, 0, 8, 2, False, 8),
    (48, 72, 3, 1, 1, 4, 3, True, 4),
    (72, 72, 3, 1, 1, 2, 3, False, 4),
    (72, 48, 3, 1, 1, 5, 2, True, 3),
    (128, 24, 1, 1, 0, 8, 2, True, 0),
    (48, 48, 2, 2, 0, 5, 2, False, 3),
    (128, 64, 3, 1, 1, 12, 1, True, 4),
    (256, 256, 3, 1, 1, 6, 1, False, 16),
    (184, 128, 1, 1, 0, 10, 1, False, 0),
    # conv_f16p (8x8x8 bricks, persistent stream): several bricks / cout groups / channel chunks per workgroup
    (32, 64, 3, 1, 1, 32, 3, True, 4),
    (64, 32, 3, 1, 1, 16, 40, True, 2),
    (16, 32, 3, 1, 1, 24, 2, False, 2),
    (32, 96, 3, 1, 1, 16, 33, False, 3),
    # conv_f16r in the one-product mode (32 output channels, weights resident in LDS; Cin = 64 is the (64, 32, .., 16, 40) case above)
    (32, 32, 3, 1, 1, 32, 3, True, 2),
    (32, 32, 3, 1, 1, 16, 5, False, 2),
    # conv_pool_f16s (k2 s2, lanes load their own operands): two N tiles, ragged bricks, Cout not a multiple of 32
    (64, 64, 2, 2, 0, 16, 2, True, 4),
    (32, 32, 2, 2, 0, 20, 2, True, 2),
    (16, 72, 2, 2, 0, 24, 1, False, 3),
]


F16_REL = 3e-3   # conv mode 3 (fp16 products) against the fp32 reference: operands carry 11 significant bits


@pytest.mark.parametrize("mode", [0, 1, 2, 3], ids=["fp32mfma", "split16", "split16-f16p", "f16"])
@pytest.mark.parametrize("case", CONV_CASES, ids=lambda c: "ci%d_co%d_k%d_s%d_d%d" % (c[0], c[1], c[2], c[3], c[5]))
def test_conv3d(ctx, case, mode):
    from neural_marionette_amd import _lib
    _lib.check(ctx.lib.nm_set_conv_mode(ctx.handle, mode), "set_conv_mode")
    Cin, Cout, ks, stride, pad, size, N, prologue, groups = case
    g = torch.Generator().manual_seed(hash(case) & 0xFFFF)
    dims = (size, size + (1 if size % 2 else 0) * 0, size)
    x = torch.randn(N, Cin, *dims, generator=g)
    w = torch.randn(Cout, Cin, ks, ks, ks, generator=g) / (Cin * ks ** 3) ** 0.5
    b = torch.randn(Cout, generator=g) * 0.1
    cp = (Cin + 7) // 8 * 8
    if prologue:
        sc = torch.rand(N, Cin, generator=g) + 0.5
        sh = torch.randn(N, Cin, generator=g) * 0.3
        xin = F.leaky_relu(x * sc[:, :, None, None, None] + sh[:, :, None, None, None], 0.01)
        scp = torch.zeros(N, cp); scp[:, :Cin] = sc
        shp = torch.zeros(N, cp); shp[:, :Cin] = sh
        slope = 0.01
    else:
        xin, scp, shp, slope = x, None, None, 1.0
    ref = F.conv3d(xin, w, b, stride=stride, padding=pad)
    od = ref.shape[2:]
    out = torch.full((N, *od, Cout), float("nan")).cuda()
    gam = torch.rand(Cout, generator=g) + 0.5
    bet = torch.randn(Cout, generator=g) * 0.2
    gsc = torch.zeros(N, Cout).cuda(); gsh = torch.zeros(N, Cout).cuda()
    xd, wd, bd, scd, shd, gd, btd = to_cl(x), dev(w), dev(b), dev(scp), dev(shp), dev(gam), dev(bet)
    _lib.check(ctx.lib.nm_op_conv3d(ctx.handle, _lib.ptr(xd), N, *dims, Cin, _lib.ptr(scd), _lib.ptr(shd), slope,
                                    _lib.ptr(wd), _lib.ptr(bd), Cout, ks, stride, pad, _lib.ptr(out), groups,
                                    _lib.ptr(gd), _lib.ptr(btd), _lib.ptr(gsc), _lib.ptr(gsh), 0), "op_conv3d")
    torch.cuda.synchronize()
    got = from_cl(out, Cout)
    assert torch.isfinite(got).all(), "unwritten / non-finite outputs"
    e = relerr(got, ref)
    if mode == 3:
        # the product of the operands rounded to fp16 (activation first, then the rounding), accumulated in fp32: sharp against that
        # definition; layers the mode leaves on the fp32 cores (Cin % 16 != 0, k = 1, tiny volumes) match the unrounded reference
        # (the kernels' affine is one fma: the exact x * scale + shift rounded once, which decides a few fp16 roundings differently
        #  from ATen's multiply-then-add)
        xin16 = xin if not prologue else F.leaky_relu((x.double() * sc.double()[:, :, None, None, None] + sh.double()[:, :, None, None, None]).float(), 0.01)
        ref16 = F.conv3d(xin16.half().double(), w.half().double(), b.double(), stride=stride, padding=pad).float()
        e16 = relerr(got, ref16)
        assert min(e, e16) < REL, f"f16-product conv: rel err {e16:.3e} to the rounded-operand product, {e:.3e} to fp32"
        assert e < F16_REL, f"f16-product conv vs fp32 reference rel err {e:.3e}"
        if groups:
            refn = F.group_norm(ref, groups, gam, bet, 1e-5)
            gotn = got * gsc.cpu()[:, :, None, None, None] + gsh.cpu()[:, :, None, None, None]
            assert relerr(gotn, refn) < F16_REL
        _lib.check(ctx.lib.nm_set_conv_mode(ctx.handle, 1), "set_conv_mode")
        return
    assert e < REL, f"conv raw output rel err {e:.3e}"
    if groups:
        refn = F.group_norm(ref, groups, gam, bet, 1e-5)
        gotn = got * gsc.cpu()[:, :, None, None, None] + gsh.cpu()[:, :, None, None, None]
        e = relerr(gotn, refn)
        assert e < REL, f"fused GroupNorm rel err {e:.3e}"


def test_conv3d_deferred_epilogue_variant_of_conv_f16p2():
    """NM355_P2_DEFER=1: conv_f16p2 with ONE accumulator per tile and the finished brick's epilogue inside the next brick's first step
    (built, measured slower, not the default - DESIGN 4 / profiles/r06_p2_defer_ab.txt): the 64-output-channel brick-aligned cases again
    on a context created with the switch, same 2e-5 bounds for the raw output and the fused GroupNorm statistics."""
    import os
    from neural_marionette_amd import _lib
    old = os.environ.get("NM355_P2_DEFER")
    os.environ["NM355_P2_DEFER"] = "1"
    try:
        cfg = _lib.NmConfig(device=0, grid_size=64, nkeypoints=24, nlatent=128, nhidden=512, nneighbor=2,
                            gaussian_sigma=1.5, sep_sigma=0.02, vol_fit_chamfer=1, use_graph_traj=1)
        c = _lib.Context(cfg)
        c.bind_stream()
        try:
            for case in [(32, 64, 3, 1, 1, 32, 3, True, 4), (64, 64, 3, 1, 1, 16, 1, True, 4), (32, 64, 3, 1, 1, 16, 2, False, 4),
                         (16, 64, 3, 1, 1, 16, 3, True, 4), (128, 128, 3, 1, 1, 16, 2, True, 8)]:
                test_conv3d(c, case, 1)
        finally:
            c.close()
    finally:
        if old is None:
            os.environ.pop("NM355_P2_DEFER", None)
        else:
            os.environ["NM355_P2_DEFER"] = old


@pytest.mark.parametrize("mode", [0, 1, 3], ids=["fp32mfma", "split16", "f16"])
@pytest.mark.parametrize("Cin,Cout,size,prologue,N", [(128, 64, 8, False, 2), (64, 32, 12, True, 2), (16, 32, 5, True, 2),
                                                       (32, 32, 16, True, 5), (48, 32, 8, False, 40), (64, 64, 16, True, 3)])
def test_conv3d_fused_upsample(ctx, Cin, Cout, size, prologue, N, mode):
    """Upsample(x2, trilinear, align_corners=False) -> Conv3d(k3) -> GroupNorm, the upsampling fused into the
    conv's staging (decoder layers .0/.1 and .7/.8 of kypt_detector.py:427-444)."""
    from neural_marionette_amd import _lib
    _lib.check(ctx.lib.nm_set_conv_mode(ctx.handle, mode), "set_conv_mode")
    g = torch.Generator().manual_seed(Cin + size)
    x = torch.randn(N, Cin, size, size, size, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, 3, generator=g) / (Cin * 27) ** 0.5
    b = torch.randn(Cout, generator=g) * 0.1
    if prologue:
        sc = torch.rand(N, Cin, generator=g) + 0.5
        sh = torch.randn(N, Cin, generator=g) * 0.3
        xin = F.leaky_relu(x * sc[:, :, None, None, None] + sh[:, :, None, None, None], 0.01)
        slope = 0.01
    else:
        sc = sh = None
        xin, slope = F.leaky_relu(x, 0.01), 0.01
    ref = F.conv3d(F.interpolate(xin, scale_factor=2.0, mode="trilinear", align_corners=False), w, b, padding=1)
    groups = Cout // 16
    gam = torch.rand(Cout, generator=g) + 0.5
    bet = torch.randn(Cout, generator=g) * 0.2
    out = torch.full((N, 2 * size, 2 * size, 2 * size, Cout), float("nan")).cuda()
    gsc = torch.zeros(N, Cout).cuda(); gsh = torch.zeros(N, Cout).cuda()
    xd, wd, bd, scd, shd, gd, btd = to_cl(x), dev(w), dev(b), dev(sc), dev(sh), dev(gam), dev(bet)
    _lib.check(ctx.lib.nm_op_conv3d(ctx.handle, _lib.ptr(xd), N, size, size, size, Cin, _lib.ptr(scd), _lib.ptr(shd), slope,
                                    _lib.ptr(wd), _lib.ptr(bd), Cout, 3, 1, 1, _lib.ptr(out), groups, _lib.ptr(gd),
                                    _lib.ptr(btd), _lib.ptr(gsc), _lib.ptr(gsh), 1), "op_conv3d(up2)")
    torch.cuda.synchronize()
    got = from_cl(out, Cout)
    assert torch.isfinite(got).all()
    tol = F16_REL if mode == 3 else REL
    _lib.check(ctx.lib.nm_set_conv_mode(ctx.handle, 1), "set_conv_mode")
    e = relerr(got, ref)
    assert e < tol, f"fused upsample+conv rel err {e:.3e}"
    refn = F.group_norm(ref, groups, gam, bet, 1e-5)
    e = relerr(got * gsc.cpu()[:, :, None, None, None] + gsh.cpu()[:, :, None, None, None], refn)
    assert e < tol, f"fused GroupNorm rel err {e:.3e}"


@pytest.mark.parametrize("mode", [1, 3], ids=["split16", "f16"])
@pytest.mark.parametrize("dims,prologue,N", [((8, 8, 8), True, 3), ((2, 8, 16), False, 2), ((16, 16, 16), True, 2), ((6, 24, 8), True, 1),
                                              ((32, 32, 32), True, 2), ((4, 40, 8), False, 5)])
def test_conv3d_fused_upsample_composite(ctx, dims, prologue, N, mode):
    """The 64 -> 32 resolution-doubling decoder layer (kypt_detector.py:441-447) on the coarse grid with composite weights
    (nm_up2c.hip): eight parity-class 3x3x3 convolutions + the signed shell corrections that restore the fine conv's zero
    padding.  Checked against ATen's Upsample -> Conv3d -> GroupNorm, separately on the outer shell (where the corrections
    act: faces, edges and corners of every side) and on the interior."""
    from neural_marionette_amd import _lib
    _lib.check(ctx.lib.nm_set_conv_mode(ctx.handle, mode), "set_conv_mode")
    Cin, Cout = 64, 32
    D, H, W = dims
    g = torch.Generator().manual_seed(D * 100 + H * 10 + W)
    x = torch.randn(N, Cin, D, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, 3, generator=g) / (Cin * 27) ** 0.5
    b = torch.randn(Cout, generator=g) * 0.1
    if prologue:
        sc = torch.rand(N, Cin, generator=g) + 0.5
        sh = torch.randn(N, Cin, generator=g) * 0.3
        xin = F.leaky_relu(x * sc[:, :, None, None, None] + sh[:, :, None, None, None], 0.01)
    else:
        sc = sh = None
        xin = F.leaky_relu(x, 0.01)
    ref = F.conv3d(F.interpolate(xin, scale_factor=2.0, mode="trilinear", align_corners=False), w, b, padding=1)
    groups = 2
    gam = torch.rand(Cout, generator=g) + 0.5
    bet = torch.randn(Cout, generator=g) * 0.2
    out = torch.full((N, 2 * D, 2 * H, 2 * W, Cout), float("nan")).cuda()
    gsc = torch.zeros(N, Cout).cuda(); gsh = torch.zeros(N, Cout).cuda()
    xd, wd, bd, scd, shd, gd, btd = to_cl(x), dev(w), dev(b), dev(sc), dev(sh), dev(gam), dev(bet)
    _lib.check(ctx.lib.nm_op_conv3d(ctx.handle, _lib.ptr(xd), N, D, H, W, Cin, _lib.ptr(scd), _lib.ptr(shd), 0.01,
                                    _lib.ptr(wd), _lib.ptr(bd), Cout, 3, 1, 1, _lib.ptr(out), groups, _lib.ptr(gd),
                                    _lib.ptr(btd), _lib.ptr(gsc), _lib.ptr(gsh), 1), "op_conv3d(up2)")
    torch.cuda.synchronize()
    got = from_cl(out, Cout)
    assert torch.isfinite(got).all()
    scale = ref.abs().max().item()
    err = (got - ref).abs()
    inner = err[:, :, 1:-1, 1:-1, 1:-1].max().item() / scale
    shell = err.clone(); shell[:, :, 1:-1, 1:-1, 1:-1] = 0
    corners = err[:, :, ::2 * D - 1, ::2 * H - 1, ::2 * W - 1].max().item() / scale
    print("composite up2 conv: interior %.2e shell %.2e corners %.2e" % (inner, shell.max().item() / scale, corners))
    tol = F16_REL if mode == 3 else REL
    assert inner < tol, f"interior rel err {inner:.3e}"
    assert shell.max().item() / scale < tol, f"shell rel err {shell.max().item() / scale:.3e}"
    refn = F.group_norm(ref, groups, gam, bet, 1e-5)
    e = relerr(got * gsc.cpu()[:, :, None, None, None] + gsh.cpu()[:, :, None, None, None], refn)
    assert e < tol, f"fused GroupNorm rel err {e:.3e}"
    # the same launch twice: bit-identical (fixed-order partial sums, no atomics)
    out2 = torch.full_like(out, float("nan")); gsc2 = torch.zeros_like(gsc); gsh2 = torch.zeros_like(gsh)
    _lib.check(ctx.lib.nm_op_conv3d(ctx.handle, _lib.ptr(xd), N, D, H, W, Cin, _lib.ptr(scd), _lib.ptr(shd), 0.01,
                                    _lib.ptr(wd), _lib.ptr(bd), Cout, 3, 1, 1, _lib.ptr(out2), groups, _lib.ptr(gd),
                                    _lib.ptr(btd), _lib.ptr(gsc2), _lib.ptr(gsh2), 1), "op_conv3d(up2)")
    torch.cuda.synchronize()
    _lib.check(ctx.lib.nm_set_conv_mode(ctx.handle, 1), "set_conv_mode")
    assert torch.equal(out, out2) and torch.equal(gsc, gsc2) and torch.equal(gsh, gsh2)


@pytest.mark.parametrize("case", [(32, 32, 3, 1, 1, (16, 16, 16), 0), (64, 64, 3, 1, 1, (16, 16, 16), 0), (64, 128, 1, 1, 0, (8, 8, 8), 0),
                                  (32, 32, 2, 2, 0, (16, 16, 16), 0), (64, 32, 3, 1, 1, (8, 8, 8), 1), (48, 72, 3, 1, 1, (4, 4, 4), 0)],
                         ids=lambda c: "ci%d_co%d_k%d_up%d" % (c[0], c[1], c[2], c[6]))
def test_conv3d_beyond_fp16_range_falls_back_to_fp32(ctx, case):
    """Split-fp16 mode with activations beyond the fp16 range (|x| up to ~3e5 > 65504): hi = fp16(x) would be inf and the
    product NaN.  The op-level entry point scans its result and re-runs the launch on the exact fp32 MFMA path: the result
    matches ATen's fp32 conv, never NaN.  (Network level: test_network_gpu.py::test_range_guard_reports_overflow.)"""
    from neural_marionette_amd import _lib
    _lib.check(ctx.lib.nm_set_conv_mode(ctx.handle, 1), "set_conv_mode")
    Cin, Cout, ks, stride, pad, dims, up2 = case
    g = torch.Generator().manual_seed(Cin + Cout)
    N = 2
    x = torch.randn(N, Cin, *dims, generator=g)
    x[0] *= 1e5                                        # frame 0 far outside the fp16 range, frame 1 ordinary
    w = torch.randn(Cout, Cin, ks, ks, ks, generator=g) / (Cin * ks ** 3) ** 0.5
    b = torch.randn(Cout, generator=g) * 0.1
    xin = F.interpolate(x, scale_factor=2.0, mode="trilinear", align_corners=False) if up2 else x
    ref = F.conv3d(xin, w, b, stride=stride, padding=pad)
    out = torch.full((N, *ref.shape[2:], Cout), float("nan")).cuda()
    groups = Cout // 16
    gam = torch.ones(Cout); bet = torch.zeros(Cout)
    gsc = torch.zeros(N, Cout).cuda(); gsh = torch.zeros(N, Cout).cuda()
    xd, wd, bd, gd, btd = to_cl(x), dev(w), dev(b), dev(gam), dev(bet)
    _lib.check(ctx.lib.nm_op_conv3d(ctx.handle, _lib.ptr(xd), N, *dims, Cin, None, None, 1.0, _lib.ptr(wd), _lib.ptr(bd), Cout, ks,
                                    stride, pad, _lib.ptr(out), groups, _lib.ptr(gd), _lib.ptr(btd), _lib.ptr(gsc), _lib.ptr(gsh), up2),
               "op_conv3d")
    torch.cuda.synchronize()
    got = from_cl(out, Cout)
    assert torch.isfinite(got).all() and torch.isfinite(gsc).all() and torch.isfinite(gsh).all()
    for n in range(N):                                 # per frame: frame 1's magnitudes are 1e5 times smaller
        e = (got[n] - ref[n]).abs().max().item() / ref[n].abs().max().item()
        assert e < REL, f"frame {n}: rel err {e:.3e}"
    assert ctx.lib.nm_ctx_check_nonfinite(ctx.handle) == 0      # the op cleared the status it consumed


@pytest.mark.parametrize("G,Cout,N", [(16, 32, 3), (24, 64, 2), (32, 32, 1)])
@pytest.mark.parametrize("mode", [0, 1], ids=["fp32mfma", "split16"])
def test_conv5_occupancy_first_layer(ctx, G, Cout, N, mode):
    """Basic3DBlock(1+3 -> Cout, k5) on cat[occ, coords] as conv(occ) + constant field (kypt_detector.py:265).
    split16: two f16 MFMAs per product for 0/1 volumes, a third (lo part of the input) for fractional occupancy."""
    from neural_marionette_amd import _lib
    from oracle import nm_oracle as O
    _lib.check(ctx.lib.nm_set_conv_mode(ctx.handle, mode), "set_conv_mode")
    g = torch.Generator().manual_seed(G + Cout)
    occ = (torch.rand(N, 1, G, G, G, generator=g) < 0.05).float()
    if N > 1:
        occ[1] = torch.rand(1, G, G, G, generator=g)          # the clip-mean net sees fractional occupancy
    w = torch.randn(Cout, 4, 5, 5, 5, generator=g) / (4 * 125) ** 0.5
    b = torch.randn(Cout, generator=g) * 0.1
    ref = F.conv3d(O.add_coords(occ), w, b, padding=2)
    groups = Cout // 16
    gam = torch.rand(Cout, generator=g) + 0.5
    bet = torch.randn(Cout, generator=g) * 0.2
    out = torch.full((N, G, G, G, Cout), float("nan")).cuda()
    gsc = torch.zeros(N, Cout).cuda(); gsh = torch.zeros(N, Cout).cuda()
    od, wd, bd, gd, btd = occ.reshape(N, G, G, G).contiguous().cuda(), dev(w), dev(b), dev(gam), dev(bet)
    _lib.check(ctx.lib.nm_op_conv5_occ(ctx.handle, _lib.ptr(od), N, G, _lib.ptr(wd), _lib.ptr(bd), Cout, _lib.ptr(out),
                                       groups, _lib.ptr(gd), _lib.ptr(btd), _lib.ptr(gsc), _lib.ptr(gsh)), "op_conv5_occ")
    torch.cuda.synchronize()
    got = from_cl(out, Cout)
    assert torch.isfinite(got).all()
    e = relerr(got, ref)
    assert e < REL, f"occupancy first layer rel err {e:.3e}"
    refn = F.group_norm(ref, groups, gam, bet, 1e-5)
    e = relerr(got * gsc.cpu()[:, :, None, None, None] + gsh.cpu()[:, :, None, None, None], refn)
    assert e < REL, f"GroupNorm rel err {e:.3e}"


@pytest.mark.parametrize("mode", [0, 1, 3], ids=["fp32mfma", "split16", "f16"])
@pytest.mark.parametrize("Cin,Cout,size,outpad,groups,N", [(72, 48, 2, 0, 3, 2), (48, 32, 5, 1, 2, 2), (32, 64, 8, 0, 4, 2), (32, 128, 3, 1, 8, 2), (32, 64, 32, 0, 4, 2),
                                                            # the f16 matrix-core kernel (conv modes 1 / 3: Cin % 16 == 0, Cout % 32 == 0, >= 4096 coarse voxels)
                                                            (64, 32, 16, 0, 4, 3), (128, 64, 8, 0, 4, 8), (48, 96, 16, 0, 4, 1)])
def test_convT2(ctx, Cin, Cout, size, outpad, groups, N, mode):
    from neural_marionette_amd import _lib
    if mode != 1 and Cin * Cout * size < 64 * 32 * 16 and size != 8:
        pytest.skip("small shapes: every mode runs the same fp32 kernel (covered in split16)")
    _lib.check(ctx.lib.nm_set_conv_mode(ctx.handle, mode), "set_conv_mode")
    g = torch.Generator().manual_seed(Cin * 131 + Cout)
    x = torch.randn(N, Cin, size, size, size, generator=g)
    w = torch.randn(Cin, Cout, 2, 2, 2, generator=g) / (Cin * 8) ** 0.5
    b = torch.randn(Cout, generator=g) * 0.1
    gam = torch.rand(Cout, generator=g) + 0.5
    bet = torch.randn(Cout, generator=g) * 0.2
    ref = F.conv_transpose3d(x, w, b, stride=2, output_padding=outpad)
    od = ref.shape[2]
    out = torch.full((N, od, od, od, Cout), float("nan")).cuda()
    gsc = torch.zeros(N, Cout).cuda(); gsh = torch.zeros(N, Cout).cuda()
    xd, wd, bd, gd, btd = to_cl(x), dev(w), dev(b), dev(gam), dev(bet)
    try:
        _lib.check(ctx.lib.nm_op_convT2(ctx.handle, _lib.ptr(xd), N, size, size, size, Cin, _lib.ptr(wd), _lib.ptr(bd),
                                        Cout, outpad, _lib.ptr(out), groups, _lib.ptr(gd), _lib.ptr(btd),
                                        _lib.ptr(gsc), _lib.ptr(gsh)), "op_convT2")
        torch.cuda.synchronize()
    finally:
        _lib.check(ctx.lib.nm_set_conv_mode(ctx.handle, 1), "set_conv_mode")
    got = from_cl(out, Cout)
    assert torch.isfinite(got).all(), "unwritten / non-finite outputs"
    e = relerr(got, ref)
    refn = F.group_norm(ref, groups, gam, bet, 1e-5)
    gotn = got * gsc.cpu()[:, :, None, None, None] + gsh.cpu()[:, :, None, None, None]
    if mode == 3:
        # sharp against the product of the operands rounded to fp16 (the mode's definition) where the f16 kernel runs, else against fp32
        ref16 = F.conv_transpose3d(x.half().double(), w.half().double(), b.double(), stride=2, output_padding=outpad).float()
        e16 = relerr(got, ref16)
        assert min(e, e16) < REL, f"f16-product convT: rel err {e16:.3e} to the rounded-operand product, {e:.3e} to fp32"
        assert e < F16_REL and relerr(gotn, refn) < F16_REL
        return
    assert e < REL, f"convT raw rel err {e:.3e}"
    e = relerr(gotn, refn)
    assert e < REL, f"convT GroupNorm rel err {e:.3e}"


def test_apply2(ctx):
    from neural_marionette_amd import _lib
    g = torch.Generator().manual_seed(5)
    N, V, Cc = 3, 125, 48
    a = torch.randn(N, V, Cc, generator=g); b = torch.randn(N, V, Cc, generator=g)
    sa = torch.rand(N, Cc, generator=g) + 0.5; ha = torch.randn(N, Cc, generator=g)
    sb = torch.rand(N, Cc, generator=g) + 0.5; hb = torch.randn(N, Cc, generator=g)
    ref = F.leaky_relu(a * sa[:, None] + ha[:, None], 0.01) + (b * sb[:, None] + hb[:, None])
    out = torch.empty(N, V, Cc).cuda()
    t = [dev(v) for v in (a, sa, ha, b, sb, hb)]
    _lib.check(ctx.lib.nm_op_apply2(ctx.handle, _lib.ptr(t[0]), _lib.ptr(t[1]), _lib.ptr(t[2]), 0.01,
                                    _lib.ptr(t[3]), _lib.ptr(t[4]), _lib.ptr(t[5]), 1.0, N, V, Cc, _lib.ptr(out)), "apply2")
    torch.cuda.synchronize()
    assert relerr(out.cpu(), ref) < 1e-6
    # single-source form
    _lib.check(ctx.lib.nm_op_apply2(ctx.handle, _lib.ptr(t[0]), None, None, 1.0, None, None, None, 1.0, N, V, Cc,
                                    _lib.ptr(out)), "apply2")
    torch.cuda.synchronize()
    assert torch.equal(out.cpu(), a)


@pytest.mark.parametrize("size,Cc", [(4, 16), (5, 8), (8, 128), (12, 64), (4, 32), (6, 32)])
def test_upsample2(ctx, size, Cc):
    from neural_marionette_amd import _lib
    g = torch.Generator().manual_seed(size)
    x = torch.randn(2, Cc, size, size, size, generator=g)
    ref = F.interpolate(x, scale_factor=2.0, mode="trilinear", align_corners=False)
    out = torch.empty(2, 2 * size, 2 * size, 2 * size, Cc).cuda()
    xd = to_cl(x)
    _lib.check(ctx.lib.nm_op_upsample2(ctx.handle, _lib.ptr(xd), 2, size, size, size, Cc, _lib.ptr(out)), "upsample2")
    torch.cuda.synchronize()
    e = relerr(from_cl(out, Cc), ref)
    assert e < 2e-6, f"trilinear rel err {e:.3e}"


def test_pack_input(ctx):
    from neural_marionette_amd import _lib
    from oracle import nm_oracle as O
    B, T, G = 2, 3, 16
    vox = (torch.rand(B, T, 1, G, G, G) < 0.1).float()
    vd = vox.cuda()
    out = torch.empty(B * T, G, G, G, 8).cuda()
    _lib.check(ctx.lib.nm_op_pack_input(ctx.handle, _lib.ptr(vd), B, T, G, 0, _lib.ptr(out)), "pack_input")
    torch.cuda.synchronize()
    ref = O.add_coords(vox.view(B * T, 1, G, G, G))
    got = out.cpu()
    assert torch.equal(from_cl(out, 4)[:, 0], ref[:, 0])
    assert (from_cl(out, 4)[:, 1:] - ref[:, 1:]).abs().max().item() <= 6e-8
    assert got[..., 4:].abs().max().item() == 0.0
    out2 = torch.empty(B, G, G, G, 8).cuda()
    _lib.check(ctx.lib.nm_op_pack_input(ctx.handle, _lib.ptr(vd), B, T, G, 1, _lib.ptr(out2)), "pack_input")
    torch.cuda.synchronize()
    assert torch.equal(from_cl(out2, 4)[:, 0], vox.mean(dim=1)[:, 0])


def test_cl_to_ncdhw(ctx):
    from neural_marionette_amd import _lib
    x = torch.randn(2, 100, 40)
    out = torch.empty(2, 40, 100).cuda()
    xd = x.cuda()
    _lib.check(ctx.lib.nm_op_cl_to_ncdhw(ctx.handle, _lib.ptr(xd), 2, 100, 40, _lib.ptr(out)), "cl_to_ncdhw")
    torch.cuda.synchronize()
    assert torch.equal(out.cpu(), x.permute(0, 2, 1))


def test_split16_matches_fp32_on_wide_dynamic_range(ctx):
    """The split-fp16 path must stay fp32-equivalent for tiny weights (fp16 subnormal range) and large activations."""
    from neural_marionette_amd import _lib
    g = torch.Generator().manual_seed(99)
    N, Cin, Cout, size = 1, 64, 64, 12
    x = torch.randn(N, Cin, size, size, size, generator=g) * torch.logspace(-3, 2, Cin)[None, :, None, None, None]
    w = torch.randn(Cout, Cin, 3, 3, 3, generator=g) * torch.logspace(-5, -1, Cout)[:, None, None, None, None]
    b = torch.zeros(Cout)
    ref = F.conv3d(x.double(), w.double(), None, padding=1)
    outs = []
    for mode in (0, 1):
        _lib.check(ctx.lib.nm_set_conv_mode(ctx.handle, mode), "set_conv_mode")
        out = torch.full((N, size, size, size, Cout), float("nan")).cuda()
        xd, wd, bd = to_cl(x), dev(w), dev(b)
        _lib.check(ctx.lib.nm_op_conv3d(ctx.handle, _lib.ptr(xd), N, size, size, size, Cin, None, None, 1.0, _lib.ptr(wd),
                                        _lib.ptr(bd), Cout, 3, 1, 1, _lib.ptr(out), 0, None, None, None, None, 0), "op_conv3d")
        torch.cuda.synchronize()
        outs.append(from_cl(out, Cout).double())
    # per output channel relative error (channels span 4 decades of weight scale)
    scale = ref.abs().amax(dim=(0, 2, 3, 4)) + 1e-300
    e32 = ((outs[0] - ref).abs().amax(dim=(0, 2, 3, 4)) / scale).max().item()
    e16 = ((outs[1] - ref).abs().amax(dim=(0, 2, 3, 4)) / scale).max().item()
    print("per-channel rel err vs fp64: fp32 MFMA %.3e, split-fp16 %.3e" % (e32, e16))
    assert e32 < 2e-5 and e16 < 2e-5
    _lib.check(ctx.lib.nm_set_conv_mode(ctx.handle, 1), "set_conv_mode")
