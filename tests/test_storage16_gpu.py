"""16-bit storage (conv mode 4 / 'bf16': BASELINE config 3 as named): activations and activation gradients of the training path with
at least 32^3 voxels per frame are stored as bfloat16; fp32 master weights, GroupNorm statistics, accumulators, Adam.

Op level (through the C ABI, nm_op_set_storage16): a bfloat16 -> fp32 conversion is exact and a store rounds to nearest even, so every
16-bit-storage kernel must agree BIT FOR BIT with its fp32-storage instantiation (conv mode 3) fed the same bfloat16-representable
values: fp32 outputs (weight / bias / GroupNorm-parameter gradients) equal, bfloat16 outputs equal to the RNE rounding of the fp32
result.  Where a bfloat16 tensor is read-modify-written (the shell of the fused-upsample layer) or an intermediate is stored in between
(the fine-grid data gradient in front of the upsample adjoint), the bound is the intermediate's rounding: 2^-8 relative.

Network level: gradients of the training loss against the fp64 oracle at 32^3 with the storage threshold lowered to 16^3 (the tensor
population of the 64^3 network one level down), whole-gradient relative L2 error with a stated bound; batch additivity, run-to-run
identity and the halved training arena at the bench shape (64^3, B = 4, T = 16)."""
import ctypes as C
import os

import pytest
import torch
import torch.nn.functional as F

from test_ops_gpu import ctx, to_cl, from_cl, relerr, dev  # noqa: F401  (ctx is a fixture)
from test_train_detector_gpu import AIST, _setup, _oracle_grads, _hip_grads

pytestmark = pytest.mark.gpu

BF = torch.bfloat16


def _set(ctx, mode, ih, oh):
    from neural_marionette_amd import _lib
    _lib.check(ctx.lib.nm_set_conv_mode(ctx.handle, mode), "set_conv_mode")
    _lib.check(ctx.lib.nm_op_set_storage16(ctx.handle, ih, oh), "op_set_storage16")


def _conv(ctx, xd, N, dims, Cin, scd, shd, slope, wd, bd, Cout, ks, stride, pad, od, groups, gd, btd, up2, out_dtype):
    from neural_marionette_amd import _lib
    out = torch.zeros((N, *od, Cout), dtype=out_dtype, device="cuda")
    gsc = torch.zeros(N, Cout).cuda(); gsh = torch.zeros(N, Cout).cuda()
    _lib.check(ctx.lib.nm_op_conv3d(ctx.handle, _lib.ptr(xd), N, *dims, Cin, _lib.ptr(scd), _lib.ptr(shd), slope, _lib.ptr(wd), _lib.ptr(bd),
                                    Cout, ks, stride, pad, _lib.ptr(out), groups, _lib.ptr(gd), _lib.ptr(btd), _lib.ptr(gsc), _lib.ptr(gsh), up2), "op_conv3d")
    torch.cuda.synchronize()
    return out, gsc, gsh


FWD_CASES = [
    # Cin, Cout, ks, stride, pad, size, N, up2, in16, out16, kernel the case is meant to reach
    (64, 64, 3, 1, 1, 32, 2, 0, 1, 1, "conv_f16p2"),
    (32, 64, 3, 1, 1, 16, 3, 0, 1, 1, "conv_f16p2 (two channel chunks, several bricks per workgroup)"),
    (32, 32, 3, 1, 1, 32, 2, 0, 1, 1, "conv_f16r (one-product modes, 32 output channels: resident weights)"),
    (64, 32, 3, 1, 1, 16, 5, 0, 1, 1, "conv_f16r (four chunks)"),
    (32, 64, 1, 1, 0, 32, 2, 0, 1, 1, "conv_f16s k1, two N tiles"),
    (64, 32, 1, 1, 0, 32, 2, 0, 1, 1, "conv_f16s k1, one N tile"),
    (128, 64, 3, 1, 1, 8, 2, 1, 0, 1, "conv_f16s fused upsample, fp32 in / bf16 out"),
    (64, 32, 3, 1, 1, 16, 2, 1, 1, 1, "conv_up2c + face + edge"),
    (32, 32, 2, 2, 0, 32, 2, 0, 1, 1, "conv_pool_f16q bf16 -> bf16"),
    (64, 64, 2, 2, 0, 32, 2, 0, 1, 0, "conv_pool_f16q bf16 -> fp32, two N tiles"),
    (32, 32, 2, 2, 0, 20, 2, 0, 1, 1, "conv_pool_f16q ragged bricks"),
]


@pytest.mark.parametrize("case", FWD_CASES, ids=lambda c: "ci%d_co%d_k%d_d%d_up%d_io%d%d" % (c[0], c[1], c[2], c[5], c[7], c[8], c[9]))
def test_conv3d_storage16_is_bit_identical_to_fp32_storage(ctx, case):
    Cin, Cout, ks, stride, pad, size, N, up2, ih, oh, _ = case
    g = torch.Generator().manual_seed(hash(case[:10]) & 0xFFFF)
    dims = (size, size, size)
    x = torch.randn(N, Cin, *dims, generator=g).to(BF).float()            # bfloat16-representable values
    w = torch.randn(Cout, Cin, ks, ks, ks, generator=g) / (Cin * ks ** 3) ** 0.5
    b = torch.randn(Cout, generator=g) * 0.1
    sc = torch.rand(N, Cin, generator=g) + 0.5
    sh = torch.randn(N, Cin, generator=g) * 0.3
    gam = torch.rand(Cout, generator=g) + 0.5
    bet = torch.randn(Cout, generator=g) * 0.2
    us = 2 if up2 else 1
    od = tuple((us * d + 2 * pad - ks) // stride + 1 for d in dims)
    groups = Cout // 16
    xd32 = to_cl(x)
    xd = xd32.to(BF) if ih else xd32
    args = (N, dims, Cin, dev(sc), dev(sh), 0.01, dev(w), dev(b), Cout, ks, stride, pad, od, groups, dev(gam), dev(bet), up2)
    _set(ctx, 3, 0, 0)
    ref, rsc, rsh = _conv(ctx, xd32, *args, torch.float32)
    _set(ctx, 4, ih, oh)
    try:
        got, gsc, gsh = _conv(ctx, xd, *args, BF if oh else torch.float32)
    finally:
        _set(ctx, 1, 0, 0)
    assert torch.isfinite(got.float()).all()
    want = ref.to(BF).float() if oh else ref
    if up2 and ih:
        # conv_up2c: the shell cells (a fine voxel on the volume border) are the main kernel's rounded value plus the face / edge
        # kernels' correction, rounded again: exact inside, two roundings on the shell
        inner = (slice(None), slice(1, -1), slice(1, -1), slice(1, -1))
        assert torch.equal(got.float()[inner], want[inner]), "interior of the fused-upsample layer"
        e = (got.float() - ref).abs().max().item() / ref.abs().max().item()
        assert e < 2.0 ** -7, e
    else:
        assert torch.equal(got.float(), want), "max diff %.3e" % (got.float() - want).abs().max().item()
    # the GroupNorm statistics come from the fp32 accumulators in both storage types
    assert relerr(gsc.cpu(), rsc.cpu()) < 1e-5 and relerr(gsh.cpu(), rsh.cpu()) < 1e-5


def test_first_layer_storage16(ctx):
    from neural_marionette_amd import _lib
    N, G, Cout = 3, 32, 32
    g = torch.Generator().manual_seed(5)
    occ = (torch.rand(N, G, G, G, generator=g) < 0.05).float().cuda()
    w = (torch.randn(Cout, 4, 5, 5, 5, generator=g) * 0.05).cuda(); b = (torch.randn(Cout, generator=g) * 0.1).cuda()
    gam = torch.ones(Cout).cuda(); bet = torch.zeros(Cout).cuda()

    def run(mode, oh):
        _set(ctx, mode, 0, oh)
        out = torch.zeros(N, G, G, G, Cout, dtype=BF if oh else torch.float32, device="cuda")
        sc = torch.zeros(N, Cout).cuda(); sh = torch.zeros(N, Cout).cuda()
        _lib.check(ctx.lib.nm_op_conv5_occ(ctx.handle, _lib.ptr(occ), N, G, _lib.ptr(w), _lib.ptr(b), Cout, _lib.ptr(out), 2, _lib.ptr(gam), _lib.ptr(bet),
                                           _lib.ptr(sc), _lib.ptr(sh)), "op_conv5_occ")
        torch.cuda.synchronize()
        return out, sc
    try:
        ref, rsc = run(3, 0)
        got, gsc = run(4, 1)
    finally:
        _set(ctx, 1, 0, 0)
    assert torch.equal(got.float(), ref.to(BF).float())
    assert torch.equal(gsc, rsc)


def test_elementwise_storage16(ctx):
    """apply2 (residual sums), trilinear upsample and its adjoint's input side, transposed conv (pool data gradient)."""
    from neural_marionette_amd import _lib
    g = torch.Generator().manual_seed(9)
    N, V, Cc = 2, 4096, 64
    a = torch.randn(N, V, Cc, generator=g).to(BF); b = torch.randn(N, V, Cc, generator=g).to(BF)
    sc = (torch.rand(N, Cc, generator=g) + 0.5).cuda(); sh = (torch.randn(N, Cc, generator=g) * 0.2).cuda()

    def apply2(ih, oh):
        _set(ctx, 4 if (ih or oh) else 3, ih, oh)
        aa, bb = (a.cuda(), b.cuda()) if ih else (a.float().cuda(), b.float().cuda())
        out = torch.zeros(N, V, Cc, dtype=BF if oh else torch.float32, device="cuda")
        _lib.check(ctx.lib.nm_op_apply2(ctx.handle, _lib.ptr(aa), _lib.ptr(sc), _lib.ptr(sh), 0.01, _lib.ptr(bb), None, None, 1.0, N, V, Cc, _lib.ptr(out)), "apply2")
        torch.cuda.synchronize()
        return out
    try:
        ref = apply2(0, 0)
        assert torch.equal(apply2(1, 1).float(), ref.to(BF).float())
        assert torch.equal(apply2(1, 0), ref)
        # upsample: (16^3 fp32 -> 32^3 bf16) and (bf16 -> bf16), the LDS-tiled and the per-cell kernel (C = 24)
        for Cu, D in ((64, 16), (24, 8)):
            x = torch.randn(N, D, D, D, Cu, generator=g).to(BF)
            outs = {}
            for ih, oh in ((0, 0), (0, 1), (1, 1)):
                _set(ctx, 4 if (ih or oh) else 3, ih, oh)
                xx = x.cuda() if ih else x.float().cuda()
                o = torch.zeros(N, 2 * D, 2 * D, 2 * D, Cu, dtype=BF if oh else torch.float32, device="cuda")
                _lib.check(ctx.lib.nm_op_upsample2(ctx.handle, _lib.ptr(xx), N, D, D, D, Cu, _lib.ptr(o)), "upsample2")
                torch.cuda.synchronize()
                outs[(ih, oh)] = o
            assert torch.equal(outs[(0, 1)].float(), outs[(0, 0)].to(BF).float())
            assert torch.equal(outs[(1, 1)].float(), outs[(0, 0)].to(BF).float())
    finally:
        _set(ctx, 1, 0, 0)


BWD_CASES = [
    # Cin, Cout, ks, stride, pad, size, N, up2, dgrad_channels, in16, out16
    (32, 32, 3, 1, 1, 32, 2, 0, 32, 1, 1),      # wgrad16z, conv_f16p data gradient
    (64, 64, 3, 1, 1, 16, 3, 0, 64, 1, 1),      # wgrad16z 4 tile pairs, conv_f16p2
    (32, 64, 3, 1, 1, 16, 2, 0, 32, 1, 1),      # ragged tile pairs, data gradient 64 -> 32 on conv_f16p
    (32, 64, 1, 1, 0, 16, 2, 0, 32, 1, 1),      # k1: wgrad_k1, conv_f16s k1 data gradient
    (32, 32, 2, 2, 0, 32, 2, 0, 32, 1, 1),      # pool: wgrad_kernel<0, 2>, transposed conv on the matrix cores with packed bf16 stores
    (64, 64, 2, 2, 0, 16, 8, 0, 64, 1, 0),      # pool across the storage threshold: bf16 input, fp32 dy
    (64, 32, 3, 1, 1, 16, 2, 1, 64, 1, 1),      # fused upsample, both sides bf16 (the 64 -> 32 decoder layer)
    (128, 64, 3, 1, 1, 8, 2, 1, 128, 0, 1),     # fused upsample across the threshold: fp32 coarse input, bf16 fine side
]


@pytest.mark.parametrize("case", BWD_CASES, ids=lambda c: "ci%d_co%d_k%d_s%d_d%d_up%d_io%d%d" % (c[0], c[1], c[2], c[3], c[5], c[7], c[9], c[10]))
def test_conv3d_backward_storage16(ctx, case):
    from neural_marionette_amd import _lib
    Cin, Cout, ks, stride, pad, size, N, up2, csel, ih, oh = case
    g = torch.Generator().manual_seed(hash(case) & 0xFFFF)
    dims = (size, size, size)
    us = 2 if up2 else 1
    od = tuple((us * d + 2 * pad - ks) // stride + 1 for d in dims)
    x = torch.randn(N, Cin, *dims, generator=g).to(BF).float()
    dy = (torch.randn(N, Cout, *od, generator=g) * 1e-3).to(BF).float()
    w = (torch.randn(Cout, Cin, ks, ks, ks, generator=g) / (Cin * ks ** 3) ** 0.5).cuda()
    sc = (torch.rand(N, Cin, generator=g) + 0.5).cuda(); sh = (torch.randn(N, Cin, generator=g) * 0.3).cuda()
    x32, dy32 = to_cl(x), to_cl(dy)

    def run(mode, ih_, oh_):
        _set(ctx, mode, ih_, oh_)
        xin = x32.to(BF) if ih_ else x32
        dyin = dy32.to(BF) if oh_ else dy32
        d_in = torch.zeros(N, *dims, csel, dtype=BF if ih_ else torch.float32, device="cuda")
        dw = torch.zeros(Cout, Cin, ks, ks, ks, device="cuda"); db = torch.zeros(Cout, device="cuda")
        _lib.check(ctx.lib.nm_op_conv3d_backward(ctx.handle, _lib.ptr(xin), N, *dims, Cin, _lib.ptr(sc), _lib.ptr(sh), 0.01, _lib.ptr(w), Cout, ks, stride, pad,
                                                 up2, _lib.ptr(dyin), _lib.ptr(d_in), csel, _lib.ptr(dw), _lib.ptr(db)), "op_conv3d_backward")
        torch.cuda.synchronize()
        return d_in, dw, db
    try:
        r_in, r_w, r_b = run(3, 0, 0)
        g_in, g_w, g_b = run(4, ih, oh)
    finally:
        _set(ctx, 1, 0, 0)
    assert torch.equal(g_b, r_b), "bias gradient"
    if up2:
        # the upsampled input is materialised in the fine side's storage type for the weight gradient, and the fine-grid data gradient is
        # stored before the adjoint: one bfloat16 rounding of an intermediate each
        assert relerr(g_w.cpu(), r_w.cpu()) < 2.0 ** -7
        assert relerr(g_in.float().cpu(), r_in.cpu()) < 2.0 ** -7
    else:
        assert torch.equal(g_w, r_w), "weight gradient: max diff %.3e" % (g_w - r_w).abs().max().item()
        want = r_in.to(BF).float() if ih else r_in
        assert torch.equal(g_in.float(), want), "data gradient: max diff %.3e" % (g_in.float() - want).abs().max().item()


def test_groupnorm_backward_and_first_layer_gradient_storage16(ctx):
    from neural_marionette_amd import _lib
    g = torch.Generator().manual_seed(21)
    N, V, Cc, groups = 3, 32768, 32, 2
    y = torch.randn(N, V, Cc, generator=g).to(BF); dA = (torch.randn(N, V, Cc, generator=g) * 1e-4).to(BF)
    gam = (torch.rand(Cc, generator=g) + 0.5).cuda(); bet = (torch.randn(Cc, generator=g) * 0.2).cuda()

    def gnb(h):
        _set(ctx, 4 if h else 3, h, h)
        yy, dd = (y.cuda(), dA.cuda()) if h else (y.float().cuda(), dA.float().cuda())
        dy = torch.zeros(N, V, Cc, dtype=BF if h else torch.float32, device="cuda")
        o = [torch.zeros(Cc, device="cuda") for _ in range(3)]
        _lib.check(ctx.lib.nm_op_gn_backward(ctx.handle, _lib.ptr(yy), N, V, Cc, groups, _lib.ptr(gam), _lib.ptr(bet), 0.01, _lib.ptr(dd), _lib.ptr(dy),
                                             _lib.ptr(o[0]), _lib.ptr(o[1]), _lib.ptr(o[2])), "op_gn_backward")
        torch.cuda.synchronize()
        return dy, o
    try:
        r_dy, r_o = gnb(0)
        g_dy, g_o = gnb(1)
        assert torch.equal(g_dy.float(), r_dy.to(BF).float())
        for a, b in zip(g_o, r_o):
            assert torch.equal(a, b)
        # first layer: weight gradient against cat[occ, coords] from a bfloat16 dy (matrix-core form on the non-empty bricks + dense form
        # of the coordinate channels on the frame sum)
        G, Cout = 32, 32
        occ = (torch.rand(N, G, G, G, generator=g) < 0.03).float().cuda()
        dyf = (torch.randn(N, G, G, G, Cout, generator=g) * 1e-3).to(BF)
        res = {}
        for h in (0, 1):
            _set(ctx, 4 if h else 3, 0, h)
            dd = dyf.cuda() if h else dyf.float().cuda()
            for sparse in (1, 0):
                dw = torch.zeros(Cout, 4, 5, 5, 5, device="cuda"); db = torch.zeros(Cout, device="cuda")
                _lib.check(ctx.lib.nm_op_conv5_occ_backward(ctx.handle, _lib.ptr(occ), N, G, Cout, _lib.ptr(dd), _lib.ptr(dw), _lib.ptr(db), sparse), "conv5_occ_backward")
                torch.cuda.synchronize()
                res[(h, sparse)] = (dw, db)
        for sparse in (1, 0):
            assert torch.equal(res[(1, sparse)][0], res[(0, sparse)][0]) and torch.equal(res[(1, sparse)][1], res[(0, sparse)][1])
    finally:
        _set(ctx, 1, 0, 0)


# ---- network level ------------------------------------------------------------------------------------------------------------------
def _l2(got, ref):
    num = sum(((got[k].double() - ref[k].double()) ** 2).sum().item() for k in ref)
    den = sum((ref[k].double() ** 2).sum().item() for k in ref)
    return (num / den) ** 0.5


def test_detector_gradients_bf16_storage_vs_fp64_oracle():
    """32^3, B = 2, T = 4 with the storage threshold at 16^3 (NM355_STORE16_MIN, read when a context is created): first layer, both pool
    convs, the 16^3 residual block and the whole decoder in bfloat16 - the tensor population of the 64^3 network, one level down.
    Stated bounds of the mode: loss 2e-3 relative, keypoints 5e-3, whole-gradient relative L2 distance from the fp64 oracle 4e-2 (the
    fp16-arithmetic mode with fp32 storage: 7e-3 measured, 2e-2 stated), every tensor with a non-negligible gradient within cosine 0.9."""
    o, sd, vox = _setup(seed=11)
    ref_loss, ref, ref_out = _oracle_grads(o, sd, vox, AIST, double=True)
    os.environ["NM355_STORE16_MIN"] = "4096"
    try:
        loss, got, out = _hip_grads(o, sd, vox, AIST, mode="bf16")
        loss2, got2, _ = _hip_grads(o, sd, vox, AIST, mode="bf16")
    finally:
        del os.environ["NM355_STORE16_MIN"]
    l16, g16, _ = _hip_grads(o, sd, vox, AIST, mode="f16")
    for k, v in got.items():
        assert torch.isfinite(v).all(), k
        assert torch.equal(v, got2[k]), f"{k}: two runs differ"
    assert loss == loss2
    d64, d16 = _l2(got, ref), _l2(got, g16)
    print("bf16 storage at 32^3 (threshold 16^3): loss %.6f (fp64 %.6f, f16 mode %.6f); whole-gradient L2 vs fp64 %.3e, vs the f16 mode %.3e (f16 vs fp64 %.3e)"
          % (loss, ref_loss, l16, d64, d16, _l2(g16, ref)))
    assert abs(loss - ref_loss) <= 2e-3 * max(1.0, abs(ref_loss))
    assert (out["keypoints"].detach().cpu().double() - ref_out["keypoints"].double()).abs().max().item() < 5e-3
    assert d64 < 4e-2, d64
    # per tensor: relative L2 distance to the fp64 gradient (a cosine of 0.9 would pass a tensor that is 44 % wrong).  Stated bound:
    # 0.30 for every tensor with a non-negligible gradient - the deep hourglass convs in front of a GroupNorm, whose backward removes
    # the common mode, are the worst (mode 'f16' with fp32 storage: 0.25, tests/test_train_detector_gpu.py); measured values printed
    gmax = max(r.abs().max().item() for r in ref.values())
    worst = ("", 0.0)
    for k, r in ref.items():
        if r.abs().max().item() > 1e-4 * gmax:
            gg, rr = got[k].double().flatten(), r.double().flatten()
            rel = ((gg - rr).norm() / rr.norm()).item()
            if rel > worst[1]:
                worst = (k, rel)
            assert rel < 0.30, (k, rel)
    print("bf16 storage: worst per-tensor relative L2 %.3f at %s" % (worst[1], worst[0]))


def test_materialised_upsample_layer_matches_the_fused_staging():
    """One-product training modes, the decoder's second fused-upsample layer (64 -> 32 channels): the forward materialises the
    upsampled, activated input once (conv_f16r on it, the same tensor feeds the weight gradient) instead of running the composite-
    weight kernel conv_up2c<.., 3> and rebuilding the tensor in the backward pass (NM355_UP2_MAT / NM355_F16R, read when a context is
    created).  The two forms round differently (fp16 of the upsampled activations against fp16 of the composite weights): the loss
    and the whole gradient agree within the mode's own distance from fp64, and each is as far from the fp64 oracle as the other."""
    o, sd, vox = _setup(seed=11)
    ref_loss, ref, _ = _oracle_grads(o, sd, vox, AIST, double=True)
    res = {}
    for mode in ("f16", "bf16"):
        for mat in ("1", "0"):
            os.environ["NM355_UP2_MAT"] = mat
            os.environ["NM355_STORE16_MIN"] = "4096"
            try:
                res[mode, mat] = _hip_grads(o, sd, vox, AIST, mode=mode)[:2]
            finally:
                del os.environ["NM355_UP2_MAT"]; del os.environ["NM355_STORE16_MIN"]
        (l1, g1), (l0, g0) = res[mode, "1"], res[mode, "0"]
        d, d1, d0 = _l2(g1, g0), _l2(g1, ref), _l2(g0, ref)
        print("%s: materialised vs fused upsample layer: loss %.6f / %.6f (fp64 %.6f), whole-gradient L2 between them %.3e, vs fp64 %.3e / %.3e" % (mode, l1, l0, ref_loss, d, d1, d0))
        assert abs(l1 - l0) <= 1e-3 * abs(l0)
        assert d < 4e-2 and d1 < 4e-2 and d1 < 1.5 * d0 + 5e-3


@pytest.mark.parametrize("mode", ["f16", "bf16"])
def test_twenty_step_adam_trajectory_vs_fp32_reference(mode, golden_dir):
    """BASELINE config 3's precision over an optimisation trajectory, not one step: 20 detector-mode Adam steps (lr 4e-4, AIST
    weights) in conv modes 'f16' and 'bf16' (storage threshold at 16^3: first layer, pools, the 16^3 residual block and the decoder in
    bfloat16) against the fp32 REFERENCE trained with torch.optim.Adam on the CPU (fixture G13, tools/make_golden.py::case_g13;
    rounds 4-5 trained the CPU oracle on the GPU box: 160 s).  Stated bound: every step's loss within 1e-2 relative of the
    reference's at that step; the loss must have gone down; all parameters finite."""
    import numpy as np
    from neural_marionette_amd import NeuralMarionette
    from neural_marionette_amd.train import DetectorTrainer
    g = np.load(os.path.join(golden_dir, "g13_detector_training20.npz"), allow_pickle=False)
    G, B, T, seed, steps = [int(v) for v in g["meta"]]
    ref = [float(v) for v in g["losses"]]
    o, sd, vox = _setup(G=G, B=B, T=T, seed=seed)
    if mode == "bf16":
        os.environ["NM355_STORE16_MIN"] = "4096"
    try:
        net = NeuralMarionette(o)
        net.load_state_dict(sd)
        net = net.cuda().train()
        net.anneal(1)
        net.set_conv_mode(mode)
        tr = DetectorTrainer(net, lr=4e-4)
        losses = [tr.step(vox.cuda())["loss"] for _ in range(steps)]
        torch.cuda.synchronize()
    finally:
        os.environ.pop("NM355_STORE16_MIN", None)
    rel = [abs(a - b) / abs(b) for a, b in zip(losses, ref)]
    print("%s 20-step trajectory: loss %.4f -> %.4f (reference %.4f -> %.4f), worst relative deviation %.2e at step %d"
          % (mode, losses[0], losses[-1], ref[0], ref[-1], max(rel), rel.index(max(rel))))
    assert max(rel) < 1e-2, list(zip(losses, ref))
    assert losses[-1] < losses[0]
    assert all(torch.isfinite(p).all() for p in net.parameters())


def test_config3_in_its_named_precision_at_the_bench_shape():
    """BASELINE config 3's per-GPU shard in bfloat16 storage: 64^3, T = 16, B = 4.  No oracle at this size: (1) batch additivity (the
    4-clip gradient is the mean of the single-clip gradients: every loss is a mean over clips, GroupNorm is per frame) within the
    mode's accuracy, (2) two runs bit-identical, (3) whole-gradient L2 distance to the fp32-storage 'f16' mode at the same size,
    (4) the training arena of the context (nm_ctx_memory) at most 13 GB against fp32 storage's 20.5 GB (10.6 GB of 16-bit tape, and the
    decoder's upsampled 64^3 x 64-channel input - 2.1 GB - which the forward materialises once for its conv and its weight gradient)."""
    from neural_marionette_amd import _lib
    o, sd, vox = _setup(G=64, B=4, T=16, seed=91)
    net = None

    def grads(v, keep=False):
        nonlocal net
        n = __import__("neural_marionette_amd").NeuralMarionette(o)
        n.load_state_dict(sd); n = n.cuda().train(); n.set_conv_mode("bf16"); n.anneal(1)
        acts = {"detector": True, "learner": False}
        n.control_active(acts); n.zero_grad()
        outp = n(v.cuda(), acts)
        loss = sum(w * outp[k] for k, w in AIST.items())
        loss.backward()
        torch.cuda.synchronize()
        gr = {"kypt_detector." + nm: p.grad.detach().cpu() for nm, p in n.kypt_detector.named_parameters()}
        if keep:
            net = n
        return float(loss.detach()), gr
    loss, g_all = grads(vox, keep=True)
    mem = (C.c_size_t * 4)()
    _lib.check(net._engine.ctx.lib.nm_ctx_memory(net._engine.ctx.handle, mem), "ctx_memory")
    print("bf16 storage, 64^3 B=4 T=16: training arena %.2f GB, weight-gradient side block %.2f GB" % (mem[1] / 1e9, mem[2] / 1e9))
    assert mem[1] <= 13e9, mem[1]
    net = None
    acc = {k: torch.zeros_like(v, dtype=torch.float64) for k, v in g_all.items()}
    for b in range(4):
        _, g_b = grads(vox[b:b + 1].contiguous())
        for k in acc:
            acc[k] += g_b[k].double() / 4
    dadd = _l2(g_all, acc)
    _, g_again = grads(vox)
    for k, v in g_all.items():
        assert torch.isfinite(v).all(), k
        assert torch.equal(v, g_again[k]), f"{k}: gradients differ between two runs"
    l16, g16, _ = _hip_grads(o, sd, vox, AIST, mode="f16")
    d16 = _l2(g_all, g16)
    print("bf16 storage, 64^3 B=4 T=16: loss %.6f (f16 mode %.6f), batch additivity L2 %.3e, whole-gradient L2 vs the f16 mode %.3e" % (loss, l16, dadd, d16))
    assert abs(loss - l16) <= 2e-3 * abs(l16)
    assert dadd < 4e-2 and d16 < 4e-2
