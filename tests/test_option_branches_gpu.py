"""Option branches away from the shipped configurations (round-5 verdict, "missing" item 1) on the HIP path.

options.gaussian_cat_type 'max' / 'sum' (kypt_detector.py:396-401): the K Gaussian channels of the voxel decoder's combined representation
all carry the maximum / the clipped sum over the K maps - the materialised combined tensor with the reduced maps, its adjoint through the
arg-max / the clip; second half of this file.

options.vol_fit_type 'gaussian' (kypt_detector_utils.py:154-169) as the reference computes it - two-dimensional Gaussian maps whose
"intensity" is the keypoint's third coordinate, the mask broadcast along the frame's first axis and across the batch
(oracle.nm_oracle.loss_volume_gaussian): last part of this file.

options.fixed_sigma = 0 (kypt_detector.py:258-260, 303-306): a trainable (K,) parameter, sigma_k = sigmoid(p_k) * 2 gaussian_sigma in the
detector's Gaussian maps (decode_from_dyna keeps the fixed width); its gradient; also together with gaussian_cat_type 'sum'.

options.affinity_ver 0 / 1 / 2 (kypt_detector.py:57-68,173-189): (N, K, K) affinity parameters; 0 = row softmax,
1 = softplus Gram matrix with a zero diagonal, rows divided by (row sum + 1e-6), 2 = softplus, zero diagonal, row softmax.  No shipped
configuration selects them (every dataset block and the pretrained options use 3); they exist so that a user flag away from the
defaults runs instead of raising.  tests/test_oracle_vs_reference.py pins the oracle to the reference for these versions bit for bit;
here: the full forward (both paths), the affinity itself, the tree the VRNN is built on, and the gradient of the AIST-weighted training
loss w.r.t. every detector parameter - the affinity parameters' adjoint is the only version-specific part - against the fp64 oracle.
Tolerances as in tests/test_keypoint_counts_gpu.py."""
import numpy as np
import pytest
import torch

from neural_marionette_amd import NeuralMarionette, HotPathOptions, synth
from neural_marionette_amd.spec import DETECTOR_LOSS_KEYS
from neural_marionette_amd.train import DETECTOR_LOSS_WEIGHTS as AIST
from oracle import nm_oracle as O

pytestmark = pytest.mark.gpu
ACTS = {"detector": True, "learner": True}


def _err(a, b):
    return (a.detach().cpu().double() - b.detach().cpu().double()).abs().max().item()


def _setup(ver, seed, G=32, B=2, T=4, cat="none", vol="chamfer", fixed_sigma=1):
    o = HotPathOptions(grid_size=G, affinity_ver=ver, gaussian_cat_type=cat, vol_fit_type=vol, fixed_sigma=fixed_sigma)
    sd = synth.make_state_dict(o, seed=seed, variant="peaky")
    assert tuple(sd["kypt_detector.affinity_params"].shape) == ((2, 24, 24) if ver < 3 else (2, 24, 23))
    vox = synth.figure_clip(B, T, G, seed=seed + 2)
    eps = synth.make_eps((T, 10, B, o.nlatent_kypt), seed=seed + 3)
    return o, sd, vox, eps


def _net(o, sd, train=False):
    net = NeuralMarionette(o)
    net.load_state_dict(sd)
    net = net.cuda()
    net = net.train() if train else net.eval()
    net.anneal(1)
    return net


@pytest.mark.parametrize("path", ["train_fwd", "inference"])
@pytest.mark.parametrize("ver", [0, 1, 2])
def test_forward_parity_affinity_versions(ver, path):
    o, sd, vox, eps = _setup(ver, 410 + ver)
    with torch.no_grad():
        ref = O.nm_forward(sd, o, vox, eps)
    net = _net(o, sd)
    assert tuple(net.kypt_detector.affinity_params.shape) == (2, 24, 24)

    def run():
        if path == "inference":
            with torch.no_grad():
                return net(vox.cuda(), ACTS, eps=eps.cuda())
        return net(vox.cuda(), ACTS, eps=eps.cuda())
    run()
    out = run()
    torch.cuda.synchronize()
    assert _err(net.kypt_detector.get_affinity(), ref["affinity"]) < 1e-6
    assert _err(out["affinity"], ref["affinity"]) < 1e-6
    assert _err(out["keypoints"], ref["keypoints"]) < 1e-4
    assert np.array_equal(net.dyna_module.parents.cpu().numpy(), ref["parents"])
    assert np.array_equal(net.dyna_module.priority.indices.cpu().numpy(), ref["order"])
    assert np.array_equal(out["best_idx"].cpu().numpy(), ref["best_idx"].numpy().astype(np.int32))
    for k in ("z_kypts", "h_kypts", "kypt_recon", "R"):
        assert _err(out[k], ref[k]) < 1e-4, k
    for k in DETECTOR_LOSS_KEYS + ("kl_kypt", "kypt_recon_loss"):
        r = float(ref[k])
        assert abs(float(out[k]) - r) <= 2e-5 * max(1.0, abs(r)), (k, float(out[k]), r)


@pytest.mark.parametrize("ver", [0, 1, 2])
def test_detector_gradients_affinity_versions(ver):
    o, sd, vox, _ = _setup(ver, 420 + ver, B=1, T=3)
    sd64, vox64 = {k: v.double() for k, v in sd.items()}, vox.double()
    names = [k for k in sd64 if k.startswith("kypt_detector.")]
    leaf = {k: sd64[k].clone().requires_grad_(True) for k in names}
    sd2 = dict(sd64); sd2.update(leaf)
    ro = O.detector_forward(sd2, o, vox64, affinity_on=True)
    ref_loss = sum(w * ro[k] for k, w in AIST.items())
    grads = torch.autograd.grad(ref_loss, [leaf[k] for k in names], allow_unused=True)
    ref = {k: (g if g is not None else torch.zeros_like(leaf[k])) for k, g in zip(names, grads)}
    net = _net(o, sd, train=True)
    acts = {"detector": True, "learner": False}
    net.control_active(acts)
    net.zero_grad()
    out = net(vox.cuda(), acts)
    loss = sum(w * out[k] for k, w in AIST.items())
    loss.backward()
    torch.cuda.synchronize()
    assert abs(float(loss) - float(ref_loss)) <= 2e-5 * max(1.0, abs(float(ref_loss)))
    got = {"kypt_detector." + n: p.grad for n, p in net.kypt_detector.named_parameters()}
    ga, ra = got["kypt_detector.affinity_params"].cpu().double(), ref["kypt_detector.affinity_params"]
    assert tuple(ga.shape) == (2, 24, 24) and ra.abs().max() > 0
    ea = (ga - ra).abs().max().item() / ra.abs().max().item()
    print("affinity_ver %d: d loss / d affinity_params relative error %.2e (largest entry %.3e)" % (ver, ea, ra.abs().max().item()))
    assert ea < 2e-3, ea
    gmax = max(r.abs().max().item() for r in ref.values())
    bad = []
    for k, r in ref.items():
        g = got[k]
        assert g is not None and tuple(g.shape) == tuple(r.shape) and torch.isfinite(g).all(), k
        e = (g.cpu().double() - r).abs().max().item() / max(r.abs().max().item(), 1e-6 * gmax, 1e-30)
        if e >= 2e-3:
            bad.append((k, e))
    assert not bad, bad[:8]


def test_affinity_version_four_is_rejected():
    with pytest.raises(NotImplementedError):
        NeuralMarionette(HotPathOptions(grid_size=32, affinity_ver=4))


# ---- gaussian_cat_type 'max' / 'sum' -----------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("path", ["train_fwd", "inference"])
@pytest.mark.parametrize("cat", ["max", "sum"])
def test_forward_parity_gaussian_cat_types(cat, path):
    o, sd, vox, eps = _setup(3, 430 + len(cat), cat=cat)
    with torch.no_grad():
        ref = O.nm_forward(sd, o, vox, eps)
        plain = O.nm_forward(sd, HotPathOptions(grid_size=32), vox, eps)
    assert (ref["recon"] - plain["recon"]).abs().max() > 1e-2          # the option changes the decoder's input for real
    net = _net(o, sd)

    def run():
        if path == "inference":
            with torch.no_grad():
                return net(vox.cuda(), ACTS, eps=eps.cuda())
        return net(vox.cuda(), ACTS, eps=eps.cuda())
    run()
    out = run()
    torch.cuda.synchronize()
    assert _err(out["keypoints"], ref["keypoints"]) < 1e-4
    e_rec = _err(out["recon"], ref["recon"])
    print("gaussian_cat_type %s %s: recon %.3e" % (cat, path, e_rec))
    assert e_rec < 1e-4
    rr = ref["recon"]
    differ = ((out["recon"].cpu() >= 0.5) != (rr >= 0.5)) & ((rr - 0.5).abs() > 1e-4)
    assert int(differ.sum()) == 0
    for k in ("z_kypts", "h_kypts", "kypt_recon", "R"):
        assert _err(out[k], ref[k]) < 1e-4, k
    for k in DETECTOR_LOSS_KEYS + ("kl_kypt", "kypt_recon_loss"):
        r = float(ref[k])
        assert abs(float(out[k].detach()) - r) <= 2e-5 * max(1.0, abs(r)), (k, float(out[k].detach()), r)
    # decode_from_dyna on the oracle's keypoints
    with torch.no_grad():
        gen = net.kypt_detector.decode_from_dyna(ref["keypoints"][:, 1:3].cuda(), ref["first_feature"].cuda(), vox[:, 0].cuda())["gen"]
        want = O.decode_from_keypoints(sd, o, ref["keypoints"][:, 1:3], ref["first_feature"], vox[:, 0])
    assert _err(gen, want) < 1e-4
    assert int((((gen.cpu() >= 0.5) != (want >= 0.5)) & ((want - 0.5).abs() > 1e-4)).sum()) == 0


@pytest.mark.parametrize("cat", ["max", "sum"])
def test_detector_gradients_gaussian_cat_types(cat):
    """'sum': the bound of every gradient test here (2e-3 of each tensor's largest entry against fp64 autograd).  'max': the arg-max over
    the K maps at every decoder voxel is a selection - where two maps are within fp32 rounding of each other an fp32 evaluation picks the
    other one, so the REFERENCE's own fp32 autograd deviates from fp64 by more than that bound (measured here, printed); the HIP path is
    held to 2e-3 plus twice that deviation, tensor by tensor the same measure."""
    o, sd, vox, _ = _setup(3, 440 + len(cat), B=1, T=3, cat=cat)

    def oracle_grads(dt):
        sdd, voxd = {k: v.to(dt) for k, v in sd.items()}, vox.to(dt)
        names = [k for k in sdd if k.startswith("kypt_detector.")]
        leaf = {k: sdd[k].clone().requires_grad_(True) for k in names}
        sd2 = dict(sdd); sd2.update(leaf)
        ro = O.detector_forward(sd2, o, voxd, affinity_on=True)
        loss = sum(w * ro[k] for k, w in AIST.items())
        grads = torch.autograd.grad(loss, [leaf[k] for k in names], allow_unused=True)
        return loss.detach(), {k: (g if g is not None else torch.zeros_like(leaf[k])).double() for k, g in zip(names, grads)}
    ref_loss, ref = oracle_grads(torch.float64)
    slack = 0.0
    if cat == "max":
        _, g32 = oracle_grads(torch.float32)
        gm = max(r.abs().max().item() for r in ref.values())
        slack = max((g32[k] - ref[k]).abs().max().item() / max(ref[k].abs().max().item(), 1e-6 * gm, 1e-30) for k in ref)
        print("gaussian_cat_type max: the oracle's own fp32 autograd deviates from fp64 by %.2e (worst tensor, relative)" % slack)
    net = _net(o, sd, train=True)
    acts = {"detector": True, "learner": False}
    net.control_active(acts)
    net.zero_grad()
    out = net(vox.cuda(), acts)
    loss = sum(w * out[k] for k, w in AIST.items())
    loss.backward()
    torch.cuda.synchronize()
    assert abs(float(loss) - float(ref_loss)) <= 2e-5 * max(1.0, abs(float(ref_loss)))
    got = {"kypt_detector." + n: p.grad for n, p in net.kypt_detector.named_parameters()}
    gmax = max(r.abs().max().item() for r in ref.values())
    worst, bad = ("", 0.0), []
    for k, r in ref.items():
        g = got[k]
        assert g is not None and tuple(g.shape) == tuple(r.shape) and torch.isfinite(g).all(), k
        e = (g.cpu().double() - r).abs().max().item() / max(r.abs().max().item(), 1e-6 * gmax, 1e-30)
        if e > worst[1]:
            worst = (k, e)
        if e >= 2e-3 + 2.0 * slack:
            bad.append((k, e))
    print("gaussian_cat_type %s: worst relative gradient error %.2e at %s" % (cat, worst[1], worst[0]))
    assert not bad, bad[:8]


# ---- vol_fit_type 'gaussian' -------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("path", ["train_fwd", "inference"])
@pytest.mark.parametrize("B", [1, 3])
def test_forward_parity_vol_fit_gaussian(B, path):
    o, sd, vox, eps = _setup(3, 450 + B, B=B, vol="gaussian")
    with torch.no_grad():
        ref = O.nm_forward(sd, o, vox, eps)
    net = _net(o, sd)

    def run():
        if path == "inference":
            with torch.no_grad():
                return net(vox.cuda(), ACTS, eps=eps.cuda())
        return net(vox.cuda(), ACTS, eps=eps.cuda())
    run()
    out = run()
    torch.cuda.synchronize()
    assert _err(out["keypoints"], ref["keypoints"]) < 1e-4
    r = float(ref["vol_fit_reg"])
    print("vol_fit_type gaussian B=%d %s: vol_fit_reg %.6f (oracle %.6f)" % (B, path, float(out["vol_fit_reg"].detach()), r))
    assert r > 0.05 * B                                     # (the cross-batch sum: grows with B)
    for k in DETECTOR_LOSS_KEYS + ("kl_kypt", "kypt_recon_loss"):
        rk = float(ref[k])
        assert abs(float(out[k].detach()) - rk) <= 2e-5 * max(1.0, abs(rk)), (k, float(out[k].detach()), rk)


@pytest.mark.parametrize("B", [1, 2])
def test_detector_gradients_vol_fit_gaussian(B):
    """The loss reaches the keypoints through the arg-max over the K two-dimensional maps (a selection, as for gaussian_cat_type 'max':
    the oracle's own fp32 deviation from fp64 is measured and allowed twice)."""
    o, sd, vox, _ = _setup(3, 460 + B, B=B, T=3, vol="gaussian")

    def oracle_grads(dt):
        sdd, voxd = {k: v.to(dt) for k, v in sd.items()}, vox.to(dt)
        names = [k for k in sdd if k.startswith("kypt_detector.")]
        leaf = {k: sdd[k].clone().requires_grad_(True) for k in names}
        sd2 = dict(sdd); sd2.update(leaf)
        ro = O.detector_forward(sd2, o, voxd, affinity_on=True)
        loss = sum(w * ro[k] for k, w in AIST.items())
        grads = torch.autograd.grad(loss, [leaf[k] for k in names], allow_unused=True)
        return loss.detach(), {k: (g if g is not None else torch.zeros_like(leaf[k])).double() for k, g in zip(names, grads)}
    ref_loss, ref = oracle_grads(torch.float64)
    _, g32 = oracle_grads(torch.float32)
    gmax = max(r.abs().max().item() for r in ref.values())
    slack = max((g32[k] - ref[k]).abs().max().item() / max(ref[k].abs().max().item(), 1e-6 * gmax, 1e-30) for k in ref)
    net = _net(o, sd, train=True)
    acts = {"detector": True, "learner": False}
    net.control_active(acts)
    net.zero_grad()
    out = net(vox.cuda(), acts)
    loss = sum(w * out[k] for k, w in AIST.items())
    loss.backward()
    torch.cuda.synchronize()
    assert abs(float(loss.detach()) - float(ref_loss)) <= 2e-5 * max(1.0, abs(float(ref_loss)))
    got = {"kypt_detector." + n: p.grad for n, p in net.kypt_detector.named_parameters()}
    worst, bad = ("", 0.0), []
    for k, r in ref.items():
        g = got[k]
        assert g is not None and tuple(g.shape) == tuple(r.shape) and torch.isfinite(g).all(), k
        e = (g.cpu().double() - r).abs().max().item() / max(r.abs().max().item(), 1e-6 * gmax, 1e-30)
        if e > worst[1]:
            worst = (k, e)
        if e >= 2e-3 + 2.0 * slack:
            bad.append((k, e))
    print("vol_fit_type gaussian B=%d: worst relative gradient error %.2e at %s (the oracle's own fp32 deviation: %.2e)" % (B, worst[1], worst[0], slack))
    assert not bad, bad[:8]


# ---- fixed_sigma = 0 ------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("path", ["train_fwd", "inference"])
def test_forward_parity_learnable_sigmas(path):
    o, sd, vox, eps = _setup(3, 470, fixed_sigma=0)
    sd["kypt_detector.vox_to_kypt.sigmas"] = torch.randn(24, generator=torch.Generator().manual_seed(471))
    with torch.no_grad():
        ref = O.nm_forward(sd, o, vox, eps)
        plain = O.nm_forward({k: v for k, v in sd.items() if not k.endswith("vox_to_kypt.sigmas")}, HotPathOptions(grid_size=32), vox, eps)
    assert (ref["recon"] - plain["recon"]).abs().max() > 1e-2
    net = _net(o, sd)
    assert [n for n, _ in net.named_parameters()][1] == "kypt_detector.vox_to_kypt.sigmas"

    def run():
        if path == "inference":
            with torch.no_grad():
                return net(vox.cuda(), ACTS, eps=eps.cuda())
        return net(vox.cuda(), ACTS, eps=eps.cuda())
    run()
    out = run()
    torch.cuda.synchronize()
    assert _err(out["keypoints"], ref["keypoints"]) < 1e-4
    assert _err(out["recon"], ref["recon"]) < 1e-4
    for k in ("z_kypts", "h_kypts", "kypt_recon", "R"):
        assert _err(out[k], ref[k]) < 1e-4, k
    for k in DETECTOR_LOSS_KEYS + ("kl_kypt", "kypt_recon_loss"):
        r = float(ref[k])
        assert abs(float(out[k].detach()) - r) <= 2e-5 * max(1.0, abs(r)), (k, float(out[k].detach()), r)
    with torch.no_grad():                                   # decode_from_dyna: the FIXED width (kypt_detector.py:226)
        gen = net.kypt_detector.decode_from_dyna(ref["keypoints"][:, 1:3].cuda(), ref["first_feature"].cuda(), vox[:, 0].cuda())["gen"]
        want = O.decode_from_keypoints(sd, o, ref["keypoints"][:, 1:3], ref["first_feature"], vox[:, 0])
    assert _err(gen, want) < 1e-4


@pytest.mark.parametrize("cat", ["none", "sum"])
def test_detector_gradients_learnable_sigmas(cat):
    # (weight seeds on which the oracle's own fp32 autograd stays within 5e-5 of fp64: about half of the random weight sets of this tiny
    #  B = 1, T = 3 problem have a near-tie in one of the losses' selections and deviate by 1e-3 ... 4e-2 in fp32 - scanned on the CPU)
    o, sd, vox, _ = _setup(3, {"none": 491, "sum": 496}[cat], B=1, T=3, cat=cat, fixed_sigma=0)
    sd["kypt_detector.vox_to_kypt.sigmas"] = torch.randn(24, generator=torch.Generator().manual_seed(481))
    sd64, vox64 = {k: v.double() for k, v in sd.items()}, vox.double()
    names = [k for k in sd64 if k.startswith("kypt_detector.")]
    leaf = {k: sd64[k].clone().requires_grad_(True) for k in names}
    sd2 = dict(sd64); sd2.update(leaf)
    ro = O.detector_forward(sd2, o, vox64, affinity_on=True)
    ref_loss = sum(w * ro[k] for k, w in AIST.items())
    grads = torch.autograd.grad(ref_loss, [leaf[k] for k in names], allow_unused=True)
    ref = {k: (g if g is not None else torch.zeros_like(leaf[k])) for k, g in zip(names, grads)}
    net = _net(o, sd, train=True)
    acts = {"detector": True, "learner": False}
    net.control_active(acts)
    net.zero_grad()
    out = net(vox.cuda(), acts)
    loss = sum(w * out[k] for k, w in AIST.items())
    loss.backward()
    torch.cuda.synchronize()
    assert abs(float(loss.detach()) - float(ref_loss)) <= 2e-5 * max(1.0, abs(float(ref_loss)))
    got = {"kypt_detector." + n: p.grad for n, p in net.kypt_detector.named_parameters()}
    gs, rs = got["kypt_detector.vox_to_kypt.sigmas"].cpu().double(), ref["kypt_detector.vox_to_kypt.sigmas"]
    assert tuple(gs.shape) == (24,) and rs.abs().max() > 0
    es = (gs - rs).abs().max().item() / rs.abs().max().item()
    print("fixed_sigma 0 (%s): d loss / d sigmas relative error %.2e (largest entry %.3e)" % (cat, es, rs.abs().max().item()))
    assert es < 2e-3
    gmax = max(r.abs().max().item() for r in ref.values())
    bad = []
    for k, r in ref.items():
        g = got[k]
        assert g is not None and tuple(g.shape) == tuple(r.shape) and torch.isfinite(g).all(), k
        e = (g.cpu().double() - r).abs().max().item() / max(r.abs().max().item(), 1e-6 * gmax, 1e-30)
        if e >= 2e-3:
            bad.append((k, e))
    assert not bad, bad[:8]


def test_learnable_sigmas_with_the_gaussian_volume_loss_are_rejected():
    with pytest.raises(NotImplementedError):
        NeuralMarionette(HotPathOptions(grid_size=32, fixed_sigma=0, vol_fit_type="gaussian"))
