"""Pins the oracle directly against the reference implementation, imported from
/root/reference.  That tree exists only in the build container, so these tests skip
on the GPU box (the committed fixtures of tests/golden/ carry the same evidence there)."""
import os
import pickle
import sys

import numpy as np
import pytest
import torch

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "model")), reason="reference tree not present")

from neural_marionette_amd import synth
from neural_marionette_amd.spec import HotPathOptions, DETECTOR_LOSS_KEYS, param_spec
from oracle import nm_oracle as O


@pytest.fixture(scope="module")
def ref_modules():
    sys.path.insert(0, REF)
    try:
        from model.neural_marionette import NeuralMarionette
        from utils.dyna_utils import process_affinity_glob
        yield NeuralMarionette, process_affinity_glob
    finally:
        sys.path.remove(REF)


def _opt(G):
    opt = pickle.load(open(os.path.join(REF, "pretrained/aist/opt.pickle"), "rb"))
    opt.grid_size = G
    return opt


def test_state_dict_layout(ref_modules):
    NeuralMarionette, _ = ref_modules
    net = NeuralMarionette(_opt(64))
    ref = [(k, tuple(v.shape)) for k, v in net.state_dict().items()]
    assert ref == [(k, tuple(s)) for k, s in param_spec(HotPathOptions())]


def test_trees_many_seeds(ref_modules):
    _, process_affinity_glob = ref_modules
    rng = np.random.default_rng(7)
    for trial in range(80):
        p = rng.standard_normal((2, 24, 23)) * [0.1, 1.0, 4.0][trial % 3]
        if trial % 5 == 0:
            p = np.round(p * 2) / 2
        aff = O.affinity_v3(torch.from_numpy(p).float())
        A, pri, par = process_affinity_glob(aff)
        A2, order, vals, parents = O.build_tree(aff)
        assert np.array_equal(par.numpy(), parents), trial
        assert np.array_equal(A.numpy(), A2), trial
        assert np.array_equal(pri.values.numpy(), vals), trial
        assert int(pri.indices[0]) == int(order[0])


@pytest.mark.parametrize("G", [64, 40])
def test_shell_initialisation_matches_reference(ref_modules, G):
    """train.py:233,262-268 of the reference: ``NeuralMarionette(opt)`` then ``network.apply(weights_init)``.  The shells
    are built from torch layer subclasses in the reference's creation order under the reference's class names, so the
    same seed gives a bit-identical state_dict before AND after the reference's own ``weights_init``, consumes the same
    amount of the RNG stream, and leaves the same requires_grad flags.  Also stand-alone KyptDetector / HSVRNNBVH."""
    NeuralMarionette, _ = ref_modules
    from model.kypt_detector import KyptDetector as RefDetector
    from model.hsvrnn_bvh import HSVRNNBVH as RefLearner
    from utils.train_utils import weights_init
    import neural_marionette_amd as nm
    from neural_marionette_amd.modules import KyptDetector, HSVRNNBVH
    opt = _opt(G)

    def same(a, b):
        sa, sb = a.state_dict(), b.state_dict()
        assert list(sa) == list(sb)
        bad = [k for k in sa if not torch.equal(sa[k], sb[k])]
        assert not bad, bad[:5]
        assert [(n, p.requires_grad) for n, p in a.named_parameters()] == [(n, p.requires_grad) for n, p in b.named_parameters()]

    torch.manual_seed(7); ref = NeuralMarionette(opt); tail_ref = torch.rand(4)
    torch.manual_seed(7); net = nm.NeuralMarionette(opt); tail = torch.rand(4)
    same(ref, net)
    assert torch.equal(tail, tail_ref)
    assert not any(bool((v == 0).all()) for k, v in net.state_dict().items() if k.endswith("weight"))
    torch.manual_seed(8); ref.apply(weights_init); tail_ref = torch.rand(4)
    torch.manual_seed(8); net.apply(weights_init); tail = torch.rand(4)
    same(ref, net)
    assert torch.equal(tail, tail_ref)
    w = net.state_dict()
    assert 5e-4 < float(w["kypt_detector.vox_to_kypt.extract_features.2.res_branch.0.weight"].std()) < 2e-3
    assert 1e-2 < float(w["kypt_detector.kypt_to_vox.decode_voxel_from_combined_representation.4.weight"].std()) < 4e-2
    # the module walk weights_init sees: same class names in the same order
    strip = lambda m: [type(x).__name__ for x in m.modules() if any(True for _ in x.parameters(recurse=False))]
    assert strip(ref) == strip(net)
    torch.manual_seed(3); a = RefDetector(opt); torch.manual_seed(3); b = KyptDetector(opt)
    same(a, b)
    torch.manual_seed(3); a = RefLearner(opt); torch.manual_seed(3); b = HSVRNNBVH(opt)
    same(a, b)


@pytest.mark.parametrize("variant,clip", [("peaky", "figure"), ("default", "bernoulli"), ("winit", "figure")])
def test_full_forward_32(ref_modules, variant, clip):
    NeuralMarionette, _ = ref_modules
    import torch.distributions.normal as tdn
    G, B, T = 32, 2, 5
    opt = _opt(G)
    o = HotPathOptions.from_any(opt)
    sd = synth.make_state_dict(o, seed=17, variant=variant)
    net = NeuralMarionette(opt).eval()
    net.load_state_dict(sd)
    net.anneal(1)
    vox = synth.figure_clip(B, T, G, seed=2) if clip == "figure" else synth.bernoulli_clip(B, T, G, seed=2)
    eps = synth.make_eps((T, 10, B, 128), seed=3)
    it = iter(eps)
    old = tdn._standard_normal
    tdn._standard_normal = lambda shape, dtype, device: next(it).clone()
    try:
        with torch.no_grad():
            ref = net(vox, {"detector": True, "learner": True})
    finally:
        tdn._standard_normal = old
    with torch.no_grad():
        mine = O.nm_forward(sd, o, vox, eps)
    for k in ("recon", "keypoints", "heatmaps", "affinity", "first_feature", "kypt_recon", "R", "z_kypts", "h_kypts"):
        assert torch.equal(ref[k], mine[k]), k
    for k in DETECTOR_LOSS_KEYS + ("kl_kypt", "kypt_recon_loss"):
        assert float(ref[k]) == float(mine[k]), k
    assert np.array_equal(net.dyna_module.parents.numpy(), mine["parents"])


@pytest.mark.parametrize("ver", [0, 1, 2])
def test_full_forward_32_other_affinity_versions(ref_modules, ver):
    """options.affinity_ver 0 / 1 / 2 (kypt_detector.py:57-68,173-189; no shipped configuration selects them): (N,K,K) affinity
    parameters, the reference's forward against the oracle's bit for bit - layout, affinity, every loss, the tree the VRNN is built on."""
    NeuralMarionette, _ = ref_modules
    import torch.distributions.normal as tdn
    G, B, T = 32, 2, 5
    opt = _opt(G)
    opt.affinity_ver = ver
    o = HotPathOptions.from_any(opt)
    assert o.affinity_ver == ver
    net = NeuralMarionette(opt).eval()
    assert [(k, tuple(v.shape)) for k, v in net.state_dict().items()] == [(k, tuple(sh)) for k, sh in param_spec(o)]
    sd = synth.make_state_dict(o, seed=23 + ver, variant="peaky")
    assert tuple(sd["kypt_detector.affinity_params"].shape) == (2, 24, 24)
    net.load_state_dict(sd)
    net.anneal(1)
    vox = synth.figure_clip(B, T, G, seed=2)
    eps = synth.make_eps((T, 10, B, 128), seed=3)
    it = iter(eps)
    old = tdn._standard_normal
    tdn._standard_normal = lambda shape, dtype, device: next(it).clone()
    try:
        with torch.no_grad():
            ref = net(vox, {"detector": True, "learner": True})
    finally:
        tdn._standard_normal = old
    with torch.no_grad():
        mine = O.nm_forward(sd, o, vox, eps)
    assert torch.equal(net.kypt_detector.get_affinity(), O.affinity(sd["kypt_detector.affinity_params"], ver))
    for k in ("recon", "keypoints", "affinity", "kypt_recon", "R", "z_kypts", "h_kypts"):
        assert torch.equal(ref[k], mine[k]), k
    for k in DETECTOR_LOSS_KEYS + ("kl_kypt", "kypt_recon_loss"):
        assert float(ref[k]) == float(mine[k]), k
    assert np.array_equal(net.dyna_module.parents.numpy(), mine["parents"])


@pytest.mark.parametrize("cat", ["max", "sum"])
def test_full_forward_32_other_gaussian_cat_types(ref_modules, cat):
    """options.gaussian_cat_type 'max' / 'sum' (kypt_detector.py:396-401; no shipped configuration selects them): the K Gaussian channels
    of the decoder's combined representation all carry the maximum / the clipped sum over the K maps.  Reference forward against the
    oracle's, bit for bit, incl. decode_from_dyna."""
    NeuralMarionette, _ = ref_modules
    import torch.distributions.normal as tdn
    G, B, T = 32, 2, 5
    opt = _opt(G)
    opt.gaussian_cat_type = cat
    o = HotPathOptions.from_any(opt)
    assert o.gaussian_cat_type == cat
    net = NeuralMarionette(opt).eval()
    assert [(k, tuple(v.shape)) for k, v in net.state_dict().items()] == [(k, tuple(sh)) for k, sh in param_spec(o)]
    sd = synth.make_state_dict(o, seed=31, variant="peaky")
    net.load_state_dict(sd)
    net.anneal(1)
    vox = synth.figure_clip(B, T, G, seed=2)
    eps = synth.make_eps((T, 10, B, 128), seed=3)
    it = iter(eps)
    old = tdn._standard_normal
    tdn._standard_normal = lambda shape, dtype, device: next(it).clone()
    try:
        with torch.no_grad():
            ref = net(vox, {"detector": True, "learner": True})
            gen = net.kypt_detector.decode_from_dyna(ref["keypoints"][:, 1:4], ref["first_feature"], vox[:, 0])
    finally:
        tdn._standard_normal = old
    with torch.no_grad():
        mine = O.nm_forward(sd, o, vox, eps)
        mine_gen = O.decode_from_keypoints(sd, o, ref["keypoints"][:, 1:4], ref["first_feature"], vox[:, 0])
    for k in ("recon", "keypoints", "kypt_recon", "R", "z_kypts", "h_kypts"):
        assert torch.equal(ref[k], mine[k]), k
    assert torch.equal(gen["gen"], mine_gen)
    for k in DETECTOR_LOSS_KEYS + ("kl_kypt", "kypt_recon_loss"):
        assert float(ref[k]) == float(mine[k]), k


def test_full_forward_32_vol_fit_gaussian(ref_modules):
    """options.vol_fit_type 'gaussian' (kypt_detector_utils.py:154-169; no shipped configuration selects it).  The reference's form
    reads a keypoint's third coordinate as the Gaussian's intensity and broadcasts the (B,1,G,G) mask across the batch
    (oracle.nm_oracle.loss_volume_gaussian spells it out): restated as it computes, bit for bit, for B = 1, 2, 3 on the loss alone and
    through the full forward."""
    NeuralMarionette, _ = ref_modules
    sys.path.insert(0, REF)
    try:
        from utils.kypt_detector_utils import get_volume_fitting_loss
    finally:
        sys.path.remove(REF)
    gen = torch.Generator().manual_seed(5)
    for B in (1, 2, 3):
        seq = (torch.rand(B, 3, 1, 16, 16, 16, generator=gen) < 0.1).float()
        kp = torch.rand(B, 3, 7, 4, generator=gen) * 2 - 1
        assert torch.equal(get_volume_fitting_loss(seq, kp, [1.5] * 7, "gaussian"), O.loss_volume_gaussian(seq, kp, 1.5))
    import torch.distributions.normal as tdn
    G, B, T = 32, 2, 5
    opt = _opt(G)
    opt.vol_fit_type = "gaussian"
    o = HotPathOptions.from_any(opt)
    sd = synth.make_state_dict(o, seed=37, variant="peaky")
    net = NeuralMarionette(opt).eval()
    net.load_state_dict(sd)
    net.anneal(1)
    vox = synth.figure_clip(B, T, G, seed=2)
    eps = synth.make_eps((T, 10, B, 128), seed=3)
    it = iter(eps)
    old = tdn._standard_normal
    tdn._standard_normal = lambda shape, dtype, device: next(it).clone()
    try:
        with torch.no_grad():
            ref = net(vox, {"detector": True, "learner": True})
    finally:
        tdn._standard_normal = old
    with torch.no_grad():
        mine = O.nm_forward(sd, o, vox, eps)
    for k in DETECTOR_LOSS_KEYS + ("kl_kypt", "kypt_recon_loss"):
        assert float(ref[k]) == float(mine[k]), k
    assert float(mine["vol_fit_reg"]) > 0


def test_full_forward_32_learnable_sigmas(ref_modules):
    """options.fixed_sigma = 0 (kypt_detector.py:258-260, 303-306; no shipped configuration selects it): a (K,) parameter
    `vox_to_kypt.sigmas`, created before the sub-modules, sigmas = sigmoid(parameter) * 2 gaussian_sigma per keypoint in the detector's
    Gaussian maps (decode_from_dyna keeps the fixed list, :226).  Layout, the seeded construction's RNG stream, forward bit for bit."""
    NeuralMarionette, _ = ref_modules
    import torch.distributions.normal as tdn
    import neural_marionette_amd as nm
    G, B, T = 32, 2, 5
    opt = _opt(G)
    opt.fixed_sigma = 0
    o = HotPathOptions.from_any(opt)
    assert o.fixed_sigma == 0
    torch.manual_seed(11); net = NeuralMarionette(opt).eval(); tail_ref = torch.rand(3)
    torch.manual_seed(11); mine_net = nm.NeuralMarionette(opt); tail_mine = torch.rand(3)
    ref_layout = [(k, tuple(v.shape)) for k, v in net.state_dict().items()]
    assert ref_layout == [(k, tuple(sh)) for k, sh in param_spec(o)]
    assert ref_layout[1] == ("kypt_detector.vox_to_kypt.sigmas", (24,))
    sa, sb = net.state_dict(), mine_net.state_dict()
    assert list(sa) == list(sb) and all(torch.equal(sa[k], sb[k]) for k in sa) and torch.equal(tail_ref, tail_mine)
    sd = synth.make_state_dict(o, seed=41, variant="peaky")
    sd["kypt_detector.vox_to_kypt.sigmas"] = torch.randn(24, generator=torch.Generator().manual_seed(42))
    net.load_state_dict(sd)
    net.anneal(1)
    vox = synth.figure_clip(B, T, G, seed=2)
    eps = synth.make_eps((T, 10, B, 128), seed=3)
    it = iter(eps)
    old = tdn._standard_normal
    tdn._standard_normal = lambda shape, dtype, device: next(it).clone()
    try:
        with torch.no_grad():
            ref = net(vox, {"detector": True, "learner": True})
            gen = net.kypt_detector.decode_from_dyna(ref["keypoints"][:, 1:3], ref["first_feature"], vox[:, 0])["gen"]
    finally:
        tdn._standard_normal = old
    with torch.no_grad():
        mine = O.nm_forward(sd, o, vox, eps)
        mine_gen = O.decode_from_keypoints(sd, o, ref["keypoints"][:, 1:3], ref["first_feature"], vox[:, 0])
    for k in ("recon", "keypoints", "kypt_recon", "R", "z_kypts", "h_kypts"):
        assert torch.equal(ref[k], mine[k]), k
    assert torch.equal(gen, mine_gen)
    for k in DETECTOR_LOSS_KEYS + ("kl_kypt", "kypt_recon_loss"):
        assert float(ref[k]) == float(mine[k]), k


def test_input_path_restatement_matches_reference():
    """synth.episodic_normalization / voxelize (the restated input path) against
    utils/dataset_utils.py of the reference: identical floats and identical occupancy grids."""
    sys.path.insert(0, REF)
    try:
        from utils import dataset_utils as DU
    finally:
        sys.path.remove(REF)
    rng = np.random.default_rng(4)
    for G, scale in ((64, 1.0), (64, 0.9), (96, 1.0)):
        pts = synth.figure_points(4, 4000, rng)
        a = DU.episodic_normalization(pts, scale=scale)
        b = synth.episodic_normalization(pts, scale=scale)
        assert np.array_equal(a, b)
        for t in range(4):
            assert np.array_equal(DU.voxelize(a[t], (G, G, G))[0], synth.voxelize(b[t], G))


def _ref_net_and_feed(ref_modules, G, seed):
    NeuralMarionette, _ = ref_modules
    opt = _opt(G)
    o = HotPathOptions.from_any(opt)
    sd = synth.make_state_dict(o, seed=seed, variant="peaky")
    net = NeuralMarionette(opt).eval()
    net.load_state_dict(sd)
    net.anneal(1)
    return net, o, sd


class _Feed:
    """hands pre-drawn eps tensors to Normal.rsample in call order (zeros once exhausted)"""

    def __init__(self, chunks):
        self.it = iter(chunks)

    def __call__(self, shape, dtype, device):
        e = next(self.it, None)
        return torch.zeros(shape, dtype=dtype) if e is None or tuple(e.shape) != tuple(shape) else e.clone()


def test_generation_driver_matches_reference_modules(ref_modules):
    """oracle.sample_generation against the rollout of vis_generation.py:81-136 re-run on the REFERENCE's own
    sub-modules (the script itself imports open3d/cv2 and cannot be imported here)."""
    import torch.distributions.normal as tdn
    from torch.distributions.normal import Normal
    import torch.nn.functional as F
    G, Tc, Tg, S = 32, 4, 5, 3
    net, o, sd = _ref_net_and_feed(ref_modules, G, 23)
    dm = net.dyna_module
    vox = synth.figure_clip(1, Tc, G, seed=3)[0]
    e_post, e_prior = synth.make_eps((Tc, S, 128), 5), synth.make_eps((Tg, S, 128), 6)
    old = tdn._standard_normal
    try:
        with torch.no_grad():
            tdn._standard_normal = _Feed([])
            det = net.kypt_detector(vox[None])
            kp = det["keypoints"]
            dm.encode(kp, det["affinity"])                     # builds the tree (its own draws are irrelevant)
            tdn._standard_normal = _Feed(list(e_post) + list(e_prior))
            h = dm.init_kypt_rnn_state.expand(S, -1)
            off = dm.get_offset(kp).expand(S, -1, -1, -1)
            gen = []
            for t in range(Tc):
                flat = kp[:, t].clone().view(1, -1).expand(S, -1)
                mu, sg = torch.chunk(dm.extract_post_dist(torch.cat([h, flat], -1)), 2, -1)
                z = Normal(mu, F.softplus(sg) + 1e-4).rsample()
                f, _ = dm.extract_kypt_from_latent_and_state(torch.cat([h, z], -1), off)
                i = (f - flat).pow(2).sum(-1).argmin()
                f, z, h = f[i][None].expand(S, -1), z[i][None].expand(S, -1), h[i][None].expand(S, -1)
                h = dm.kypt_rnn_cell(torch.cat([f, z], -1), h)
            for t in range(Tg):
                mu, sg = torch.chunk(dm.extract_prior_dist(h), 2, -1)
                z = Normal(mu, F.softplus(sg) + 1e-4).rsample()
                f, _ = dm.extract_kypt_from_latent_and_state(torch.cat([h, z], -1), off)
                gen.append(f.view(-1, 24, 4))
                h = dm.kypt_rnn_cell(torch.cat([f, z], -1), h)
            gen = torch.stack(gen, 0)[None]
            full0 = torch.cat([kp[:, :Tc], gen[:, :, 0]], dim=1)
            vox0 = net.kypt_detector.decode_from_dyna(full0, det["first_feature"], vox[None, 0])["gen"][0]
    finally:
        tdn._standard_normal = old
    with torch.no_grad():
        mine = O.sample_generation(sd, o, vox, Tg, S, e_post, e_prior)
    assert torch.equal(mine["keypoints_gen"], gen)
    assert torch.equal(mine["keypoints_cond"], kp[:, :Tc])
    assert torch.equal(mine["voxels_raw"][0], vox0)


def test_interpolation_driver_matches_reference_modules(ref_modules):
    """oracle.sample_interpolation against the loop of vis_interpolation.py:80-143 on the reference's sub-modules."""
    import torch.distributions.normal as tdn
    from torch.distributions.normal import Normal
    import torch.nn.functional as F
    G, T, S, rate = 32, 7, 24, 3
    net, o, sd = _ref_net_and_feed(ref_modules, G, 29)
    dm = net.dyna_module
    vox = synth.figure_clip(1, T, G, seed=8)[0]
    ea, eb = synth.make_eps((T, S, 128), 9), synth.make_eps((T, S, 128), 10)
    draws = []
    for t in range(T):
        draws += [ea[t], eb[t]] if (t % rate == 0 or t == T - 1) else [ea[t]]
    old = tdn._standard_normal
    try:
        with torch.no_grad():
            tdn._standard_normal = _Feed([])
            det = net.kypt_detector(vox[None])
            kp = det["keypoints"]
            dm.encode(kp, det["affinity"])
            tdn._standard_normal = _Feed(draws)
            h = dm.init_kypt_rnn_state.expand(S, -1)
            off = dm.get_offset(kp).expand(S, -1, -1, -1)
            sel, pend = [], []
            for t in range(T):
                flat = kp[:, t].clone().view(1, -1).expand(S, -1)
                if t % rate == 0 or t == T - 1:
                    mu, sg = torch.chunk(dm.extract_post_dist(torch.cat([h, flat], -1)), 2, -1)
                    pm, ps = torch.chunk(dm.extract_prior_dist(h), 2, -1)
                    z = Normal(mu, F.softplus(sg) + 1e-4).rsample()
                    zc = Normal(pm, F.softplus(ps) + 1e-4).rsample()
                    f, _ = dm.extract_kypt_from_latent_and_state(torch.cat([h, z], -1), off)
                    fc, _ = dm.extract_kypt_from_latent_and_state(torch.cat([h, zc], -1), off)
                    i = (f - flat).pow(2).sum(-1).argmin()
                    f, z, h = f[i][None].expand(S, -1), z[i][None].expand(S, -1), h[i][None].expand(S, -1)
                    j = (fc - f).pow(2).sum(-1).argmin()
                    pend.append(flat)
                    sel += [s_[j].view(24, 4) for s_ in pend]
                    pend = []
                else:
                    pm, ps = torch.chunk(dm.extract_prior_dist(h), 2, -1)
                    z = Normal(pm, F.softplus(ps) + 1e-4).rsample()
                    f, _ = dm.extract_kypt_from_latent_and_state(torch.cat([h, z], -1), off)
                    pend.append(f)
                h = dm.kypt_rnn_cell(torch.cat([f, z], -1), h)
            sel = torch.stack(sel, 0)[None]
            sel[0, :, :, -1] = sel[0, 0, :, -1]
    finally:
        tdn._standard_normal = old
    with torch.no_grad():
        mine = O.sample_interpolation(sd, o, vox, rate, S, ea, eb)
    assert torch.equal(mine["keypoints"], sel)


def test_eval_metrics_restatement_matches_reference():
    """oracle.voxel_chamfer_distance / semantic_votes vs utils/eval_utils.py run live (the fixture g7 pins the same numbers)."""
    sys.path.insert(0, REF)
    from utils import eval_utils as ref_eval
    from neural_marionette_amd import synth
    from oracle import nm_oracle as O
    vox, recon, kp, gt = synth.eval_inputs(seed=11, B=1, T=2, G=24, K=24, Kg=9)
    ch = ref_eval.voxel_chamfer_distance(None, dict(voxel=vox.clone(), recon=recon.clone()))
    pf = O.voxel_chamfer_distance(vox, recon)
    assert abs(pf.mean().item() - ch["scores_log"]) <= 1e-15
    se = ref_eval.semantic_scores(None, dict(keypoints=kp.clone(), gt_keypoints=gt.clone()))
    closest, counts = O.semantic_votes(kp, gt)
    assert np.array_equal(counts.numpy(), se["scores"].astype(np.int64))
    assert O.semantic_log(counts) == se["scores_log"]


# ---- keypoint counts of the reference's other dataset configs (dataset/config.py:97 panda K = 12, :124 hanco K = 28 with
# gaussian_sigma = 1.0; train.py:60 default 22): the shells construct, the state_dict layout is the reference's, and the oracle stays
# bit-identical to the reference there - so the GPU parity tests of tests/test_keypoint_counts_gpu.py are pinned for those K too.
def _opt_k(G, K, sigma=None):
    opt = _opt(G)
    opt.nkeypoints = K
    if sigma is not None:
        opt.gaussian_sigma = sigma
    return opt


@pytest.mark.parametrize("K", [12, 22, 28])
def test_state_dict_layout_other_keypoint_counts(ref_modules, K):
    NeuralMarionette, _ = ref_modules
    import neural_marionette_amd as nm
    opt = _opt_k(64, K)
    torch.manual_seed(5); ref = NeuralMarionette(opt)
    torch.manual_seed(5); net = nm.NeuralMarionette(opt)
    sa, sb = ref.state_dict(), net.state_dict()
    assert [(k, tuple(v.shape)) for k, v in sa.items()] == [(k, tuple(v.shape)) for k, v in sb.items()]
    assert [(k, tuple(v.shape)) for k, v in sa.items()] == [(k, tuple(s)) for k, s in param_spec(HotPathOptions(nkeypoints=K))]
    assert all(torch.equal(sa[k], sb[k]) for k in sa)
    assert tuple(sa["kypt_detector.affinity_params"].shape) == (2, K, K - 1)
    assert tuple(sa["kypt_detector.kypt_to_vox.adjust_combined_representation.0.weight"].shape) == (128, 128 + 2 * K + 3, 1, 1, 1)
    assert tuple(sa["dyna_module.joint_matrix_decoder.2.weight"].shape) == (6 * K, 128)


@pytest.mark.parametrize("K,sigma", [(12, None), (22, None), (28, 1.0)])
def test_full_forward_32_other_keypoint_counts(ref_modules, K, sigma):
    NeuralMarionette, _ = ref_modules
    import torch.distributions.normal as tdn
    G, B, T = 32, 2, 4
    opt = _opt_k(G, K, sigma)
    o = HotPathOptions.from_any(opt)
    assert o.nkeypoints == K
    sd = synth.make_state_dict(o, seed=40 + K, variant="peaky")
    gen = torch.Generator().manual_seed(K)
    sd["kypt_detector.affinity_params"] = torch.randn(sd["kypt_detector.affinity_params"].shape, generator=gen)
    net = NeuralMarionette(opt).eval()
    net.load_state_dict(sd)
    net.anneal(1)
    vox = synth.figure_clip(B, T, G, seed=2)
    eps = synth.make_eps((T, 10, B, 128), seed=3)
    it = iter(eps)
    old = tdn._standard_normal
    tdn._standard_normal = lambda shape, dtype, device: next(it).clone()
    try:
        with torch.no_grad():
            ref = net(vox, {"detector": True, "learner": True})
    finally:
        tdn._standard_normal = old
    with torch.no_grad():
        mine = O.nm_forward(sd, o, vox, eps)
    for k in ("recon", "keypoints", "heatmaps", "affinity", "first_feature", "kypt_recon", "R", "z_kypts", "h_kypts"):
        assert torch.equal(ref[k], mine[k]), k
    for k in DETECTOR_LOSS_KEYS + ("kl_kypt", "kypt_recon_loss"):
        assert float(ref[k]) == float(mine[k]), k
    assert np.array_equal(net.dyna_module.parents.numpy(), mine["parents"])
