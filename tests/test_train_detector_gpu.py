"""Detector-mode training (train.py:388-404 with pretrained_mode = 0): gradients of the weighted sum of the 11
detector losses w.r.t. every kypt_detector.* parameter, HIP backward kernels (through NeuralMarionette and the C ABI)
against autograd of the CPU oracle (bit-identical forward to the reference, see test_oracle_vs_reference.py) on the same
seeded inputs.  Single-loss weightings localise a failure to one backward kernel family."""
import numpy as np
import pytest
import torch

from neural_marionette_amd import NeuralMarionette, HotPathOptions, synth
from neural_marionette_amd.spec import DETECTOR_LOSS_KEYS
from oracle import nm_oracle as O

pytestmark = pytest.mark.gpu

# AIST loss weights of the reference (train.py:177-181 / pretrained/aist/opt.pickle)
AIST = dict(recon_loss=100.0, sparsity_loss=5.0, separation_loss=0.1, vol_fit_reg=10.0, kypt_const_loss=0.0,
            local_const_loss=1e-3, time_const_loss=1.0, sparsity_const_loss=0.01, intensity_const_loss=0.01,
            graph_traj_loss=1.0, graph_vol_loss=0.0)

# relative to each tensor's largest gradient entry, against the fp64 oracle (the fp32 oracle's own distance: up to 3e-3)
TOL = 2e-3

WEIGHTINGS = {
    "aist": AIST,
    "recon": dict(recon_loss=100.0),
    "sparsity": dict(sparsity_loss=5.0),
    "separation": dict(separation_loss=0.1),
    "volfit": dict(vol_fit_reg=10.0),
    "graph": dict(local_const_loss=1.0, time_const_loss=1.0, sparsity_const_loss=1.0),
    "traj": dict(graph_traj_loss=1.0),
}


def _setup(G=32, B=2, T=4, seed=11, variant="peaky", K=24):
    o = HotPathOptions(grid_size=G, nkeypoints=K)
    sd = synth.make_state_dict(o, seed=seed, variant=variant)
    gen = torch.Generator().manual_seed(seed + 1)
    sd["kypt_detector.affinity_params"] = torch.randn(sd["kypt_detector.affinity_params"].shape, generator=gen)
    vox = synth.figure_clip(B, T, G, seed=seed + 2)
    return o, sd, vox


_ORACLE_GRAD_CACHE = {}


def _oracle_grads(o, sd, vox, weights, affinity_on=True, double=False):
    """Autograd of the oracle.  double=True: the same graph in fp64 — the reference value the fp32 gradients of both
    implementations scatter around (the fp32 oracle itself is 1e-3 relative away from it on the deepest layers).
    Cached per (weights, clip, loss weighting) for the session: several tests compare different conv modes against the same
    fp64 gradient (10-15 s of CPU each)."""
    key = (o.grid_size, o.nkeypoints, tuple(vox.shape), float(vox.sum()), float(sd["kypt_detector.affinity_params"].double().sum()),
           float(sd["kypt_detector.kypt_to_vox.decode_voxel_from_combined_representation.11.weight"].double().abs().sum()),
           tuple(sorted(weights.items())), affinity_on, double)
    if key in _ORACLE_GRAD_CACHE:
        return _ORACLE_GRAD_CACHE[key]
    res = _oracle_grads_uncached(o, sd, vox, weights, affinity_on, double)
    _ORACLE_GRAD_CACHE[key] = res
    return res


def _oracle_grads_uncached(o, sd, vox, weights, affinity_on=True, double=False):
    if double:
        sd, vox = {k: v.double() for k, v in sd.items()}, vox.double()
    names = [k for k in sd if k.startswith("kypt_detector.")]
    leaf = {k: sd[k].clone().requires_grad_(True) for k in names}
    sd2 = dict(sd); sd2.update(leaf)
    out = O.detector_forward(sd2, o, vox, affinity_on=affinity_on)
    loss = sum(w * out[k] for k, w in weights.items())
    grads = torch.autograd.grad(loss, [leaf[k] for k in names], allow_unused=True)
    return float(loss.detach()), {k: (g if g is not None else torch.zeros_like(leaf[k])) for k, g in zip(names, grads)}, out


def _hip_grads(o, sd, vox, weights, affinity_on=True, mode=None):
    net = NeuralMarionette(o)
    net.load_state_dict(sd)
    net = net.cuda().train()
    if mode:
        net.set_conv_mode(mode)
    if affinity_on:
        net.anneal(1)
    assert net.kypt_detector.affinity_start == affinity_on
    acts = {"detector": True, "learner": False}
    net.control_active(acts)
    net.zero_grad()
    out = net(vox.cuda(), acts)
    loss = sum(w * out[k] for k, w in weights.items())
    loss.backward()
    torch.cuda.synchronize()
    grads = {"kypt_detector." + n: (p.grad.detach().cpu() if p.grad is not None else None) for n, p in net.kypt_detector.named_parameters()}
    return float(loss.detach()), grads, out


def _compare(ref, got, tol, report=True):
    worst = ("", 0.0)
    gmax = max(r.abs().max().item() for r in ref.values())
    bad = []
    for k, r in ref.items():
        g = got[k]
        assert g is not None, f"{k}: no gradient"
        assert torch.isfinite(g).all(), f"{k}: non-finite gradient"
        scale = max(r.abs().max().item(), 1e-6 * gmax, 1e-30)
        e = (g.double() - r.double()).abs().max().item() / scale
        if e > worst[1]:
            worst = (k, e)
        if e >= tol:
            bad.append((k, e, r.abs().max().item()))
    if report:
        print("worst relative gradient error %.2e at %s (largest |grad| %.3e)" % (worst[1], worst[0], gmax))
    assert not bad, "gradient mismatch: " + "; ".join("%s rel %.2e (|g|max %.2e)" % b for b in bad[:12])


@pytest.mark.parametrize("which", list(WEIGHTINGS))
def test_detector_gradients_vs_oracle_autograd(which):
    o, sd, vox = _setup()
    w = WEIGHTINGS[which]
    ref_loss, ref, ref_out = _oracle_grads(o, sd, vox, w, double=True)
    loss, got, out = _hip_grads(o, sd, vox, w)
    assert abs(loss - ref_loss) <= 2e-5 * max(1.0, abs(ref_loss)), (loss, ref_loss)
    _compare(ref, got, tol=TOL)


def test_detector_gradients_vs_reference_fixture(golden_dir):
    """The reference's own autograd on the same seeded case (tests/golden/g8_detector_grads.npz, tools/make_golden.py g8): every
    997th element and the max-abs of each of the 315 gradients."""
    import os
    g = np.load(os.path.join(golden_dir, "g8_detector_grads.npz"))
    G, B, T, seed = [int(v) for v in g["meta"]]
    o, sd, vox = _setup(G=G, B=B, T=T, seed=seed)
    w = {k: float(v) for k, v in zip(DETECTOR_LOSS_KEYS, g["weights"])}
    loss, got, out = _hip_grads(o, sd, vox, w)
    assert abs(loss - float(g["loss"])) <= 2e-5 * abs(float(g["loss"]))
    for i, k in enumerate(DETECTOR_LOSS_KEYS):
        assert abs(float(out[k]) - float(g["losses"][i])) <= 2e-5 * max(1.0, abs(float(g["losses"][i]))), k
    gmax = max(float(g[f][2]) for f in g.files if f.startswith("g:"))
    worst = 0.0
    for name, gr in got.items():
        ref = g["g:" + name[len("kypt_detector."):]]
        flat = gr.reshape(-1).double()
        scale = max(float(ref[2]), 1e-6 * gmax)
        e = np.abs(flat[::997].numpy() - ref[3:]).max() / scale
        worst = max(worst, e)
        assert e < 3e-3, (name, e)      # both sides are fp32: the fixture itself is up to 1e-3 from the fp64 gradient
    print("worst relative difference to the reference's gradients %.2e" % worst)


@pytest.mark.parametrize("seed", [11, 104])
def test_detector_gradients_f16_mode(seed):
    """Reduced-precision conv mode 'f16' (nm_set_conv_mode 3: conv products of fp16-rounded operands, fp32 accumulation and storage -
    the training arithmetic BASELINE.json's config 3 asks for under the name bf16, with three more operand bits) against the fp64
    oracle, training weighting.  Stated tolerance of the mode (measured: tools/diag_f16.py - loss 5e-5, keypoints 4e-4, global
    gradient L2 7e-3, worst tensor L2 0.10 / cosine 0.995 on the deep hourglass convs in front of a GroupNorm, whose backward
    removes the common mode): loss 5e-4 relative, keypoints 2e-3, whole-gradient L2 distance 2e-2, every parameter tensor's
    gradient within 0.25 L2-relative and 0.98 cosine of the fp64 gradient.  The fp32-equivalent modes are held to 2e-3 per entry."""
    o, sd, vox = _setup(seed=seed)
    ref_loss, ref, ref_out = _oracle_grads(o, sd, vox, AIST, double=True)
    loss, got, out = _hip_grads(o, sd, vox, AIST, mode="f16")
    assert abs(loss - ref_loss) <= 5e-4 * max(1.0, abs(ref_loss)), (loss, ref_loss)
    assert (out["keypoints"].detach().cpu().double() - ref_out["keypoints"].double()).abs().max().item() < 2e-3
    gmax = max(r.abs().max().item() for r in ref.values())
    num = den = 0.0
    for k, r in ref.items():
        g, r = got[k].double(), r.double()
        assert torch.isfinite(g).all(), k
        num += ((g - r) ** 2).sum().item(); den += (r ** 2).sum().item()
        if r.abs().max().item() > 1e-6 * gmax:
            rel = ((g - r).norm() / r.norm()).item()
            cos = (g.flatten() @ r.flatten()).item() / (g.norm().item() * r.norm().item())
            assert rel < 0.25 and cos > 0.98, (k, rel, cos)
    assert (num / den) ** 0.5 < 2e-2, (num / den) ** 0.5


def test_detector_gradients_exact_fp32_mode():
    """Exact fp32-MFMA convolutions everywhere (conv mode 'fp32'): under the training weighting the HIP gradients sit closer to the
    fp64 gradients than the fp32 oracle does.  (The single-loss weightings are not run in this mode: a keypoint-only loss sends
    an almost constant gradient field through every GroupNorm, whose backward removes the common mode — the 2e-6 accumulation
    noise of the fp32 MFMA chain is amplified there by 1e3-1e4, measured; with the reconstruction loss present it is not.)"""
    o, sd, vox = _setup()
    ref_loss, ref, _ = _oracle_grads(o, sd, vox, AIST, double=True)
    loss, got, _ = _hip_grads(o, sd, vox, AIST, mode="fp32")
    assert abs(loss - ref_loss) <= 2e-5 * max(1.0, abs(ref_loss))
    _compare(ref, got, tol=1e-3)


def test_detector_gradients_before_affinity_start():
    """anneal(0): affinity is None, the graph losses are zero and affinity_params gets no gradient (kypt_detector.py:71-78,111-117)."""
    o, sd, vox = _setup(seed=23)
    ref_loss, ref, _ = _oracle_grads(o, sd, vox, AIST, affinity_on=False, double=True)
    loss, got, _ = _hip_grads(o, sd, vox, AIST, affinity_on=False)
    assert abs(loss - ref_loss) <= 2e-5 * max(1.0, abs(ref_loss))
    _compare(ref, got, tol=TOL)
    assert got["kypt_detector.affinity_params"].abs().max().item() == 0.0


def test_detector_gradients_odd_hourglass_40():
    """G = 40: hourglass sizes 10 -> 5 -> 2 -> 1 with output_padding in the transposed convs (vox_modules.py:81)."""
    o, sd, vox = _setup(G=40, B=1, T=3, seed=31)
    ref_loss, ref, _ = _oracle_grads(o, sd, vox, AIST, double=True)
    loss, got, _ = _hip_grads(o, sd, vox, AIST)
    assert abs(loss - ref_loss) <= 2e-5 * max(1.0, abs(ref_loss))
    _compare(ref, got, tol=TOL)


@pytest.mark.parametrize("K", [16, 32])
@pytest.mark.parametrize("mode", ["split16", "fp32"])
def test_detector_gradients_other_keypoint_counts(K, mode):
    """K = 16 / 32 keypoints (the combined representation has 2K + 131 channels, the heads K), both conv modes.
    Seeds: the losses contain selections (nearest keypoint of the chamfer term, strongest neighbour of the graph terms); a seed on
    which one of them is a near-tie makes the gradient a coin flip of the last fp32 bit.  K = 32, seed 103 is NOT run here: it lands
    3.3e-3 from the fp64 oracle in every conv mode for another reason - one-ulp forward differences amplified by the GroupNorm
    backward passes - which test_seed_103_deviation_is_rounding_noise_amplification measures instead of avoiding."""
    o, sd, vox = _setup(G=32, B=1, T=3, seed={16: 87, 32: 104}[K], K=K)
    ref_loss, ref, _ = _oracle_grads(o, sd, vox, AIST, double=True)
    loss, got, _ = _hip_grads(o, sd, vox, AIST, mode=mode)
    assert abs(loss - ref_loss) <= 2e-5 * max(1.0, abs(ref_loss))
    _compare(ref, got, tol=TOL)


def test_detector_training_trajectory_vs_reference_fixture(golden_dir):
    """Three detector-mode training steps (Adam lr 4e-4, AIST weights) against fixture G13 - the REFERENCE's own training run
    (torch.optim.Adam on its autograd, train.py:376-412; tools/make_golden.py::case_g13): the loss of every step, and every updated
    tensor after the third step on the fixture's sub-sample.  (Rounds 2-5 trained the CPU oracle on the GPU box here.)"""
    import numpy as np, os
    from neural_marionette_amd.train import DetectorTrainer
    g = np.load(os.path.join(golden_dir, "g13_detector_training20.npz"), allow_pickle=False)
    G, B, T, seed, _ = [int(v) for v in g["meta"]]
    o, sd, vox = _setup(G=G, B=B, T=T, seed=seed)
    ref_losses = [float(v) for v in g["losses"][:3]]
    net = NeuralMarionette(o)
    net.load_state_dict(sd)
    net = net.cuda().train()
    net.anneal(1)
    tr = DetectorTrainer(net, lr=4e-4)
    losses = [tr.step(vox.cuda())["loss"] for _ in range(3)]
    torch.cuda.synchronize()
    print("detector training losses", losses, "reference", ref_losses)
    for a, b in zip(losses, ref_losses):
        assert abs(a - b) <= 2e-4 * abs(b), (losses, ref_losses)
    assert losses[2] < losses[0]
    # updated weights: Adam's first steps move every weight by ~lr * sign(g), so an entry whose gradient is rounding noise may
    # legitimately end up 2 * lr * steps away; the bulk must agree far below lr
    num = den = 0.0
    for n, p in net.kypt_detector.named_parameters():
        r = g["w3:" + n]
        f = p.detach().reshape(-1).double().cpu()
        num += float((f[::997] - torch.from_numpy(r[2:])).abs().sum()); den += r.size - 2
        # (a tensor's SUM is not compared: where a gradient is rounding noise Adam's first steps move the entry by +-lr whatever its size, and
        #  72 such entries of a bias differ by a few lr in their sum between any two fp32 evaluations)
    print("mean weight difference after 3 steps %.2e over %d sampled entries (lr 4e-4, each weight moved ~1e-3)" % (num / den, den))
    assert num / den < 2e-5


def test_joint_step_detector_and_learner_active():
    """Both modules active in one step (detector_time and learner_time overlapping, train.py:177-199): the detector trains on its
    11 losses, the learner on keypoints.detach() (neural_marionette.py:53); every one of the 336 trainable tensors gets the
    gradient the oracle's autograd gives for the same total loss."""
    from neural_marionette_amd.train import DETECTOR_LOSS_WEIGHTS, LEARNER_LOSS_WEIGHTS
    o, sd, vox = _setup(seed=51)
    B, T = vox.shape[:2]
    eps = synth.make_eps((T, 10, B, o.nlatent_kypt), seed=52)
    # oracle
    names = [k for k in sd if k != "dyna_module.offset_param"]
    leaf = {k: sd[k].clone().double().requires_grad_(True) for k in names}
    sd2 = {k: v.double() for k, v in sd.items()}; sd2.update(leaf)
    det = O.detector_forward(sd2, o, vox.double(), affinity_on=True)
    net = NeuralMarionette(o)
    net.load_state_dict(sd)
    net = net.cuda().train()
    net.anneal(1)
    acts = {"detector": True, "learner": True}
    net.control_active(acts)
    net.zero_grad()
    out = net(vox.cuda(), acts, eps=eps.cuda())
    order, parents = net.dyna_module.priority.indices.cpu().numpy(), net.dyna_module.parents.cpu().numpy()
    enc = O.vrnn_encode(sd2, o, det["keypoints"].detach(), order, parents, eps.double())
    ref_loss = sum(w * det[k] for k, w in DETECTOR_LOSS_WEIGHTS.items()) + sum(w * enc[k] for k, w in LEARNER_LOSS_WEIGHTS.items())
    ref = dict(zip(names, torch.autograd.grad(ref_loss, [leaf[k] for k in names], allow_unused=True)))
    loss = sum(w * out[k] for k, w in DETECTOR_LOSS_WEIGHTS.items()) + sum(w * out[k] for k, w in LEARNER_LOSS_WEIGHTS.items())
    loss.backward()
    torch.cuda.synchronize()
    assert abs(float(loss.detach()) - float(ref_loss.detach())) <= 2e-5 * abs(float(ref_loss.detach()))
    got = {n: p.grad for n, p in net.named_parameters()}
    assert got["dyna_module.offset_param"] is None or float(got["dyna_module.offset_param"].abs().max()) == 0.0
    checked = 0
    for k, r in ref.items():
        assert r is not None and got[k] is not None, k
        e = (got[k].cpu().double() - r).abs().max().item() / max(r.abs().max().item(), 1e-12)
        assert e < 3e-3, (k, e)
        checked += 1
    assert checked == 336


def test_training_api_misuse_is_reported():
    """nm_detector_backward without a training forward, and a training forward without the training weight packs, fail loudly."""
    from neural_marionette_amd import _lib
    o, sd, vox = _setup(seed=61)
    net = NeuralMarionette(o)
    net.load_state_dict(sd)
    net = net.cuda().eval()
    net.anneal(1)
    with torch.no_grad():
        net(vox.cuda(), {"detector": True, "learner": False})
    eng = net._engine
    dl = torch.ones(len(DETECTOR_LOSS_KEYS), device="cuda")
    g = torch.zeros(3, device="cuda")
    arr = (_lib.NmNamedTensor * 1)()
    arr[0].name, arr[0].data, arr[0].numel = b"kypt_detector.affinity_params", g.data_ptr(), g.numel()
    with pytest.raises(_lib.NmError, match="no training forward"):
        eng.call("nm_detector_backward", _lib.ptr(dl), arr, 1)
    kp = torch.empty(2, 4, o.nkeypoints, 4, device="cuda")
    with pytest.raises(_lib.NmError, match="nm_ctx_set_training"):
        eng.call("nm_detector_forward_train", _lib.ptr(vox.cuda()), 2, 4, 1, _lib.ptr(kp), _lib.ptr(kp), _lib.ptr(kp), _lib.ptr(kp), None, _lib.ptr(kp))


def test_detector_gradients_batch_additivity_at_64cubed():
    """Full-size property (64^3, the bench grid), no oracle needed: every loss is a mean over clips and GroupNorm is per frame, so the
    gradient of a two-clip batch is the average of the two single-clip gradients."""
    o, sd, vox = _setup(G=64, B=2, T=4, seed=81)
    _, g_ab, _ = _hip_grads(o, sd, vox, AIST)
    _, g_a, _ = _hip_grads(o, sd, vox[:1].contiguous(), AIST)
    _, g_b, _ = _hip_grads(o, sd, vox[1:].contiguous(), AIST)
    gmax = max(v.abs().max().item() for v in g_ab.values())
    worst = 0.0
    for k, v in g_ab.items():
        avg = 0.5 * (g_a[k].double() + g_b[k].double())
        scale = max(avg.abs().max().item(), 1e-6 * gmax)
        e = (v.double() - avg).abs().max().item() / scale
        worst = max(worst, e)
        assert e < 2e-3, (k, e)
    print("batch additivity at 64^3: worst relative deviation %.2e" % worst)
    # run-to-run identical (fixed-order reductions everywhere, integer atomics only for the abs-max)
    _, g_ab2, _ = _hip_grads(o, sd, vox, AIST)
    for k, v in g_ab.items():
        assert torch.equal(v, g_ab2[k]), f"{k}: gradients differ between two runs"


def test_config3_per_gpu_shape_training_step():
    """BASELINE config 3's per-GPU shape (64^3, T = 16, B = 4 clips per GPU, detector-mode training step, train.py:376-412), no oracle
    needed: (1) the loss of the training forward equals the inference forward's, (2) the gradient of the 4-clip batch is the average
    of the four single-clip gradients (every loss is a mean over clips, GroupNorm is per frame), (3) two runs are bit-identical,
    (4) DetectorTrainer.step - direct library calls into the all-reduce bucket - produces exactly the autograd path's gradients
    and its Adam step moves every parameter."""
    from neural_marionette_amd.train import DetectorTrainer, DETECTOR_LOSS_WEIGHTS
    o, sd, vox = _setup(G=64, B=4, T=16, seed=91)
    loss, g_all, out = _hip_grads(o, sd, vox, AIST)
    # (1) inference forward on the same weights / clips
    net = NeuralMarionette(o); net.load_state_dict(sd); net = net.cuda().eval(); net.anneal(1)
    with torch.no_grad():
        inf = net(vox.cuda(), {"detector": True, "learner": False})
    loss_inf = float(sum(w * inf[k] for k, w in AIST.items()))
    assert abs(loss - loss_inf) <= 1e-6 * max(1.0, abs(loss_inf)), (loss, loss_inf)
    # (the inference forward takes shortcuts the training forward cannot - two hourglass levels in one launch, the split 1x1 conv -
    # whose summation orders differ: same values to fp32 rounding, not the same bits)
    assert (out["keypoints"].cpu() - inf["keypoints"].cpu()).abs().max().item() < 2e-6
    # (2) batch additivity
    acc = {k: torch.zeros_like(v, dtype=torch.float64) for k, v in g_all.items()}
    for b in range(4):
        _, g_b, _ = _hip_grads(o, sd, vox[b:b + 1].contiguous(), AIST)
        for k in acc:
            acc[k] += g_b[k].double() / 4
    gmax = max(v.abs().max().item() for v in g_all.values())
    worst = 0.0
    for k, v in g_all.items():
        scale = max(acc[k].abs().max().item(), 1e-6 * gmax)
        e = (v.double() - acc[k]).abs().max().item() / scale
        worst = max(worst, e)
        assert e < 2e-3, (k, e)
    print("config-3 shape: batch additivity worst relative deviation %.2e, loss %.6f" % (worst, loss))
    # (3) run-to-run identity
    _, g_again, _ = _hip_grads(o, sd, vox, AIST)
    for k, v in g_all.items():
        assert torch.equal(v, g_again[k]), f"{k}: gradients differ between two runs"
    # (4) the trainer's direct path
    net2 = NeuralMarionette(o); net2.load_state_dict(sd); net2 = net2.cuda().train(); net2.anneal(1)
    before = {k: v.detach().clone() for k, v in net2.state_dict().items()}
    tr = DetectorTrainer(net2, lr=4e-4)
    log = tr.step(vox.cuda())
    assert abs(log["loss"] - loss) <= 1e-6 * max(1.0, abs(loss))
    for n, p in net2.kypt_detector.named_parameters():
        k = "kypt_detector." + n
        assert p.grad.data_ptr() == tr.bucket.views[k].data_ptr()
        assert torch.equal(p.grad.cpu(), g_all[k]), f"{k}: trainer gradient differs from the autograd path"
        if g_all[k].abs().max() > 0:
            assert not torch.equal(p.detach(), before[k]), f"{k}: not updated"
    assert set(log) >= {"loss", *[k for k in DETECTOR_LOSS_WEIGHTS]}


@pytest.mark.parametrize("mode", [None, "f16", "bf16"])
def test_vanishing_loss_weights_give_finite_proportional_gradients(mode):
    """Power-of-two operand scaling of the data-gradient convs (nm_grad.hip make_scale): with loss weights of 1e-30 every dy is
    ~1e-30 and the scale 2^k must stay finite in both directions (the exponent is clamped to +-100): gradients finite and
    proportional to the unscaled ones.  Also in the one-product modes: there the k2 s2 weight gradients (wgrad16k2) and
    the transposed convs / pool data gradients (convT2_f16) run on the f16 matrix cores too, and a dY of 1e-30 that reached them
    unscaled would flush to zero in fp16 - every dY they see must have gone through the same power-of-two scaling ('bf16': storage
    threshold at 16^3 so that the 32^3 tensors and their gradients are bfloat16)."""
    import os
    o, sd, vox = _setup(G=32, B=1, T=3, seed=95)
    if mode == "bf16":
        os.environ["NM355_STORE16_MIN"] = "4096"
    try:
        _, g1, _ = _hip_grads(o, sd, vox, AIST, mode=mode)
        tiny = {k: w * 1e-30 for k, w in AIST.items()}
        _, g2, _ = _hip_grads(o, sd, vox, tiny, mode=mode)
    finally:
        os.environ.pop("NM355_STORE16_MIN", None)
    gmax = max(v.abs().max().item() for v in g1.values())
    # (bfloat16 gradient storage rounds g and 1e-30 g independently: 2^-8 relative instead of fp32's 2^-24)
    tol = 1e-3 if mode != "bf16" else 2e-2
    for k, v in g1.items():
        assert torch.isfinite(g2[k]).all(), k
        scale = max(v.abs().max().item(), 1e-6 * gmax)
        e = (g2[k].double() * 1e30 - v.double()).abs().max().item() / scale
        assert e < tol, (k, e, mode)


def test_trainer_with_frozen_parameters():
    """KyptDetector.anneal keeps affinity_params frozen while affinity_anneal > nepoch (train.py:86, kypt_detector.py:71-78), and a user
    may freeze any layer: DetectorTrainer.step must run (the library still writes every detector gradient into the bucket) and leave
    the frozen tensors untouched while the others move exactly as without the freeze."""
    from neural_marionette_amd.train import DetectorTrainer
    o = HotPathOptions(grid_size=32, affinity_anneal=5)
    sd = synth.make_state_dict(o, seed=71, variant="peaky")
    vox = synth.figure_clip(1, 3, 32, seed=72).cuda()
    frozen = "vox_to_kypt.extract_features.2.res_branch.0.weight"

    def run(freeze):
        net = NeuralMarionette(o); net.load_state_dict(sd); net = net.cuda().train()
        net.anneal(0)                                         # affinity_anneal 5 > epoch 0: affinity off, affinity_params frozen
        assert net.kypt_detector.affinity_start is False and net.kypt_detector.affinity_params.requires_grad is False
        tr = DetectorTrainer(net, lr=1e-3)
        if freeze:
            dict(net.kypt_detector.named_parameters())[frozen].requires_grad = False
        logs = [tr.step(vox) for _ in range(2)]
        return net, logs

    net_a, log_a = run(False)
    net_b, log_b = run(True)
    pa, pb = dict(net_a.kypt_detector.named_parameters()), dict(net_b.kypt_detector.named_parameters())
    assert torch.equal(pa["affinity_params"].cpu(), sd["kypt_detector.affinity_params"])
    assert torch.equal(pb[frozen].cpu(), sd["kypt_detector." + frozen]) and not torch.equal(pa[frozen].cpu(), sd["kypt_detector." + frozen])
    assert pb[frozen].grad is None
    assert log_a[0]["loss"] == log_b[0]["loss"]
    # epoch 5: anneal starts the affinity and un-freezes it; the same trainer object keeps working (Adam state grows by one tensor)
    tr = DetectorTrainer(net_a, lr=1e-3)
    tr.step(vox)
    net_a.anneal(5)
    assert net_a.kypt_detector.affinity_params.requires_grad and net_a.kypt_detector.affinity_start
    before = net_a.kypt_detector.affinity_params.detach().clone()
    tr.step(vox)
    assert not torch.equal(before, net_a.kypt_detector.affinity_params.detach())


def test_trainer_step_rccl_one_rank_group():
    """The distributed leg of DetectorTrainer.step on ONE device: a 1-rank `nccl` (= RCCL) process group, both bucket chunks really
    all-reduced on the side stream behind the library-recorded event; the step must equal the no-group step bit for bit (a 1-rank
    sum is the identity) and the collectives must have been issued."""
    import os, socket
    import torch.distributed as dist
    from neural_marionette_amd.train import DetectorTrainer, GradBucket
    o, sd, vox = _setup(G=32, B=2, T=3, seed=73)

    def run():
        net = NeuralMarionette(o); net.load_state_dict(sd); net = net.cuda().train(); net.anneal(1)
        tr = DetectorTrainer(net, lr=4e-4)
        logs = [tr.step(vox.cuda()) for _ in range(2)]
        torch.cuda.synchronize()
        return net, tr, logs

    net0, _, log0 = run()
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    calls = []
    real = dist.all_reduce

    def counting(t, *a, **k):
        calls.append((t.numel(), torch.cuda.current_stream().cuda_stream))
        return real(t, *a, **k)
    try:
        GradBucket.always_reduce = True
        dist.all_reduce = counting
        net1, tr1, log1 = run()
    finally:
        dist.all_reduce = real
        GradBucket.always_reduce = False
        dist.destroy_process_group()
    n_dec = tr1.bucket.chunks[0][1] - tr1.bucket.chunks[0][0]
    n_rest = tr1.bucket.chunks[1][1] - tr1.bucket.chunks[1][0]
    assert [c[0] for c in calls] == [n_dec, n_rest, n_dec, n_rest], calls              # two chunks per step, decoder chunk first
    assert all(c[1] == tr1.bucket.comm_stream.cuda_stream for c in calls), "the collectives must run on the bucket's side stream"
    assert n_dec > 0 and n_dec + n_rest == sum(p.numel() for p in net1.kypt_detector.parameters())
    assert [l["loss"] for l in log0] == [l["loss"] for l in log1]
    for (n, p), (_, q) in zip(net0.named_parameters(), net1.named_parameters()):
        assert torch.equal(p, q), n


def test_gradient_properties_f16_mode_at_64cubed():
    """Conv mode 'f16' at the bench grid (64^3): batch additivity of the gradient and run-to-run identity, no oracle needed."""
    o, sd, vox = _setup(G=64, B=2, T=4, seed=83)
    l_ab, g_ab, _ = _hip_grads(o, sd, vox, AIST, mode="f16")
    _, g_a, _ = _hip_grads(o, sd, vox[:1].contiguous(), AIST, mode="f16")
    _, g_b, _ = _hip_grads(o, sd, vox[1:].contiguous(), AIST, mode="f16")
    gmax = max(v.abs().max().item() for v in g_ab.values())
    worst = 0.0
    for k, v in g_ab.items():
        assert torch.isfinite(v).all(), k
        avg = 0.5 * (g_a[k].double() + g_b[k].double())
        scale = max(avg.abs().max().item(), 1e-6 * gmax)
        worst = max(worst, (v.double() - avg).abs().max().item() / scale)
    print("f16 mode, 64^3: batch additivity worst relative deviation %.2e" % worst)
    # (the 2e-3 of the fp32-equivalent modes: measured 2.1e-4.  Rounds 2-3 measured 7.6e-3 here and blamed the batch-wide power-of-two
    # scale of dy; it was the transposed-conv kernel choice, made on the BATCH's voxel count, that gave the 8^3 level fp32 arithmetic
    # alone and fp16 arithmetic in a batch - the f16 kernel's rule is per frame now, nm_convT2_f16_eligible)
    assert worst < 2e-3
    l2, g2, _ = _hip_grads(o, sd, vox, AIST, mode="f16")
    assert l2 == l_ab
    for k, v in g_ab.items():
        assert torch.equal(v, g2[k]), k
    # against the fp32-equivalent mode at the same size: loss 5e-4, whole-gradient L2 within the mode's stated 2e-2
    l32, g32, _ = _hip_grads(o, sd, vox, AIST)
    assert abs(l_ab - l32) <= 5e-4 * abs(l32)
    num = sum(((g_ab[k].double() - g32[k].double()) ** 2).sum().item() for k in g32)
    den = sum((g32[k].double() ** 2).sum().item() for k in g32)
    print("f16 vs split16 at 64^3: loss %.6f vs %.6f, whole-gradient L2 distance %.2e" % (l_ab, l32, (num / den) ** 0.5))
    assert (num / den) ** 0.5 < 2e-2


def test_gradient_is_bit_identical_over_many_evaluations():
    """Forty evaluations of the same detector gradient (G = 32, B = 2, T = 3) must agree bit for bit.  Round 4 found the weight gradient
    of one decoder layer differing by ~1e-4 in one evaluation out of five: the split-fp16 weight-gradient kernels read their LDS
    scale / shift table without a barrier behind its fill, and beside the main stream's kernels a wave ran ahead often enough to see
    what the previous workgroup had left there (one run in ~80 at the end of round 3, when the kernel mostly ran alone).  The
    64^3 identity tests repeat an evaluation once; a race of that rate needs many."""
    o, sd, vox = _setup(G=32, B=2, T=3, seed=73)
    net = NeuralMarionette(o); net.load_state_dict(sd); net = net.cuda().train(); net.anneal(1)
    acts = {"detector": True, "learner": False}
    net.control_active(acts)
    v = vox.cuda()

    def grads():
        net.zero_grad()
        out = net(v, acts)
        sum(w * out[k] for k, w in AIST.items()).backward()
        torch.cuda.synchronize()
        return {n: p.grad.detach().clone() for n, p in net.kypt_detector.named_parameters() if p.grad is not None}
    ref = grads()
    for i in range(40):
        g = grads()
        bad = [k for k in ref if not torch.equal(ref[k], g[k])]
        assert not bad, "evaluation %d differs in %s" % (i, bad[:4])


def test_seed_103_deviation_is_rounding_noise_amplification():
    """K = 32, seed 103: every HIP conv mode lands 3.3e-3 (relative to the tensor's largest entry) from the fp64 oracle on
    the early layers of the per-frame net, where the neighbouring seeds give 1e-4.  Measured here in fp64 alone, free of any fp32
    evaluation order: the fp64 oracle's own gradient JUMPS by 1.0e-3 (same metric) under some perturbations of the weights by one fp32
    ulp (6e-8 relative) and moves by 6e-6 under others of the same size (draws 1 and 2 below); seed 104 moves by 1.6e-6.  So within
    one ulp of this weight set the gradient is discontinuous - the network is piecewise smooth (LeakyReLU kinks, the max over keypoints,
    the nearest-keypoint selection of the chamfer term), and an element of a coarse layer (4^3 voxels at the hourglass bottom: one
    flipped LeakyReLU branch there carries 1 / 64 of a channel) sits within an ulp of its kink - and an fp32 evaluation can land on
    either side of it: torch's own fp32 autograd of the oracle sits 3.1e-3 from fp64 at this seed (2e-4 at seed 104), the HIP path 3.3e-3 in
    every conv mode and every tree so far.  The test pins exactly that: the jump exists in exact arithmetic, the neighbour has none, and
    the HIP gradient is within a few jumps of fp64.
    (Rounds 2-4 measured the distance between two HIP evaluations whose GroupNorm statistics differ by one ulp (NM355_GN_DIAG=1)
    instead: 5e-4 ... 3e-3 in those trees, 7.9e-5 after round 5 reordered the heat-map sums - whether ONE particular one-ulp change
    crosses the tie depends on the rounding realisation, so that figure is printed, not asserted.)"""
    import os
    o, sd, vox = _setup(G=32, B=1, T=3, seed=103, K=32)
    _, ref64, _ = _oracle_grads(o, sd, vox, AIST, double=True)
    gmax = max(r.abs().max().item() for r in ref64.values())

    def dist(a, b):
        w = 0.0
        for k, r in ref64.items():
            scale = max(r.abs().max().item(), 1e-6 * gmax)
            w = max(w, (a[k].double() - b[k].double()).abs().max().item() / scale)
        return w
    # conditioning in fp64: weights perturbed by one fp32 ulp (relative), rounded to fp32, evaluated in fp64
    cond = []
    for draw in (1, 2):
        gen = torch.Generator().manual_seed(draw)
        sdp = {k: ((v.double() * (1 + (torch.rand(v.shape, generator=gen, dtype=torch.float64) * 2 - 1) * 6e-8)).float() if v.is_floating_point() else v)
               for k, v in sd.items()}
        _, gp, _ = _oracle_grads_uncached(o, sdp, vox, AIST, double=True)
        cond.append(dist(gp, ref64))
    o4, sd4, vox4 = _setup(G=32, B=1, T=3, seed=104, K=32)
    _, ref4, _ = _oracle_grads_uncached(o4, sd4, vox4, AIST, double=True)
    gen = torch.Generator().manual_seed(1)
    sd4p = {k: ((v.double() * (1 + (torch.rand(v.shape, generator=gen, dtype=torch.float64) * 2 - 1) * 6e-8)).float() if v.is_floating_point() else v)
            for k, v in sd4.items()}
    _, g4p, _ = _oracle_grads_uncached(o4, sd4p, vox4, AIST, double=True)
    g4max = max(r.abs().max().item() for r in ref4.values())
    cond104 = max((g4p[k].double() - r.double()).abs().max().item() / max(r.abs().max().item(), 1e-6 * g4max) for k, r in ref4.items())
    l_n, g_n, _ = _hip_grads(o, sd, vox, AIST, mode="fp32")
    os.environ["NM355_GN_DIAG"] = "1"
    try:
        l_d, g_d, _ = _hip_grads(o, sd, vox, AIST, mode="fp32")
    finally:
        del os.environ["NM355_GN_DIAG"]
    noise, d_n, d_d = dist(g_n, g_d), dist(g_n, ref64), dist(g_d, ref64)
    print("seed 103: fp64 gradient under one-ulp weight perturbations moves by %s (seed 104: %.2e); HIP fp32 vs fp64: %.2e and %.2e "
          "(GroupNorm statistics one ulp apart; the two differ by %.2e); losses %.8f / %.8f" % (" / ".join("%.2e" % c for c in cond), cond104, d_n, d_d, noise, l_n, l_d))
    assert abs(l_n - l_d) <= 1e-6 * abs(l_n)                  # the forward is insensitive to the one-ulp difference ...
    assert max(cond) > 3e-4 and cond104 < 3e-5                 # ... this case's gradient jumps, in exact arithmetic (its neighbour's does not)
    assert max(d_n, d_d) <= 6.0 * max(cond)                    # a few such jumps' worth
