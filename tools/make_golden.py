#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE (imported from /root/reference)
on the build's seeded synthetic weights and inputs.

Runs only in the build container (the reference tree does not travel to the GPU
box).  Fixtures hold seeds + expected outputs of the reference itself; inputs and
weights are re-created from the seeds by ``neural_marionette_amd.synth``.

  python tools/make_golden.py            # writes every case
  python tools/make_golden.py g2 g3      # subset

The VRNN noise is injected by replacing ``torch.distributions.normal._standard_normal``
(the name ``Normal.rsample`` resolves at call time).
"""
from __future__ import annotations

import os
import pickle
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)

from neural_marionette_amd.spec import HotPathOptions, DETECTOR_LOSS_KEYS  # noqa: E402
from neural_marionette_amd import synth  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
S = 10


def _ref_opt(G):
    opt = pickle.load(open(os.path.join(REF, "pretrained/aist/opt.pickle"), "rb"))
    opt.grid_size = G
    return opt


def _ref_net(opt, sd):
    from model.neural_marionette import NeuralMarionette
    net = NeuralMarionette(opt).eval()
    net.load_state_dict(sd)
    net.anneal(1)           # switches the affinity graph on (kypt_detector.py:71-78)
    return net


class EpsFeed:
    """Feeds pre-drawn eps tensors to Normal.rsample in call order."""

    def __init__(self, chunks):
        self.chunks = list(chunks)
        self.i = 0

    def __call__(self, shape, dtype, device):
        e = self.chunks[self.i]
        self.i += 1
        assert tuple(shape) == tuple(e.shape), (tuple(shape), tuple(e.shape))
        return e.clone()

    def __enter__(self):
        import torch.distributions.normal as tdn
        self._tdn, self._old = tdn, tdn._standard_normal
        tdn._standard_normal = self
        return self

    def __exit__(self, *a):
        self._tdn._standard_normal = self._old


def _np(x):
    return x.detach().cpu().numpy()


def _sub(x, step):
    """strided spatial sub-sample of the last three dims"""
    return _np(x[..., ::step, ::step, ::step])


def case_g1():
    """Config 1 of BASELINE.json: 64^3, B=1, T=4, detector forward only."""
    G, B, T, wseed, iseed = 64, 1, 4, 11, 21
    opt = _ref_opt(G)
    o = HotPathOptions.from_any(opt)
    sd = synth.make_state_dict(o, seed=wseed, variant="peaky")
    net = _ref_net(opt, sd)
    vox = synth.figure_clip(B, T, G, seed=iseed)
    with torch.no_grad():
        r = net.kypt_detector(vox)
    np.savez_compressed(
        os.path.join(OUT, "g1_detector64.npz"),
        meta=np.array([G, B, T, wseed, iseed]), variant="peaky", clip="figure",
        keypoints=_np(r["keypoints"]),
        losses=np.array([float(r[k]) for k in DETECTOR_LOSS_KEYS], dtype=np.float64),
        heatmaps_sub=_sub(r["heatmaps"], 2), heatmaps_sum=_np(r["heatmaps"].sum(dim=(3, 4, 5))),
        first_feature_sub=_sub(r["first_feature"], 2),
        first_feature_sum=_np(r["first_feature"].double().sum(dim=(2, 3, 4))),
        recon_sub=_sub(r["recon"], 4), recon_sum=_np(r["recon"].double().sum(dim=(2, 3, 4, 5))),
        recon_occ=_np((r["recon"] >= 0.5).sum(dim=(2, 3, 4, 5))),
        recon_margin=np.array(float((r["recon"] - 0.5).abs().min())),
        affinity=_np(r["affinity"]),
    )
    print("g1 ok", float(r["recon_loss"]))


def case_g2():
    """32^3 B=2 T=4 full forward (detector + VRNN.encode) with recorded eps."""
    G, B, T, wseed, iseed, eseed = 32, 2, 4, 3, 5, 9
    opt = _ref_opt(G)
    o = HotPathOptions.from_any(opt)
    sd = synth.make_state_dict(o, seed=wseed, variant="peaky")
    net = _ref_net(opt, sd)
    vox = synth.figure_clip(B, T, G, seed=iseed)
    eps = synth.make_eps((T, S, B, o.nlatent_kypt), seed=eseed)
    with torch.no_grad(), EpsFeed(eps):
        r = net(vox, {"detector": True, "learner": True})
    d = net.dyna_module
    np.savez_compressed(
        os.path.join(OUT, "g2_forward32.npz"),
        meta=np.array([G, B, T, wseed, iseed, eseed]), variant="peaky", clip="figure",
        keypoints=_np(r["keypoints"]), heatmaps=_np(r["heatmaps"]),
        first_feature=_np(r["first_feature"]),
        recon_sub=_sub(r["recon"], 2), recon_sum=_np(r["recon"].double().sum(dim=(2, 3, 4, 5))),
        recon_occ=_np((r["recon"] >= 0.5).sum(dim=(2, 3, 4, 5))),
        losses=np.array([float(r[k]) for k in DETECTOR_LOSS_KEYS], dtype=np.float64),
        affinity=_np(r["affinity"]),
        kypt_recon=_np(r["kypt_recon"]), R=_np(r["R"]), z_kypts=_np(r["z_kypts"]),
        h_kypts=_np(r["h_kypts"]), kl_kypt=np.array(float(r["kl_kypt"])),
        kypt_recon_loss=np.array(float(r["kypt_recon_loss"])),
        parents=_np(d.parents), order=_np(d.priority.indices), order_values=_np(d.priority.values),
        A=_np(d.A),
    )
    print("g2 ok", float(r["kl_kypt"]))


def case_g3():
    """Skeleton trees: affinity (N,K,K,1) -> (A, parents, priority) for several seeds."""
    from utils.dyna_utils import process_affinity_glob
    K, N = 24, 2

    def affinity_input(params):           # an (N,K,K,1) row-stochastic affinity with a zero diagonal (the fixture stores it)
        P = torch.softmax(params, dim=-1)
        W = torch.zeros(N, K, K, dtype=params.dtype)
        for k in range(K):
            W[:, k, :k] = P[:, k, :k]
            W[:, k, k + 1:] = P[:, k, k:]
        return W[..., None]

    affs, As, pars, ords, vals = [], [], [], [], []
    for seed in range(12):
        rng = np.random.default_rng([seed, 0x73EE])
        if seed == 0:
            params = torch.ones(N, K, K - 1)            # the all-ties init of the reference
        elif seed < 4:
            params = torch.from_numpy(np.round(rng.standard_normal((N, K, K - 1)) * 2) / 2).float()  # many ties
        else:
            params = torch.from_numpy(rng.standard_normal((N, K, K - 1)) * (1 + seed % 3)).float()
        aff = affinity_input(params)
        A, pri, par = process_affinity_glob(aff)
        affs.append(_np(aff)); As.append(_np(A)); pars.append(_np(par))
        ords.append(_np(pri.indices)); vals.append(_np(pri.values))
    np.savez_compressed(os.path.join(OUT, "g3_trees.npz"), affinity=np.stack(affs), A=np.stack(As),
                        parents=np.stack(pars), order=np.stack(ords), order_values=np.stack(vals))
    print("g3 ok")


def case_g4():
    """generate(): 32^3, B=2, Tcond=3, Ttot=8 with recorded eps."""
    G, B, T, Tc, wseed, iseed, eseed = 32, 2, 8, 3, 4, 6, 10
    opt = _ref_opt(G)
    opt.Tcond = Tc
    o = HotPathOptions.from_any(opt)
    sd = synth.make_state_dict(o, seed=wseed, variant="peaky")
    net = _ref_net(opt, sd)
    vox = synth.figure_clip(B, T, G, seed=iseed)
    Z = o.nlatent_kypt
    e_enc = synth.make_eps((Tc, S, B, Z), seed=eseed)
    e_post = synth.make_eps((Tc, S, B, Z), seed=eseed + 1)
    e_prior = synth.make_eps((T - Tc, B, Z), seed=eseed + 2)
    acts = {"detector": True, "learner": True}
    with torch.no_grad():
        with EpsFeed(e_enc):        # a first encode builds the tree (generate needs .parents)
            net(vox[:, :Tc].contiguous(), acts)
        with EpsFeed(list(e_post) + list(e_prior)):
            r = net.generate(vox, acts)
    d = net.dyna_module
    np.savez_compressed(
        os.path.join(OUT, "g4_generate32.npz"),
        meta=np.array([G, B, T, Tc, wseed, iseed, eseed]), variant="peaky", clip="figure",
        keypoints=_np(r["keypoints"]), gen_sub=_sub(r["gen"], 2),
        gen_sum=_np(r["gen"].double().sum(dim=(2, 3, 4, 5))),
        gen_occ=_np((r["gen"] >= 0.5).sum(dim=(2, 3, 4, 5))),
        parents=_np(d.parents), order=_np(d.priority.indices),
    )
    print("g4 ok")


def case_g5():
    """Odd-size plumbing (output_padding path of the hourglass): 40^3, B=1, T=3 detector."""
    G, B, T, wseed, iseed = 40, 1, 3, 7, 8
    opt = _ref_opt(G)
    o = HotPathOptions.from_any(opt)
    sd = synth.make_state_dict(o, seed=wseed, variant="peaky")
    net = _ref_net(opt, sd)
    vox = synth.figure_clip(B, T, G, seed=iseed)
    with torch.no_grad():
        r = net.kypt_detector(vox)
    np.savez_compressed(
        os.path.join(OUT, "g5_detector40.npz"),
        meta=np.array([G, B, T, wseed, iseed]), variant="peaky", clip="figure",
        keypoints=_np(r["keypoints"]), heatmaps=_np(r["heatmaps"]),
        losses=np.array([float(r[k]) for k in DETECTOR_LOSS_KEYS], dtype=np.float64),
        recon_sub=_sub(r["recon"], 2), recon_sum=_np(r["recon"].double().sum(dim=(2, 3, 4, 5))),
    )
    print("g5 ok")


def case_g6():
    """Learner-mode training gradients (pretrained_mode=1: detector frozen, train.py:146,177-181):
    d(1.0 * kypt_recon_loss + 0.003 * kl_kypt) / d(dyna_module parameters) by the reference's autograd,
    B=3, T=5, recorded eps.  Every gradient is stored as (sum, abs-sum, every 97th element)."""
    B, T, wseed, kseed, eseed = 3, 5, 13, 14, 15
    opt = _ref_opt(32)
    o = HotPathOptions.from_any(opt)
    sd = synth.make_state_dict(o, seed=wseed, variant="default")
    net = _ref_net(opt, sd)
    g = torch.Generator().manual_seed(kseed)
    kp = torch.rand(B, T, o.nkeypoints, 4, generator=g) * torch.tensor([1.6, 1.6, 1.6, 1.0]) - torch.tensor([0.8, 0.8, 0.8, 0.0])
    eps = synth.make_eps((T, S, B, o.nlatent_kypt), seed=eseed)
    aff = net.kypt_detector.get_affinity().detach()
    for p in net.parameters():
        p.grad = None
    with EpsFeed(eps):
        r = net.dyna_module.encode(kp, aff)
    loss = 1.0 * r["kypt_recon_loss"] + 0.003 * r["kl_kypt"]
    loss.backward()
    out = dict(meta=np.array([B, T, wseed, kseed, eseed]), loss=np.array(float(loss)),
               kypt_recon_loss=np.array(float(r["kypt_recon_loss"])), kl_kypt=np.array(float(r["kl_kypt"])),
               parents=_np(net.dyna_module.parents), order=_np(net.dyna_module.priority.indices))
    for name, p in net.dyna_module.named_parameters():
        if p.grad is None:
            continue
        gflat = p.grad.reshape(-1).double()
        out["g:" + name] = np.concatenate([[gflat.sum().item(), gflat.abs().sum().item()], gflat[::97].numpy()])
    np.savez_compressed(os.path.join(OUT, "g6_learner_grads.npz"), **out)
    print("g6 ok", float(loss), sorted(k for k in out if k.startswith("g:"))[:3])


def case_g7():
    """utils/eval_utils.py on seeded inputs: voxel_chamfer_distance and semantic_scores of the reference."""
    from utils import eval_utils as ref_eval
    vox, recon, kp, gt = synth.eval_inputs()
    ch = ref_eval.voxel_chamfer_distance(None, dict(voxel=vox.clone(), recon=recon.clone()))
    se = ref_eval.semantic_scores(None, dict(keypoints=kp.clone(), gt_keypoints=gt.clone()))
    np.savez_compressed(os.path.join(OUT, "g7_eval_metrics.npz"), seed=7, B=2, T=3, G=32, K=24, Kg=17,
                        recon_checksum=np.float64(recon.double().sum().item()),
                        chamfer_scores=np.array(ch["scores"], dtype=np.float64), chamfer_log=np.float64(ch["scores_log"]),
                        semantic_scores=np.array(se["scores"], dtype=np.int64), semantic_log=np.float32(se["scores_log"]))
    print("g7 ok", ch["scores_log"], se["scores_log"])


def case_g8():
    """Detector-mode training gradients (pretrained_mode=0, train.py:270-276,388-404): d(sum_k w_k * loss_k) / d(kypt_detector
    parameters) by the reference's autograd with the AIST loss weights (train.py:177-181), 32^3, B=2, T=4, 'peaky' weights and
    random affinity logits.  Every gradient is stored as (sum, abs-sum, max-abs, every 997th element)."""
    G, B, T, seed = 32, 2, 4, 11
    opt = _ref_opt(G)
    o = HotPathOptions.from_any(opt)
    sd = synth.make_state_dict(o, seed=seed, variant="peaky")
    gen = torch.Generator().manual_seed(seed + 1)
    sd["kypt_detector.affinity_params"] = torch.randn(sd["kypt_detector.affinity_params"].shape, generator=gen)
    vox = synth.figure_clip(B, T, G, seed=seed + 2)
    net = _ref_net(opt, sd).train()
    acts = {"detector": True, "learner": False}
    net.control_active(acts)
    for p in net.parameters():
        p.grad = None
    log = net(vox, acts)
    w = dict(recon_loss=opt.recon_weight, sparsity_loss=opt.sparse_weight, separation_loss=opt.sep_weight, vol_fit_reg=opt.vol_reg_weight,
             kypt_const_loss=opt.kypt_const_weight, local_const_loss=opt.local_const_weight, time_const_loss=opt.time_const_weight,
             sparsity_const_loss=opt.sparsity_const_weight, intensity_const_loss=opt.intensity_const_weight,
             graph_traj_loss=opt.graph_traj_weight, graph_vol_loss=opt.graph_vol_weight)
    loss = sum(float(w[k]) * log[k] for k in DETECTOR_LOSS_KEYS)
    loss.backward()
    out = dict(meta=np.array([G, B, T, seed]), loss=np.array(float(loss)), weights=np.array([float(w[k]) for k in DETECTOR_LOSS_KEYS]),
               losses=np.array([float(log[k]) for k in DETECTOR_LOSS_KEYS]))
    n = 0
    for name, p in net.kypt_detector.named_parameters():
        assert p.grad is not None, name
        gflat = p.grad.reshape(-1).double()
        out["g:" + name] = np.concatenate([[gflat.sum().item(), gflat.abs().sum().item(), gflat.abs().max().item()], gflat[::997].numpy()])
        n += 1
    np.savez_compressed(os.path.join(OUT, "g8_detector_grads.npz"), **out)
    print("g8 ok", float(loss), n, "tensors")


CASES = dict(g1=case_g1, g2=case_g2, g3=case_g3, g4=case_g4, g5=case_g5, g6=case_g6, g7=case_g7, g8=case_g8)

if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)
    for name in (sys.argv[1:] or list(CASES)):
        CASES[name]()
