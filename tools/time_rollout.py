#!/usr/bin/env python3
"""BASELINE config 5: prior rollout latency (Tcond = 5 posterior steps + 64 prior steps), B = 1 and 3.
NM355_VRNN_MID=0 runs a prior step as six dependent launches, 1 (default) as three (vrnn_prior_mid_kernel);
NM355_VRNN_GRAPH=0 enqueues the launches one by one, 1 (default) replays the captured HIP graph.
usage: time_rollout.py <out.pt> [compare.pt]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neural_marionette_amd import NeuralMarionette, HotPathOptions, synth
from oracle import nm_oracle as O
o = HotPathOptions(grid_size=32, Tcond=5)
sd = synth.make_state_dict(o, seed=21, variant="default")
net = NeuralMarionette(o); net.load_state_dict(sd); net = net.cuda().eval(); net.anneal(1)
res = {}
for B in (1, 3):
    K, Z, Tc, Tt = o.nkeypoints, o.nlatent_kypt, 5, 69
    g = torch.Generator().manual_seed(B)
    kp = torch.rand(B, Tc, K, 4, generator=g) * 1.6 - 0.8
    e_post = synth.make_eps((Tc, 10, B, Z), seed=50).cuda(); e_prior = synth.make_eps((Tt - Tc, B, Z), seed=51).cuda()
    aff = O.affinity_v3(sd["kypt_detector.affinity_params"]).cuda()
    d = net.dyna_module
    kpc = kp.cuda()
    for _ in range(3):
        out = d.generate(kpc, aff, Ttot=Tt, Tcond=Tc, eps_post=e_post, eps_prior=e_prior)
    torch.cuda.synchronize()
    ts = []
    for _ in range(20):
        t0 = time.perf_counter()
        out = d.generate(kpc, aff, Ttot=Tt, Tcond=Tc, eps_post=e_post, eps_prior=e_prior)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    ts.sort()
    print("MID=%s GRAPH=%s B=%d: 69-step generate median %.3f ms = %.1f us/step (min %.1f)" % (os.environ.get("NM355_VRNN_MID", "1"), os.environ.get("NM355_VRNN_GRAPH", "1"), B, ts[10] * 1e3, ts[10] * 1e6 / Tt, ts[0] * 1e6 / Tt))
    res[B] = out["keypoints_gen"].cpu()
torch.save(res, sys.argv[1])
if len(sys.argv) > 2:
    ref = torch.load(sys.argv[2])
    print("bit-identical to %s:" % sys.argv[2], all(torch.equal(res[b], ref[b]) for b in res), "max diff", max(float((res[b] - ref[b]).abs().max()) for b in res))
