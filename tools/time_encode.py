#!/usr/bin/env python3
"""VRNN encode alone on an idle device (the posterior steps of the forward run beside the decoder): ms per call and us per
timestep for B = 1 and 4, T = 16, S = 10 (the bench shape is B = 4).  usage: time_encode.py"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neural_marionette_amd import NeuralMarionette, HotPathOptions, synth
from oracle import nm_oracle as O
o = HotPathOptions(grid_size=64)
sd = synth.make_state_dict(o, seed=21, variant="default")
net = NeuralMarionette(o); net.load_state_dict(sd); net = net.cuda().eval(); net.anneal(1)
d = net.dyna_module
aff = O.affinity_v3(sd["kypt_detector.affinity_params"]).cuda()
K, Z, T, S = o.nkeypoints, o.nlatent_kypt, 16, 10
for B in (1, 4):
    g = torch.Generator().manual_seed(B)
    kp = (torch.rand(B, T, K, 4, generator=g) * 1.6 - 0.8).cuda()
    eps = synth.make_eps((T, S, B, Z), seed=50).cuda()
    with torch.no_grad():
        for _ in range(3): d.encode(kp, aff, SAMPLE_NUM=S, eps=eps)
        torch.cuda.synchronize(); ts = []
        for _ in range(20):
            t0 = time.perf_counter(); d.encode(kp, aff, SAMPLE_NUM=S, eps=eps); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    ts.sort()
    print("B=%d: encode median %.3f ms = %.1f us per timestep (min %.1f)" % (B, ts[10] * 1e3, ts[10] * 1e6 / T, ts[0] * 1e6 / T))
