#!/usr/bin/env python3
"""Forward step of the bench shape with and without the VRNN encode (learner) part: how much of the step is the second stream's
contention with the detector's persistent conv kernels."""
import sys, os, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from neural_marionette_amd import NeuralMarionette, HotPathOptions, synth
G, T, B, S = 64, 16, 4, 10
opts = HotPathOptions(grid_size=G)
sd = synth.make_state_dict(opts, seed=42, variant="peaky")
net = NeuralMarionette(opts); net.load_state_dict(sd); net = net.cuda().eval(); net.anneal(1)
vox = synth.figure_clip(B, T, G, seed=1).cuda()
eps = synth.make_eps((T, S, B, opts.nlatent_kypt), seed=100).cuda()
def run(acts, n=10, w=3):
    with torch.no_grad():
        for _ in range(w): net(vox, acts, eps=eps)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): net(vox, acts, eps=eps)
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
for name, acts in (("detector + learner", {"detector": True, "learner": True}), ("detector only", {"detector": True, "learner": False}),
                   ("detector + learner", {"detector": True, "learner": True})):
    print("%-20s %.2f ms/step" % (name, run(acts)))
