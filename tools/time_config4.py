#!/usr/bin/env python3
"""BASELINE config 4 (96^3, T = 8, B = 2) forward time; environment switches (NM355_*) are read when the context is created."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neural_marionette_amd import NeuralMarionette, HotPathOptions, synth
G = int(sys.argv[1]) if len(sys.argv) > 1 else 96
B, T = (2, 8) if G == 96 else (4, 16)
o = HotPathOptions(grid_size=G)
net = NeuralMarionette(o); net.load_state_dict(synth.make_state_dict(o, seed=7, variant="peaky")); net = net.cuda().eval(); net.anneal(1)
vox = synth.figure_clip(B, T, G, seed=3).cuda(); eps = synth.make_eps((T, 10, B, o.nlatent_kypt), seed=4).cuda()
acts = {"detector": True, "learner": True}
with torch.no_grad():
    for _ in range(3): net(vox, acts, eps=eps)
    torch.cuda.synchronize(); ts = []
    for _ in range(10):
        t0 = time.perf_counter(); net(vox, acts, eps=eps); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
ts.sort()
print("G=%d B=%d T=%d: median %.3f ms (min %.3f)  %s" % (G, B, T, ts[5] * 1e3, ts[0] * 1e3, " ".join(k + "=" + v for k, v in os.environ.items() if k.startswith("NM355_"))))
