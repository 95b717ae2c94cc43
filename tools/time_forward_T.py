#!/usr/bin/env python3
"""Forward time as a function of the clip length T at B = 4 (64^3): time = a + b T.  The per-clip part a holds the clip-mean
(spatio-temporal) feature net - 4 frames of wide layers on the side stream - and fixed overheads; b is the per-frame encoder + decoder."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neural_marionette_amd import NeuralMarionette, HotPathOptions, synth
o = HotPathOptions(grid_size=64)
net = NeuralMarionette(o); net.load_state_dict(synth.make_state_dict(o, seed=42, variant="peaky")); net = net.cuda().eval(); net.anneal(1)
acts = {"detector": True, "learner": True}
res = {}
for T in (4, 8, 16, 32):
    vox = synth.figure_clip(4, T, 64, seed=1).cuda(); eps = synth.make_eps((T, 10, 4, o.nlatent_kypt), seed=2).cuda()
    with torch.no_grad():
        for _ in range(3): net(vox, acts, eps=eps)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(8): net(vox, acts, eps=eps)
        torch.cuda.synchronize(); res[T] = (time.perf_counter() - t0) / 8 * 1e3
    print("T=%2d: %.3f ms per forward (%.3f ms per frame-batch of 4)" % (T, res[T], res[T] / T))
b = (res[32] - res[8]) / 24; a = res[16] - 16 * b
print("fit: per-frame-batch b = %.3f ms, per-call a = %.3f ms (clip-mean net + fixed costs)" % (b, a))
