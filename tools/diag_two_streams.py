#!/usr/bin/env python3
"""Experiment: does splitting the B = 4 bench batch into two half-batches on two HIP streams (two contexts, two host threads) hide
the launch gaps / small-kernel latency of one behind the kernels of the other?  Prints ms per 64 frames for: one context B = 4;
two contexts B = 2 each, concurrently; two contexts B = 2, one after the other."""
import os, sys, time, threading
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neural_marionette_amd import NeuralMarionette, HotPathOptions, synth
G, T, S = 64, 16, 10
o = HotPathOptions(grid_size=G)
sd = synth.make_state_dict(o, seed=42, variant="peaky")
acts = {"detector": True, "learner": True}
def mk():
    n = NeuralMarionette(o); n.load_state_dict(sd); n = n.cuda().eval(); n.anneal(1); return n
vox = synth.figure_clip(4, T, G, seed=1).cuda(); eps = synth.make_eps((T, S, 4, o.nlatent_kypt), seed=100).cuda()
net = mk()
halves = [(mk(), vox[:2].contiguous(), eps[:, :, :2].contiguous(), torch.cuda.Stream()), (mk(), vox[2:].contiguous(), eps[:, :, 2:].contiguous(), torch.cuda.Stream())]
steps = 10
with torch.no_grad():
    for _ in range(3): net(vox, acts, eps=eps)
    for n, v, e, s in halves:
        with torch.cuda.stream(s):
            for _ in range(3): n(v, acts, eps=e)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps): ref = net(vox, acts, eps=eps)
    torch.cuda.synchronize(); one = (time.perf_counter() - t0) / steps * 1e3
    outs = [None, None]
    def work(i):
        n, v, e, s = halves[i]
        with torch.no_grad(), torch.cuda.stream(s):
            for _ in range(steps): outs[i] = n(v, acts, eps=e)
    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in th: t.start()
    for t in th: t.join()
    torch.cuda.synchronize(); two = (time.perf_counter() - t0) / steps * 1e3
    t0 = time.perf_counter()
    work(0); work(1)
    torch.cuda.synchronize(); seq = (time.perf_counter() - t0) / steps * 1e3
    same = torch.equal(torch.cat([outs[0]["keypoints"], outs[1]["keypoints"]]), ref["keypoints"])
print(f"one context B=4: {one:.3f} ms   two contexts B=2+2 concurrent: {two:.3f} ms   two contexts B=2 then B=2: {seq:.3f} ms   keypoints identical: {same}")
