#!/usr/bin/env python3
"""Times NeuralMarionette.sample_interpolation at the demo's S = 10 000 rows (vis_interpolation.py:91-143).
NM355_VRNN_GEMM=0 keeps the one-wavefront-per-row kernels for every batch size (A/B)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neural_marionette_amd import NeuralMarionette, HotPathOptions, synth
o = HotPathOptions(grid_size=64)
sd = synth.make_state_dict(o, seed=29, variant="peaky")
net = NeuralMarionette(o); net.load_state_dict(sd); net = net.cuda().eval(); net.anneal(1)
T, S = 21, 10000
vox = synth.figure_clip(1, T, 64, seed=8)[0].cuda()
ea, eb = synth.make_eps((T, S, 128), 9).cuda(), synth.make_eps((T, S, 128), 10).cuda()
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = net.sample_interpolation(vox, sample_rate=10, sample_num=S, eps_a=ea, eps_b=eb)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("GEMM=%s  sample_interpolation T=%d S=%d: %.1f ms total, %.2f ms per frame (detector + %d VRNN steps + decode); picks %s" % (
    os.environ.get("NM355_VRNN_GEMM", "1"), T, S, dt * 1e3, dt * 1e3 / T, T, out["picks"]))
