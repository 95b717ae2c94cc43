#!/usr/bin/env python3
"""Stand-alone HSVRNNBVH.encode, six launches per posterior step (NM355_VRNN_POST_CHAIN=0) against the persistent posterior chain (=1),
two contexts in one process, and the fused forward (config 2) with the chain off / forced on inside it (=2).  usage: time_encode_ab.py"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neural_marionette_amd import NeuralMarionette, HotPathOptions, synth
from oracle import nm_oracle as O
o = HotPathOptions(grid_size=64)
sd = synth.make_state_dict(o, seed=42, variant="peaky")
nets = {}
for sw in ("0", "1", "2"):
    os.environ["NM355_VRNN_POST_CHAIN"] = sw
    n = NeuralMarionette(o); n.load_state_dict(sd); n = n.cuda().eval(); n.anneal(1); n.set_conv_mode("split16")
    with torch.no_grad(): n.kypt_detector.get_affinity()
    nets[sw] = n
del os.environ["NM355_VRNN_POST_CHAIN"]
aff = O.affinity_v3(sd["kypt_detector.affinity_params"]).cuda()
K, Z, T, S = o.nkeypoints, o.nlatent_kypt, 16, 10
for B in (1, 4):
    g = torch.Generator().manual_seed(B)
    kp = (torch.rand(B, T, K, 4, generator=g) * 1.6 - 0.8).cuda()
    eps = synth.make_eps((T, S, B, Z), seed=50).cuda()
    for sw in ("0", "1"):
        d = nets[sw].dyna_module
        with torch.no_grad():
            for _ in range(3): d.encode(kp, aff, SAMPLE_NUM=S, eps=eps)
            torch.cuda.synchronize(); ts = []
            for _ in range(30):
                t0 = time.perf_counter(); d.encode(kp, aff, SAMPLE_NUM=S, eps=eps); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        ts.sort()
        print("encode alone B=%d T=%d S=%d, %s: median %.3f ms = %.1f us per timestep (min %.1f)" % (
            B, T, S, "six launches per step" if sw == "0" else "persistent chain     ", ts[15] * 1e3, ts[15] * 1e6 / T, ts[0] * 1e6 / T))
vox = synth.figure_clip(4, 16, 64, seed=1).cuda()
eps = synth.make_eps((16, 10, 4, Z), seed=100).cuda()
acts = {"detector": True, "learner": True}
for rep in range(2):
    for sw in ("0", "2"):
        n = nets[sw]
        with torch.no_grad():
            for _ in range(3): n(vox, acts, eps=eps)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(20): n(vox, acts, eps=eps)
            torch.cuda.synchronize()
        print("fused forward (config 2), encode %s: %.3f ms per step" % ("as launches beside the decoder" if sw == "0" else "as the persistent chain       ", (time.perf_counter() - t0) / 20 * 1e3))
