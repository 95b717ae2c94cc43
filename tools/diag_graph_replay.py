"""Diagnostic: the fused forward captured into a HIP graph through torch.cuda.graph (both library streams join the capture through
their events) and replayed, against the plain call sequence.  python tools/diag_graph_replay.py [steps]"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from neural_marionette_amd import NeuralMarionette, HotPathOptions, synth

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
G, T, B, S = 64, 16, 4, 10
o = HotPathOptions(grid_size=G)
net = NeuralMarionette(o); net.load_state_dict(synth.make_state_dict(o, seed=42, variant="peaky")); net = net.cuda().eval(); net.anneal(1)
acts = {"detector": True, "learner": True}
vox = synth.figure_clip(B, T, G, seed=1).cuda(); eps = synth.make_eps((T, S, B, o.nlatent_kypt), seed=100).cuda()
with torch.no_grad():
    for _ in range(3): ref = net(vox, acts, eps=eps)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps): net(vox, acts, eps=eps)
    torch.cuda.synchronize()
    plain = (time.perf_counter() - t0) / steps * 1e3
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        net(vox, acts, eps=eps)                      # warm-up on the capture stream (binds the ctx to it)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        out = net(vox, acts, eps=eps)
    torch.cuda.synchronize()
    for _ in range(3): g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps): g.replay()
    torch.cuda.synchronize()
    graph = (time.perf_counter() - t0) / steps * 1e3
    e = (out["keypoints"] - ref["keypoints"]).abs().max().item()
print(f"plain {plain:.3f} ms/step   graph replay {graph:.3f} ms/step   keypoints max diff {e:.1e}")
