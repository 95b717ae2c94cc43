"""Per-call wall time of the headline forward (64^3, T = 16, B = 4): isolated calls (synchronised each) and back-to-back runs of several
lengths - what the host side of a call costs and whether the device clock sags over a run.  usage: time_forward_steps.py"""
import sys, time, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from neural_marionette_amd import NeuralMarionette, HotPathOptions, synth
o = HotPathOptions(grid_size=64)
net = NeuralMarionette(o)
net.load_state_dict(synth.make_state_dict(o, seed=42, variant="peaky"))
net = net.cuda().eval(); net.anneal(1)
vox = synth.figure_clip(4, 16, 64, seed=77).cuda()
eps = synth.make_eps((16, 10, 4, o.nlatent_kypt), seed=78).cuda()
acts = {"detector": True, "learner": True}
def fwd():
    with torch.no_grad():
        return net(vox, acts, eps=eps)
for _ in range(5): fwd()
torch.cuda.synchronize()
ts = []
for i in range(10):
    torch.cuda.synchronize(); t0 = time.perf_counter(); fwd(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    ts.append(((t1 - t0) * 1e3, (t2 - t0) * 1e3))
print("isolated calls: host return after %s ms, done after %s ms" % (" ".join("%.2f" % a for a, _ in ts), " ".join("%.2f" % b for _, b in ts)))
for n in (5, 10, 20, 50, 100):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n): fwd()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("%3d calls back to back: %.3f ms per call (host loop returned after %.3f ms per call)" % (n, (t2 - t0) * 1e3 / n, (t1 - t0) * 1e3 / n))
