"""Which gradient entries differ between repeated evaluations of the same detector gradient (G = 32, B = 2, T = 3): the pattern of a
cross-stream race.  usage: diag_grad_race.py [repeats]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import test_train_detector_gpu as T
from neural_marionette_amd import NeuralMarionette
o, sd, vox = T._setup(G=32, B=2, T=3, seed=73)
net = NeuralMarionette(o); net.load_state_dict(sd); net = net.cuda().train(); net.anneal(1)
acts = {"detector": True, "learner": False}
net.control_active(acts)
v = vox.cuda()
def grads():
    net.zero_grad()
    out = net(v, acts)
    loss = sum(w * out[k] for k, w in T.AIST.items())
    loss.backward()
    torch.cuda.synchronize()
    g = {n: p.grad.detach().clone() for n, p in net.kypt_detector.named_parameters() if p.grad is not None}
    for k in ("keypoints", "heatmaps", "recon", "first_feature"):          # forward outputs too: a race there shows upstream of every gradient
        g["forward:" + k] = out[k].detach().clone()
    g["forward:losses"] = torch.stack([out[k].detach() for k in T.AIST])
    return g
ref = grads()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
nbad = 0
from neural_marionette_amd import _lib
events = 0
quiet = os.environ.get("DIAG_QUIET", "0") == "1"
for i in range(n):
    try:
        g = grads()
    except _lib.NmError as e:
        events += 1
        continue
    if any(not torch.equal(ref[k], g[k]) for k in ref):
        events += 1
        if quiet:
            continue
    for k in ref:
        if not torch.equal(ref[k], g[k]):
            d = (ref[k] != g[k])
            idx = d.nonzero()
            nbad += 1
            rel = ((ref[k] - g[k]).abs().max() / ref[k].abs().max()).item()
            print("run %d: %s shape %s: %d entries differ, max rel %.2e; first idx %s last idx %s" % (i, k, tuple(ref[k].shape), idx.shape[0], rel, idx[0].tolist(), idx[-1].tolist()))
            if idx.shape[1] == 5:
                print("   distinct co:", sorted(set(idx[:, 0].tolist()))[:40], " distinct ci:", sorted(set(idx[:, 1].tolist()))[:70], " taps:", sorted(set((idx[:, 2] * 9 + idx[:, 3] * 3 + idx[:, 4]).tolist())))
print("evaluations %d: events (an evaluation that differs or raises) %d; differing (run, tensor) pairs %d   [%s]" % (n, events, nbad, " ".join("%s=%s" % (k, v) for k, v in os.environ.items() if k.startswith("NM355_"))))
