"""Run-to-run identity of two training steps (G = 32, B = 2, T = 3; two fresh networks in one process).  usage: diag_determinism.py [repeats]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import test_train_detector_gpu as T
from neural_marionette_amd import NeuralMarionette
from neural_marionette_amd.train import DetectorTrainer
o, sd, vox = T._setup(G=32, B=2, T=3, seed=73)
def run():
    net = NeuralMarionette(o); net.load_state_dict(sd); net = net.cuda().train(); net.anneal(1)
    tr = DetectorTrainer(net, lr=4e-4)
    logs = [tr.step(vox.cuda()) for _ in range(2)]
    torch.cuda.synchronize()
    return net, [l["loss"] for l in logs]
ref_net, ref = run()
bad = 0
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    net, l = run()
    diff = [n for (n, p), (_, q) in zip(ref_net.named_parameters(), net.named_parameters()) if not torch.equal(p, q)]
    if l != ref or diff:
        bad += 1
        print("run %d differs: losses %s vs %s; %d tensors, first %s" % (i, l, ref, len(diff), diff[:3]))
print("env", {k: v for k, v in os.environ.items() if k.startswith("NM355")}, "-> %d differing runs" % bad)
