"""Rare non-finite reconstruction of the TRAINING forward (found by tests/test_train_detector_gpu.py::test_gradient_is_bit_identical_over_many_evaluations:
one evaluation in ~1000): counts the events over many forward calls (G = 32, B = 2, T = 3).  usage: diag_forward_nan.py [calls] (switches from the environment)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import test_train_detector_gpu as T
from neural_marionette_amd import NeuralMarionette, _lib
o, sd, vox = T._setup(G=32, B=2, T=3, seed=73)
train = os.environ.get("DIAG_EVAL", "0") != "1"
net = NeuralMarionette(o); net.load_state_dict(sd); net = net.cuda(); net = net.train() if train else net.eval(); net.anneal(1)
acts = {"detector": True, "learner": False}
net.control_active(acts)
v = vox.cuda()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
ref = None; bad = 0; diff = 0; err = 0
for i in range(n):
    try:
        if train and os.environ.get("DIAG_BACKWARD", "0") == "1":
            net.zero_grad()
            out = net(v, acts)
            sum(w * out[k] for k, w in T.AIST.items()).backward()
        elif train:
            out = net(v, acts)                  # (grad mode: nm_detector_forward_train)
        else:
            with torch.no_grad():
                out = net.kypt_detector(v)
        r = out["recon"].detach()
        torch.cuda.synchronize()
    except _lib.NmError as e:
        err += 1
        continue
    if ref is None: ref = r.clone()
    if not torch.isfinite(r).all(): bad += 1; print("call %d: %d non-finite recon entries, frames %s" % (i, int((~torch.isfinite(r)).sum()), sorted(set((~torch.isfinite(r)).nonzero()[:, :2].flatten().tolist()))[:8]), flush=True)
    elif not torch.equal(r, ref): diff += 1
print("calls %d (%s): non-finite %d, finite but different %d, range-guard errors %d   [%s]" % (n, "train forward" if train else "inference forward", bad, diff, err,
      " ".join("%s=%s" % (k, v) for k, v in os.environ.items() if k.startswith("NM355_"))))
