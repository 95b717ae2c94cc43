#!/usr/bin/env python3
"""Localise the K = 32 / seed 103 gradient deviation: per loss family, HIP exact-fp32 gradients vs the fp64 oracle."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import test_train_detector_gpu as T
o, sd, vox = T._setup(G=32, B=1, T=3, seed=int(os.environ.get("SEED", "103")), K=32)
fams = dict(T.WEIGHTINGS)
fams.update(local=dict(local_const_loss=1.0), time=dict(time_const_loss=1.0), sparsity_const=dict(sparsity_const_loss=1.0))
for name, w in fams.items():
    _, ref, rout = T._oracle_grads(o, sd, vox, w, double=True)
    _, got, out = T._hip_grads(o, sd, vox, w, mode="fp32")
    gmax = max(r.abs().max().item() for r in ref.values())
    worst = ("", 0.0)
    for k, r in ref.items():
        scale = max(r.abs().max().item(), 1e-6 * gmax, 1e-30)
        e = (got[k].double() - r.double()).abs().max().item() / scale
        if e > worst[1]: worst = (k, e)
    print("%-14s worst %.2e at %s  (|g|max %.2e)  loss hip %.6e ref %.6e" % (name, worst[1], worst[0], gmax, sum(float(out[k]) * v for k, v in w.items()), sum(float(rout[k]) * v for k, v in w.items())))
    if name in ("graph", "local", "time", "sparsity_const", "traj"):
        k = "kypt_detector.affinity_params"
        d = (got[k].double() - ref[k].double()).abs()
        i = int(d.argmax()); print("     affinity_params grad: max abs diff %.3e at flat %d, ref there %.3e, got %.3e" % (d.max().item(), i, ref[k].flatten()[i].item(), got[k].flatten()[i].item()))
