#!/usr/bin/env python3
"""The VRNN row kernels by batch size (nm_vrnn_gru = one linear_rows launch for W_hh h + one gru_rows launch), idle device, run under
rocprofv3 --kernel-trace --stats to read the per-kernel durations.  usage: time_vrnn_rows.py"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neural_marionette_amd import NeuralMarionette, HotPathOptions, synth, _lib
o = HotPathOptions(grid_size=32)
sd = synth.make_state_dict(o, seed=21, variant="default")
net = NeuralMarionette(o); net.load_state_dict(sd); net = net.cuda().eval(); net.anneal(1)
eng = net.dyna_module._eng(); ctx = eng.ready()
K, Z, H = o.nkeypoints, o.nlatent_kypt, o.nhidden_kypt
for B in (1, 2, 4, 8, 16):
    x = torch.randn(B, K * 4 + Z, device="cuda"); h = torch.randn(B, H, device="cuda"); out = torch.empty(B, H, device="cuda")
    for _ in range(5): eng.call("nm_vrnn_gru", _lib.ptr(x), _lib.ptr(h), B, _lib.ptr(out))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(200): eng.call("nm_vrnn_gru", _lib.ptr(x), _lib.ptr(h), B, _lib.ptr(out))
    torch.cuda.synchronize()
    print("B=%2d: %.1f us per nm_vrnn_gru (two launches)" % (B, (time.perf_counter() - t0) / 200 * 1e6))
