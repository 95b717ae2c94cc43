#!/usr/bin/env python3
"""Diagnostic: where a step of the persistent rollout kernel spends its time.  Phase stamps (s_memrealtime, 10 ns) of thread 0 of worker
workgroup 0 and of the first middle workgroup, averaged over steps 8..63 of a 64-step rollout at B = 1 / 3.
usage: NM355_VRNN_GRAPH=0 [NM355_CHAIN_XCD=0|1] python3 tools/diag_chain_stamps.py"""
import ctypes as C, os, sys
import numpy as np, torch
os.environ.setdefault("NM355_VRNN_GRAPH", "0")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neural_marionette_amd import NeuralMarionette, HotPathOptions, synth, _lib
from oracle import nm_oracle as O
raw = C.CDLL(_lib.LIB_PATH)
o = HotPathOptions(grid_size=32, Tcond=5)
sd = synth.make_state_dict(o, seed=21, variant="default")
net = NeuralMarionette(o); net.load_state_dict(sd); net = net.cuda().eval(); net.anneal(1)
aff = O.affinity_v3(sd["kypt_detector.affinity_params"]).cuda()
d = net.dyna_module
for B in (1, 3):
    K, Z, Tc, Tt = o.nkeypoints, o.nlatent_kypt, 5, 69
    kp = (torch.rand(B, Tc, K, 4, generator=torch.Generator().manual_seed(B)) * 1.6 - 0.8).cuda()
    e_post = synth.make_eps((Tc, 10, B, Z), seed=50).cuda(); e_prior = synth.make_eps((Tt - Tc, B, Z), seed=51).cuda()
    for _ in range(3):
        d.generate(kp, aff, Ttot=Tt, Tcond=Tc, eps_post=e_post, eps_prior=e_prior)
    torch.cuda.synchronize()
    stamps = torch.zeros(2 * 128 * 8, dtype=torch.int64, device="cuda")
    raw.nm_diag_set_chain_stamps(C.c_void_p(stamps.data_ptr()))
    d.generate(kp, aff, Ttot=Tt, Tcond=Tc, eps_post=e_post, eps_prior=e_prior)
    torch.cuda.synchronize()
    raw.nm_diag_set_chain_stamps(C.c_void_p(0))
    raw_s = stamps.view(2, 128, 8).cpu().numpy().astype(np.float64)
    clk = (raw_s[1, 63, 7] - raw_s[1, 8, 7]) / ((raw_s[1, 63, 0] - raw_s[1, 8, 0]) * 10e-9) / 1e9
    print("shader clock during the rollout: %.2f GHz (s_memtime ticks per s_memrealtime second)" % clk)
    s = raw_s * 0.01          # us
    w, m = s[0, 8:63], s[1, 8:63]
    step = (s[0, 9:64, 0] - s[0, 8:63, 0]).mean()
    print("B=%d XCD=%s: step %.2f us" % (B, os.environ.get("NM355_CHAIN_XCD", "1"), step))
    print("  worker 0 : wait for h %.2f | h-phase rows (x B) %.2f | wait for keypoints|latent %.2f | GRU units (x B) %.2f | to next step %.2f" % (
        (w[:, 1] - w[:, 0]).mean(), (w[:, 2] - w[:, 1]).mean(), (w[:, 3] - w[:, 2]).mean(), (w[:, 4] - w[:, 3]).mean(), (s[0, 9:64, 0] - w[:, 4]).mean()))
    print("  middle 0 : wait for hid/rh/jh %.2f | A (distribution, z) %.2f | B (decoder hidden) %.2f | C (heads) %.2f | D (kinematics) %.2f | stores %.2f" % (
        (m[:, 1] - m[:, 0]).mean(), (m[:, 2] - m[:, 1]).mean(), (m[:, 3] - m[:, 2]).mean(), (m[:, 4] - m[:, 3]).mean(), (m[:, 5] - m[:, 4]).mean(), (m[:, 6] - m[:, 5]).mean()))
    print("  middle's poll completes %.2f us after worker 0 finished its h-phase rows; worker's poll completes %.2f us after the middle's stores" % (
        (m[:, 1] - w[:, 2]).mean(), (w[:, 3] - m[:, 6]).mean()))
