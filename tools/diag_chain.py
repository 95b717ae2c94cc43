#!/usr/bin/env python3
"""Persistent rollout kernel (vrnn_prior_chain_kernel) vs the launch-per-phase steps, both contexts alive in ONE process (no stale-memory
coincidences): outputs compared bit for bit on poisoned output tensors, median time per generated step."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neural_marionette_amd import NeuralMarionette, HotPathOptions, synth
from oracle import nm_oracle as O
o = HotPathOptions(grid_size=32, Tcond=5)
sd = synth.make_state_dict(o, seed=21, variant="default")
def mk(chain, graph="1"):
    os.environ["NM355_VRNN_CHAIN"] = chain; os.environ["NM355_VRNN_GRAPH"] = graph
    n = NeuralMarionette(o); n.load_state_dict(sd); n = n.cuda().eval(); n.anneal(1)
    n.dyna_module.get_offset  # noqa
    return n
aff = O.affinity_v3(sd["kypt_detector.affinity_params"]).cuda()
nets = {}
for name, chain, graph in (("launches", "0", "1"), ("chain", "1", "1"), ("chain-eager", "1", "0")):
    nets[name] = mk(chain, graph)
    # force context creation now (the switches are read then)
    nets[name].dyna_module.generate((torch.rand(1, 5, 24, 4) * 1.6 - 0.8).cuda(), aff, Ttot=6, Tcond=5, eps_post=synth.make_eps((5, 10, 1, 128), seed=1).cuda(), eps_prior=synth.make_eps((1, 1, 128), seed=2).cuda())
torch.cuda.synchronize()
K, Z, Tc, Tt = o.nkeypoints, o.nlatent_kypt, 5, 69
for B in (3, 1, 2, 4):
    g = torch.Generator().manual_seed(B)
    kp = (torch.rand(B, Tc, K, 4, generator=g) * 1.6 - 0.8).cuda()
    e_post = synth.make_eps((Tc, 10, B, Z), seed=50).cuda(); e_prior = synth.make_eps((Tt - Tc, B, Z), seed=51).cuda()
    outs = {}
    for name, n in nets.items():
        d = n.dyna_module
        for _ in range(3):
            out = d.generate(kp, aff, Ttot=Tt, Tcond=Tc, eps_post=e_post, eps_prior=e_prior)
        torch.cuda.synchronize()
        ts = []
        for _ in range(20):
            t0 = time.perf_counter()
            out = d.generate(kp, aff, Ttot=Tt, Tcond=Tc, eps_post=e_post, eps_prior=e_prior)
            torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        ts.sort()
        outs[name] = out["keypoints_gen"].clone()
        print("B=%d %-12s 69-step generate median %.3f ms (min %.3f) = %.1f us/step; finite %s; |kp| mean %.4f" %
              (B, name, ts[10] * 1e3, ts[0] * 1e3, ts[10] * 1e6 / Tt, bool(torch.isfinite(outs[name]).all()), float(outs[name].abs().mean())))
    for name in ("chain", "chain-eager"):
        print("   %s bit-identical to launches: %s (max diff %.3e)" % (name, torch.equal(outs[name], outs["launches"]), float((outs[name] - outs["launches"]).abs().max())))
# prior-only rollout entry point (sample_generation's loop): T = 64 from a given state
for B in (1, 3):
    d0, d1 = nets["launches"].dyna_module, nets["chain"].dyna_module
    h = torch.randn(B, 512, generator=torch.Generator().manual_seed(7)).cuda() * 0.1
    kp = (torch.rand(B, Tc, K, 4, generator=torch.Generator().manual_seed(B)) * 1.6 - 0.8).cuda()
    off0, off1 = d0.get_offset(kp), d1.get_offset(kp)
    eps = synth.make_eps((64, B, Z), seed=77).cuda()
    r0 = d0.rollout(h, off0, eps); r1 = d1.rollout(h, off1, eps)
    torch.cuda.synchronize()
    print("rollout B=%d: keypoints identical %s, final state identical %s" % (B, torch.equal(r0[0], r1[0]), torch.equal(r0[1], r1[1])))
    for name, d, off in (("launches", d0, off0), ("chain", d1, off1)):
        ts = []
        for _ in range(20):
            t0 = time.perf_counter(); d.rollout(h, off, eps); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        ts.sort()
        print("   %-9s 64-step rollout median %.3f ms = %.1f us/step" % (name, ts[10] * 1e3, ts[10] * 1e6 / 64))
