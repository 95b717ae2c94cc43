"""Run-to-run identity stress: repeated evaluations of the detector gradient / the joint step / the forward in several conv modes and
grid sizes, every result compared bit for bit with the first (a race of a few per cent needs dozens of repeats to show; round 4 found
one in the weight-gradient kernels this way).  usage: stress_identity.py [repeats]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import test_train_detector_gpu as T
from neural_marionette_amd import NeuralMarionette, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 and not sys.argv[1].startswith("--") else 30
total_bad = 0
# --concurrent: a second context runs forwards of another size from a second thread the whole time (its kernels perturb the timing of
# everything checked here; the checked results must not change)
if "--concurrent" in sys.argv:
    import threading
    _stop = False
    def _noise():
        torch.cuda.set_device(0)
        o2, sd2, vox2 = T._setup(G=48, B=2, T=3, seed=99)
        net2 = NeuralMarionette(o2); net2.load_state_dict(sd2); net2 = net2.cuda().eval(); net2.anneal(1)
        v2 = vox2.cuda(); e2 = synth.make_eps((3, 10, 2, o2.nlatent_kypt), seed=3).cuda()
        s2 = torch.cuda.Stream()
        with torch.cuda.stream(s2):
            while not _stop:
                with torch.no_grad():
                    net2(v2, {"detector": True, "learner": True}, eps=e2)
                s2.synchronize()
    _th = threading.Thread(target=_noise, daemon=True); _th.start()

def detector_grads(G, B, Tt, mode, reps, env=None):
    global total_bad
    old = {}
    for k, v in (env or {}).items():
        old[k] = os.environ.get(k); os.environ[k] = v
    try:
        o, sd, vox = T._setup(G=G, B=B, T=Tt, seed=73)
        net = NeuralMarionette(o); net.load_state_dict(sd); net = net.cuda().train(); net.anneal(1)
        net.set_conv_mode(mode)
    finally:
        for k, v in old.items():
            if v is None: os.environ.pop(k, None)
            else: os.environ[k] = v
    acts = {"detector": True, "learner": False}
    net.control_active(acts)
    v = vox.cuda()
    def grads():
        net.zero_grad()
        out = net(v, acts)
        sum(w * out[k] for k, w in T.AIST.items()).backward()
        torch.cuda.synchronize()
        return {k: p.grad.detach().clone() for k, p in net.kypt_detector.named_parameters() if p.grad is not None}
    ref = grads(); bad = 0
    for i in range(reps):
        g = grads()
        d = [k for k in ref if not torch.equal(ref[k], g[k])]
        if d:
            bad += 1
            print("   evaluation %d: %d tensors differ, first %s" % (i, len(d), d[:3]))
    print("detector gradient  G=%d B=%d T=%d mode=%-7s env=%s: %d of %d evaluations differ" % (G, B, Tt, mode, env or {}, bad, reps), flush=True)
    total_bad += bad

def joint_step(G, B, Tt, reps):
    global total_bad
    o, sd, vox = T._setup(G=G, B=B, T=Tt, seed=31)
    acts = {"detector": True, "learner": True}
    v = vox.cuda()
    eps = synth.make_eps((Tt, 10, B, o.nlatent_kypt), seed=5).cuda()
    net = NeuralMarionette(o); net.load_state_dict(sd); net = net.cuda().train(); net.anneal(1)
    net.control_active(acts)
    def grads():
        net.zero_grad()
        out = net(v, acts, eps=eps)
        loss = sum(w * out[k] for k, w in T.AIST.items()) + out.get("kypt_recon_loss", 0) if False else sum(w * out[k] for k, w in T.AIST.items())
        loss = loss + sum(out[k] for k in out if k.endswith("_loss") and k not in T.AIST and torch.is_tensor(out[k]) and out[k].requires_grad)
        loss.backward()
        torch.cuda.synchronize()
        return {k: p.grad.detach().clone() for k, p in net.named_parameters() if p.grad is not None}
    ref = grads(); bad = 0
    for i in range(reps):
        g = grads()
        d = [k for k in ref if not torch.equal(ref[k], g[k])]
        if d:
            bad += 1
            print("   evaluation %d: %d tensors differ, first %s" % (i, len(d), d[:3]))
    print("joint step (detector + learner) G=%d B=%d T=%d: %d of %d evaluations differ (%d gradient tensors)" % (G, B, Tt, bad, reps, len(ref)), flush=True)
    total_bad += bad

def forward(G, B, Tt, mode, reps):
    global total_bad
    o, sd, vox = T._setup(G=G, B=B, T=Tt, seed=11)
    net = NeuralMarionette(o); net.load_state_dict(sd); net = net.cuda().eval(); net.anneal(1); net.set_conv_mode(mode)
    v = vox.cuda(); eps = synth.make_eps((Tt, 10, B, o.nlatent_kypt), seed=5).cuda()
    acts = {"detector": True, "learner": True}
    keys = ("keypoints", "heatmaps", "recon", "z_kypts", "h_kypts", "kypt_recon")
    def f():
        with torch.no_grad():
            out = net(v, acts, eps=eps)
        torch.cuda.synchronize()
        return {k: out[k].detach().clone() for k in keys}
    ref = f(); bad = 0
    for i in range(reps):
        g = f()
        d = [k for k in ref if not torch.equal(ref[k], g[k])]
        if d:
            bad += 1; print("   evaluation %d differs in %s" % (i, d))
    print("forward            G=%d B=%d T=%d mode=%-7s: %d of %d evaluations differ" % (G, B, Tt, mode, bad, reps), flush=True)
    total_bad += bad

def rollout(B, reps):
    """HSVRNNBVH.generate: 5 posterior + 64 prior steps (the persistent prior chain: cross-workgroup hand-offs by polling)."""
    global total_bad
    from neural_marionette_amd import HotPathOptions
    o = HotPathOptions(grid_size=32, Tcond=5)
    sd = synth.make_state_dict(o, seed=21, variant="default")
    net = NeuralMarionette(o); net.load_state_dict(sd); net = net.cuda().eval(); net.anneal(1)
    K, Z = o.nkeypoints, o.nlatent_kypt
    g = torch.Generator().manual_seed(B)
    kp = (torch.rand(B, 5, K, 4, generator=g) * 1.6 - 0.8).cuda()
    e_post = synth.make_eps((5, 10, B, Z), seed=50).cuda(); e_prior = synth.make_eps((64, B, Z), seed=51).cuda()
    with torch.no_grad():
        aff = net.kypt_detector.get_affinity().detach() if hasattr(net.kypt_detector, "get_affinity") else None
    if aff is None:
        from oracle import nm_oracle as O
        aff = O.affinity_v3(sd["kypt_detector.affinity_params"]).cuda()
    d = net.dyna_module
    def f():
        out = d.generate(kp, aff, Ttot=69, Tcond=5, eps_post=e_post, eps_prior=e_prior)
        torch.cuda.synchronize()
        return {k: out[k].detach().clone() for k in ("keypoints_cond", "keypoints_gen")}
    ref = f(); bad = 0
    for i in range(reps):
        r = f()
        if any(not torch.equal(ref[k], r[k]) for k in ref):
            bad += 1
    print("generate (persistent prior chain) B=%d: %d of %d evaluations differ" % (B, bad, reps), flush=True)
    total_bad += bad

for B in (1, 3, 4):
    rollout(B, 4 * n)
for mode in ("split16", "f16", "fp32"):
    detector_grads(32, 2, 3, mode, n)
detector_grads(32, 2, 3, "bf16", n, env={"NM355_STORE16_MIN": "4096"})
detector_grads(40, 1, 3, "split16", n)
detector_grads(64, 1, 4, "split16", max(6, n // 4))
detector_grads(64, 1, 4, "bf16", max(6, n // 4))
joint_step(32, 2, 3, n)
for mode in ("split16", "f16"):
    forward(64, 2, 4, mode, max(8, n // 2))
forward(48, 1, 3, "split16", n)
if "--concurrent" in sys.argv:
    _stop = True; _th.join(timeout=30)
print("TOTAL differing evaluations:", total_bad)
