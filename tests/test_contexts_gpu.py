""""separate ctxs are independent".  Conv arithmetic mode, launch-profiler records and the non-finite status word are
per nm_ctx (NmLaunchState in csrc/nm_common.h), so two networks in one process - e.g. bench.py's secondary configs beside the
headline network - do not share a mode.  Interleaved calls on two contexts in different modes must reproduce, bit for bit, what each
context computes alone."""
import ctypes as C

import pytest
import torch

from neural_marionette_amd import NeuralMarionette, HotPathOptions, synth, _lib

pytestmark = pytest.mark.gpu

ACTS = {"detector": True, "learner": True}


def _net(o, sd, mode):
    net = NeuralMarionette(o)
    net.load_state_dict(sd)
    net = net.cuda().eval()
    net.anneal(1)
    net.set_conv_mode(mode)
    return net


def test_two_contexts_in_different_conv_modes_interleaved():
    o = HotPathOptions(grid_size=32)
    sd = synth.make_state_dict(o, seed=3, variant="peaky")
    vox = synth.figure_clip(2, 4, 32, seed=4).cuda()
    eps = synth.make_eps((4, 10, 2, o.nlatent_kypt), seed=5).cuda()
    keys = ("keypoints", "heatmaps", "recon", "first_feature", "z_kypts", "h_kypts")
    alone = {}
    for mode in ("fp32", "split16", "f16"):
        n = _net(o, sd, mode)
        with torch.no_grad():
            n(vox, ACTS, eps=eps)                     # (first call builds the tree through the two-call path)
            out = n(vox, ACTS, eps=eps)
        torch.cuda.synchronize()
        alone[mode] = {k: out[k].clone() for k in keys}
        del n
    assert not torch.equal(alone["fp32"]["recon"], alone["split16"]["recon"])       # the modes do differ in the last bits
    assert not torch.equal(alone["f16"]["recon"], alone["split16"]["recon"])
    a, b, c = _net(o, sd, "fp32"), _net(o, sd, "split16"), _net(o, sd, "f16")
    with torch.no_grad():
        for n in (a, b, c):
            n(vox, ACTS, eps=eps)
        for rep in range(2):                          # interleave: every call follows a call of another context in another mode
            oa = a(vox, ACTS, eps=eps); ob = b(vox, ACTS, eps=eps); oc = c(vox, ACTS, eps=eps)
            ob2 = b(vox, ACTS, eps=eps); oa2 = a(vox, ACTS, eps=eps)
    torch.cuda.synchronize()
    for mode, outs in (("fp32", (oa, oa2)), ("split16", (ob, ob2)), ("f16", (oc,))):
        for o_ in outs:
            for k in keys:
                assert torch.equal(o_[k], alone[mode][k]), f"context in mode {mode}: {k} changed when interleaved with other contexts"
    lib = _lib.load()
    assert lib.nm_get_conv_mode(a._engine.ctx.handle) == 0 and lib.nm_get_conv_mode(b._engine.ctx.handle) == 1
    assert lib.nm_get_conv_mode(c._engine.ctx.handle) == 3


def test_profiler_records_belong_to_their_context():
    o = HotPathOptions(grid_size=32)
    sd = synth.make_state_dict(o, seed=3, variant="peaky")
    vox = synth.figure_clip(1, 2, 32, seed=4).cuda()
    a, b = _net(o, sd, "split16"), _net(o, sd, "split16")
    det = {"detector": True, "learner": False}
    with torch.no_grad():
        a(vox, det); b(vox, det)
    lib = _lib.load()
    ha, hb = a._engine.ctx.handle, b._engine.ctx.handle
    _lib.check(lib.nm_prof_enable(ha, 1))
    with torch.no_grad():
        b(vox, det)                                   # not profiled: must leave no record anywhere
        a(vox, det)
    _lib.check(lib.nm_prof_enable(ha, 0))

    def launches(h):
        tot = 0
        for v in range(13):
            ms, fl, n = C.c_double(), C.c_double(), C.c_int64()
            _lib.check(lib.nm_prof_read(h, v, C.byref(ms), C.byref(fl), C.byref(n)))
            tot += n.value
        return tot
    na, nb = launches(ha), launches(hb)
    assert nb == 0 and na > 20, (na, nb)
    # twice the calls on a -> twice the records; b's calls in between still add nothing
    _lib.check(lib.nm_prof_enable(ha, 1))
    with torch.no_grad():
        a(vox, det); b(vox, det); a(vox, det)
    _lib.check(lib.nm_prof_enable(ha, 0))
    assert launches(ha) == 2 * na and launches(hb) == 0


def test_nonfinite_status_is_per_context():
    """An activation beyond the fp16 range in one context's split-fp16 forward raises that context's status word only."""
    o = HotPathOptions(grid_size=32)
    sd = synth.make_state_dict(o, seed=3, variant="peaky")
    bad = {k: v.clone() for k, v in sd.items()}
    k0 = "kypt_detector.vox_to_kypt.extract_features.0.block.0.weight"
    bad[k0] = bad[k0] * 1e9                         # first-layer output ~1e9 * O(1): beyond 65504 before the GroupNorm, inf in the split
    vox = synth.figure_clip(1, 2, 32, seed=4).cuda()
    good_net, bad_net = _net(o, sd, "split16"), _net(o, bad, "split16")
    det = {"detector": True, "learner": False}
    with torch.no_grad():
        good_net(vox, det); bad_net(vox, det); good_net(vox, det)
    good_net.check_finite()                           # must not raise
    raised = False
    try:
        bad_net.check_finite()
    except _lib.NmError:
        raised = True
    if not raised:
        pytest.skip("the scaled weights did not leave the fp16 range on this path (GroupNorm statistics stayed finite)")
    good_net.check_finite()
