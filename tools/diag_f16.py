#!/usr/bin/env python3
"""Conv mode 'f16' (fp16 products, fp32 accumulation) against the fp64 oracle: loss / gradient distances that the tests' tolerances
for that mode are taken from."""
import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_train_detector_gpu as T
for which in ("aist",):
    for seed in (11, 104):
        o, sd, vox = T._setup(seed=seed)
        w = T.WEIGHTINGS[which]
        ref_loss, ref, ref_out = T._oracle_grads(o, sd, vox, w, double=True)
        for mode in ("split16", "f16"):
            loss, got, out = T._hip_grads(o, sd, vox, w, mode=mode)
            gmax = max(r.abs().max().item() for r in ref.values())
            worst = ("", 0.0); l2n = 0.0; l2d = 0.0; wl2 = ("", 0.0); wcos = ("", 1.0)
            for k, r in ref.items():
                g = got[k]; scale = max(r.abs().max().item(), 1e-6 * gmax, 1e-30)
                e = (g.double() - r.double()).abs().max().item() / scale
                l2n += ((g.double() - r.double()) ** 2).sum().item(); l2d += (r.double() ** 2).sum().item()
                if e > worst[1]: worst = (k, e)
                tl2 = ((g.double() - r.double()).norm() / max(r.double().norm().item(), 1e-30)).item()
                cos = (g.double().flatten() @ r.double().flatten()).item() / max(g.double().norm().item() * r.double().norm().item(), 1e-300)
                if r.abs().max().item() > 1e-6 * gmax:
                    if tl2 > wl2[1]: wl2 = (k, tl2)
                    if cos < wcos[1]: wcos = (k, cos)
            kp = (out["keypoints"].detach().cpu().double() - ref_out["keypoints"].double()).abs().max().item() if "keypoints" in out and "keypoints" in ref_out else float("nan")
            print("    worst per-tensor L2 rel %.2e (%s); worst cosine %.5f (%s)" % (wl2[1], wl2[0][14:], wcos[1], wcos[0][14:]))
            print("%-6s seed %3d %-8s loss rel %.2e  worst grad rel %.2e (%s)  global L2 rel %.2e  kp %.2e" % (which, seed, mode, abs(loss - ref_loss) / max(1, abs(ref_loss)), worst[1], worst[0], (l2n / l2d) ** 0.5, kp))
