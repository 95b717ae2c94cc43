"""Batch additivity of the detector gradient in a reduced-precision conv mode at 64^3, per tensor (diagnostic for
tests/test_train_detector_gpu.py::test_gradient_properties_f16_mode_at_64cubed).  usage: diag_additivity.py [mode]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import test_train_detector_gpu as T
mode = sys.argv[1] if len(sys.argv) > 1 else "f16"
o, sd, vox = T._setup(G=64, B=2, T=4, seed=83)
_, g_ab, _ = T._hip_grads(o, sd, vox, T.AIST, mode=mode)
_, g_a, _ = T._hip_grads(o, sd, vox[:1].contiguous(), T.AIST, mode=mode)
_, g_b, _ = T._hip_grads(o, sd, vox[1:].contiguous(), T.AIST, mode=mode)
gmax = max(v.abs().max().item() for v in g_ab.values())
rows = []
for k, v in g_ab.items():
    avg = 0.5 * (g_a[k].double() + g_b[k].double())
    scale = max(avg.abs().max().item(), 1e-6 * gmax)
    rows.append(((v.double() - avg).abs().max().item() / scale, k, avg.abs().max().item(), g_a[k].abs().max().item(), g_b[k].abs().max().item()))
rows.sort(reverse=True)
print("env K2F16=%s CONVT_F16=%s mode %s gmax %.3e" % (os.environ.get("NM355_WGRAD_K2F16"), os.environ.get("NM355_CONVT_F16"), mode, gmax))
for r in rows[:8]:
    print("  %.3e  %-60s |avg|max %.3e  |a| %.3e |b| %.3e" % r)
