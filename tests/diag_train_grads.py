"""Diagnostic (lives under tests/ because it uses the oracle as the checker): detector-mode gradients of the HIP path vs the CPU
oracle in fp64, and the fp32 oracle's own distance to it.   python tests/diag_train_grads.py G B T seed weighting [modes...]"""
import sys, os, importlib.util
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))      # repo root (this file sits in tests/)
sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location("t", os.path.join(ROOT, "tests", "test_train_detector_gpu.py"))
m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)

G, B, T, seed, wname = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
o, sd, vox = m._setup(G=G, B=B, T=T, seed=seed)
w = m.WEIGHTINGS[wname]
_, g32, _ = m._oracle_grads(o, sd, vox, w)
_, g64, _ = m._oracle_grads(o, sd, vox, w, double=True)
for mode in (sys.argv[6:] or ["split16", "fp32"]):
    os.environ["NM355_CONV_MODE"] = mode
    loss, got, _ = m._hip_grads(o, sd, vox, w)
    gmax = max(r.abs().max().item() for r in g64.values())
    rows = []
    for k in g64:
        sc = max(g64[k].abs().max().item(), 1e-6 * gmax)
        rows.append(((got[k].double() - g64[k]).abs().max().item() / sc, (g32[k].double() - g64[k]).abs().max().item() / sc, k))
    rows.sort(reverse=True)
    print("mode", mode, "loss", loss)
    for e, e32, k in rows[:int(os.environ.get("TOPN", "12"))]:
        print("  hip %.2e  oracle32 %.2e  %s" % (e, e32, k))
