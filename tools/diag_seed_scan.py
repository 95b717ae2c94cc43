import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import test_train_detector_gpu as T
K = int(sys.argv[1])
for seed in [int(v) for v in sys.argv[2:]]:
    o, sd, vox = T._setup(G=32, B=1, T=3, seed=seed, K=K)
    ref_loss, ref, _ = T._oracle_grads(o, sd, vox, T.AIST, double=True)
    res = []
    for mode in ("split16", "fp32"):
        loss, got, fw = T._hip_grads(o, sd, vox, T.AIST, mode=mode)
        worst = max(((got[k].double() - ref[k].double()).abs().max().item() / max(ref[k].abs().max().item(), 1e-30), k) for k in ref)
        res.append("%s %.2e" % (mode, worst[0]))
    print("K", K, "seed", seed, res, flush=True)
