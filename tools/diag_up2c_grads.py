import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import test_train_detector_gpu as T
K = int(sys.argv[1]); out = sys.argv[2]
o, sd, vox = T._setup(G=32, B=1, T=3, seed=71 + K, K=K)
loss, got, fw = T._hip_grads(o, sd, vox, T.AIST, mode=os.environ.get("DIAG_MODE") or None)
torch.save(dict(loss=loss, grads=got, recon=fw["recon"].cpu(), kp=fw["keypoints"].cpu()), out)
if len(sys.argv) > 3:
    a = torch.load(sys.argv[3]); b = torch.load(out)
    print("loss", a["loss"], b["loss"], "recon max diff %.3e" % (a["recon"] - b["recon"]).abs().max().item())
    d = (a["recon"] - b["recon"]).abs()[0, 0, 0]
    print("recon diff interior %.3e shell %.3e" % (d[1:-1, 1:-1, 1:-1].max().item(), max(d[0].max(), d[-1].max(), d[:, 0].max(), d[:, -1].max(), d[:, :, 0].max(), d[:, :, -1].max()).item()))
    rows = []
    for k in a["grads"]:
        ga, gb = a["grads"][k], b["grads"][k]
        rows.append(((ga - gb).abs().max().item() / max(ga.abs().max().item(), 1e-30), k, ga.abs().max().item()))
    rows.sort(reverse=True)
    for r in rows[:8]: print("%.3e %s |g|max %.3e" % r)
if len(sys.argv) > 3:
    ref_loss, ref, ref_out = T._oracle_grads(o, sd, vox, T.AIST, double=True)
    r64 = ref_out["recon"].double()
    for name, x in (("first", a), ("second", b)):
        e = (x["recon"].double() - r64).abs().max().item()
        ek = (x["kp"].double() - ref_out["keypoints"].double()).abs().max().item()
        worst = max(((x["grads"][k].double() - ref[k].double()).abs().max().item() / max(ref[k].abs().max().item(), 1e-30), k) for k in ref)
        print(name, "recon err vs fp64 %.3e  keypoints %.3e  worst grad rel err %.3e (%s)  loss %.9f vs %.9f" % (e, ek, worst[0], worst[1], x["loss"], ref_loss))
