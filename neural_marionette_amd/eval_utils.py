"""Device versions of the reference's evaluation metrics (utils/eval_utils.py), same call surface:

    evaluate(name, scores_dict, params)        :4-10      name in {'semantic', 'voxel_chamfer'}
    evaluate_final(name, scores_dict)          :12-27     (without the reference's savetxt side effects)
    voxel_chamfer_distance(scores, params)     :29-55     params: voxel, recon (B,T,1,G,G,G) on the device, network=<NeuralMarionette>
    semantic_scores(scores, params)            :59-90     params: keypoints (B,T,K,4), gt_keypoints (B,T,K',3), network=...

Differences: the inputs are not modified in place (the reference thresholds params['recon'] and overwrites low-intensity
keypoints); `params['network']` (or `params['ctx']`) names the NeuralMarionette whose device context runs the kernels.
The chamfer distance is computed from exact integer distance transforms instead of the (N, M) distance matrix."""
import numpy as np
import torch

from . import _lib


def _ctx(params):
    net = params.get("network")
    if net is not None:
        return net._engine.ready()
    ctx = params.get("ctx")
    if ctx is None:
        raise _lib.NmError("eval_utils: params needs 'network' (a neural_marionette_amd.NeuralMarionette) or 'ctx'")
    return ctx


def evaluate(name, scores_dict, params):
    if name == "semantic":
        return semantic_scores(scores_dict[name], params)
    if name == "voxel_chamfer":
        return voxel_chamfer_distance(scores_dict[name], params)
    raise ValueError("invalid evaluation metric.")


def evaluate_final(name, scores_dict):
    if name == "semantic":
        scores = np.array(scores_dict[name], dtype=np.float64)
        scores = scores / scores[0].sum()
        return scores.max(axis=-1).mean()
    if name == "voxel_chamfer":
        return np.array(scores_dict[name]).mean() * 1e4              # note that result is 1e4X (eval_utils.py:25)
    raise ValueError("invalid evaluation metric.")


def chamfer_per_frame(ctx, voxel, recon):
    """(B,T) float64 chamfer distances of every frame (device tensor)."""
    B, T = voxel.shape[:2]
    G = voxel.shape[-1]
    v = voxel.detach().to(torch.float32).contiguous()
    r = recon.detach().to(torch.float32).contiguous()
    out = torch.empty(B, T, dtype=torch.float64, device=v.device)
    _lib.check(ctx.lib.nm_eval_voxel_chamfer(ctx.handle, _lib.ptr(v), _lib.ptr(r), B, T, G, _lib.ptr(out)), "eval_voxel_chamfer")
    return out


def voxel_chamfer_distance(scores, params):
    if scores is None:
        scores = []
    pf = chamfer_per_frame(_ctx(params), params["voxel"], params["recon"]).cpu().numpy()
    B, T = pf.shape
    for b in range(B):
        scores.append([float(pf[b].sum() / T)])
    return dict(scores=scores, scores_log=float(pf.sum() / (B * T)))


def semantic_scores(scores, params):
    kp = params["keypoints"].detach().to(torch.float32).contiguous()
    gt = params["gt_keypoints"].detach().to(torch.float32).contiguous()
    B, T, K, _ = kp.shape
    Kg = gt.shape[2]
    ctx = _ctx(params)
    closest = torch.empty(B * T, Kg, dtype=torch.int32, device=kp.device)
    counts = torch.zeros(Kg, K, dtype=torch.int64, device=kp.device)
    _lib.check(ctx.lib.nm_eval_semantic(ctx.handle, _lib.ptr(kp), _lib.ptr(gt), B * T, K, Kg, _lib.ptr(closest), counts.data_ptr()),
               "eval_semantic")
    c = counts.cpu().numpy()
    if scores is None:
        scores = np.zeros((Kg, K))
    scores += c
    temp = np.array([(row / row.sum()).max() for row in c.astype(np.float64)], dtype=np.float32)
    return dict(scores=scores, scores_log=temp.mean())
