"""CPU ORACLE — test infrastructure, not product code.

A functional restatement, on PyTorch *CPU* ops, of the reference's hot path
(jinseokbae/neural_marionette): the voxel keypoint detector and the hierarchical
skeleton VRNN.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import this module; the product package
``neural_marionette_amd`` never does.

Where the arithmetic lives: every primitive below is an ATen CPU kernel of the
third-party dependency the reference itself uses (PyTorch; the reference pins
torch==1.7.1+cu110 in setup.sh:2, this image carries torch 2.10.0).  Semantics
that were checked against the reference in this container:
GroupNorm eps 1e-5 / biased variance, LeakyReLU slope 0.01, the residual
``F.leaky_relu(x, True)`` being the identity, Softplus beta 1 / threshold 20,
trilinear ``align_corners=False``, BCELoss log clamp at -100, GRUCell gate order
r,z,n, lower median, first-minimum argmin.

Pinning: the reference has no tests or golden vectors of its own, so this oracle is
pinned by (1) ``tests/test_oracle_vs_reference.py`` which imports the reference
from /root/reference when that tree exists (build container only) and compares
every output on identical seeded weights/inputs, and (2) the fixtures under
``tests/golden/`` written by ``tools/make_golden.py`` from the reference itself.

Each function cites the reference lines it follows.
"""
from __future__ import annotations

import heapq
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = Dict[str, Tensor]

GN_EPS = 1e-5
LRELU = 0.01


# --------------------------------------------------------------------------------------
# building blocks (modules/vox_modules.py)
# --------------------------------------------------------------------------------------
def _gn(x: Tensor, sd: SD, key: str) -> Tensor:
    w = sd[key + ".weight"]
    return F.group_norm(x, w.numel() // 16, w, sd[key + ".bias"], GN_EPS)


def conv_gn_lrelu(x: Tensor, sd: SD, conv: str, gn: str, stride: int, pad: int) -> Tensor:
    """Conv3d -> GroupNorm(C/16) -> LeakyReLU(0.01): vox_modules.py:11-16 (k5/k3 'same')
    and :52-57 (k2 s2 'pool')."""
    y = F.conv3d(x, sd[conv + ".weight"], sd[conv + ".bias"], stride=stride, padding=pad)
    return F.leaky_relu(_gn(y, sd, gn), LRELU)


def res_block(x: Tensor, sd: SD, p: str) -> Tensor:
    """vox_modules.py:22-47.  The trailing ``F.leaky_relu(res + skip, True)`` passes
    True as negative_slope (== 1.0), i.e. it is the identity."""
    r = F.conv3d(x, sd[p + ".res_branch.0.weight"], sd[p + ".res_branch.0.bias"], padding=1)
    r = F.leaky_relu(_gn(r, sd, p + ".res_branch.1"), LRELU)
    r = F.conv3d(r, sd[p + ".res_branch.3.weight"], sd[p + ".res_branch.3.bias"], padding=1)
    r = _gn(r, sd, p + ".res_branch.4")
    if (p + ".skip_con.0.weight") in sd:
        s = F.conv3d(x, sd[p + ".skip_con.0.weight"], sd[p + ".skip_con.0.bias"])
        s = _gn(s, sd, p + ".skip_con.1")
    else:
        s = x
    return r + s


def pool_block(x: Tensor, sd: SD, p: str) -> Tensor:
    return conv_gn_lrelu(x, sd, p + ".stride_conv.0", p + ".stride_conv.1", 2, 0)


def up_block(x: Tensor, sd: SD, p: str, outpad: int) -> Tensor:
    """ConvTranspose3d(k2,s2,output_padding) -> GN -> LeakyReLU: vox_modules.py:63-75."""
    y = F.conv_transpose3d(x, sd[p + ".block.0.weight"], sd[p + ".block.0.bias"], stride=2,
                           output_padding=outpad)
    return F.leaky_relu(_gn(y, sd, p + ".block.1"), LRELU)


def hourglass(x: Tensor, sd: SD, p: str, N: int) -> Tensor:
    """vox_modules.py:78-120; output paddings from :81."""
    op3, op2, op1 = (N // 4) % 2, (N // 2) % 2, N % 2
    s1 = res_block(x, sd, p + ".skip_res1")
    x = res_block(pool_block(x, sd, p + ".encoder_pool1"), sd, p + ".encoder_res1")
    s2 = res_block(x, sd, p + ".skip_res2")
    x = res_block(pool_block(x, sd, p + ".encoder_pool2"), sd, p + ".encoder_res2")
    s3 = res_block(x, sd, p + ".skip_res3")
    x = res_block(pool_block(x, sd, p + ".encoder_pool3"), sd, p + ".encoder_res3")
    x = res_block(x, sd, p + ".decoder_res3")
    x = up_block(x, sd, p + ".decoder_upsample3", op3) + s3
    x = res_block(x, sd, p + ".decoder_res2")
    x = up_block(x, sd, p + ".decoder_upsample2", op2) + s2
    x = res_block(x, sd, p + ".decoder_res1")
    x = up_block(x, sd, p + ".decoder_upsample1", op1) + s1
    return x


def feature_net(x: Tensor, sd: SD, p: str, g: int, taps: Optional[dict] = None) -> Tensor:
    """kypt_detector.py:264-272: Basic3D(k5) / Pool / Res / Pool / HG(N=g) / Res."""
    x = conv_gn_lrelu(x, sd, p + ".0.block.0", p + ".0.block.1", 1, 2)
    if taps is not None: taps[p + ".0"] = x
    x = pool_block(x, sd, p + ".1")
    if taps is not None: taps[p + ".1"] = x
    x = res_block(x, sd, p + ".2")
    if taps is not None: taps[p + ".2"] = x
    x = pool_block(x, sd, p + ".3")
    if taps is not None: taps[p + ".3"] = x
    x = hourglass(x, sd, p + ".4", g)
    if taps is not None: taps[p + ".4"] = x
    x = res_block(x, sd, p + ".5")
    if taps is not None: taps[p + ".5"] = x
    return x


# --------------------------------------------------------------------------------------
# detector utilities (utils/kypt_detector_utils.py)
# --------------------------------------------------------------------------------------
def coord_channels(X: Sequence[int]) -> Tensor:
    """(D, X1..XD) ramps linspace(-1,1,Xd), 'ij' order: kypt_detector_utils.py:19-24."""
    lin = [torch.linspace(-1.0, 1.0, n) for n in X]
    return torch.stack(torch.meshgrid(*lin, indexing="ij"), dim=0)


def add_coords(x: Tensor) -> Tensor:
    """kypt_detector_utils.py:4-26 — channel order [input..., x1, x2, x3]."""
    B = x.shape[0]
    c = coord_channels(x.shape[2:]).to(x.dtype)
    return torch.cat([x, c[None].expand(B, *c.shape)], dim=1)


def heatmap_to_keypoints(hm: Tensor) -> Tensor:
    """kypt_detector_utils.py:28-55.  (B,K,g,g,g) -> (B,K,4) = (x1,x2,x3,intensity);
    linear (not softmax) normalisation of marginal sums of hm + 1e-6."""
    B, K = hm.shape[:2]
    G = hm.shape[2:]
    inten = hm.mean(dim=(2, 3, 4))
    inten = inten / (inten.max(dim=-1, keepdim=True).values + 1e-6)
    cs = []
    for d in range(3):
        other = tuple(i + 2 for i in range(3) if i != d)
        w = (hm + 1e-6).sum(dim=other)                      # (B,K,Gd)
        w = w / w.sum(dim=-1, keepdim=True)
        cs.append((w * torch.linspace(-1.0, 1.0, G[d])).sum(dim=-1))
    return torch.stack(cs + [inten], dim=-1)


def gaussian_map(kp: Tensor, sigma, g: int) -> Tensor:
    """kypt_detector_utils.py:57-90 for a python-float sigma: width = 2 (sigma/g)^2 in
    python double arithmetic, map = ((1 * e_x1) * e_x2) * e_x3 * intensity, each
    e_d = exp(-(lin - c_d)^2 / width).  kp (B,K,4) -> (B,K,g,g,g).  A (K,) tensor of sigmas (fixed_sigma = 0: one 0-dim
    tensor per keypoint in the reference's K calls) gives a per-keypoint width in tensor arithmetic."""
    width = 2.0 * (sigma / g) ** 2.0
    if torch.is_tensor(width):
        width = width[None, :, None]
    lin = torch.linspace(-1.0, 1.0, g)
    B, K = kp.shape[:2]
    m = torch.ones(B, K, g, g, g)
    shapes = [(B, K, g, 1, 1), (B, K, 1, g, 1), (B, K, 1, 1, g)]
    for d in range(3):
        e = (-(lin[None, None] - kp[:, :, d, None]).pow(2) / width).exp()
        m = m * e.reshape(shapes[d])
    return m * kp[:, :, 3, None, None, None]


def loss_sparsity(heatmaps: Tensor) -> Tensor:
    """kypt_detector_utils.py:92-103 -> (B,T)."""
    return heatmaps.mean(dim=(3, 4, 5)).abs().mean(dim=2)


def loss_separation(kp: Tensor, sep_sigma: float) -> Tensor:
    """kypt_detector_utils.py:105-133 -> (B,)."""
    xyz = kp[..., :3]
    K = xyz.shape[2]
    disp = xyz - xyz.mean(dim=1, keepdim=True)
    d2 = (disp[:, :, :, None] - disp[:, :, None]).pow(2).sum(-1).mean(dim=1)
    m = (-d2 / (2.0 * sep_sigma ** 2.0)).exp()
    return (m.sum(dim=(1, 2)) - K) / (K * (K - 1))


def loss_volume_chamfer(seq: Tensor, kp: Tensor) -> Tensor:
    """kypt_detector_utils.py:140-153 -> (B,T): occupancy-masked mean over voxels of the
    squared distance to the nearest keypoint."""
    B, T = seq.shape[:2]
    c = coord_channels(seq.shape[3:])                      # (3,G,G,G)
    out = []
    for t in range(T):
        p = kp[:, t, :, :3][:, :, :, None, None, None]      # (B,K,3,1,1,1)
        d = (c[None, None] - p).pow(2).sum(dim=2)           # (B,K,G,G,G)
        d = d.min(dim=1, keepdim=True).values * seq[:, t]
        out.append(d.sum(dim=(1, 2, 3, 4)) / seq[:, t].sum(dim=(1, 2, 3, 4)))
    return torch.stack(out, dim=1)


def loss_volume_gaussian(seq: Tensor, kp: Tensor, sigma: float) -> Tensor:
    """vol_fit_type 'gaussian', kypt_detector_utils.py:154-169, AS THE REFERENCE COMPUTES IT -> (B,T).  The call there hands a
    (B,1,3) slice of coordinates to extract_gaussian_map_from_keypoints (:57-90), which takes the last entry for an intensity: the map
    is TWO-dimensional, m_k[i,j] = ((1 exp(-(lin_i - c0)^2 / w)) exp(-(lin_j - c1)^2 / w)) c2 with w = 2 (4 sigma / G)^2, and the mask
    max_k m_k has shape (B,1,G,G); multiplied with the (B,1,G,G,G) frame it broadcasts along the frame's FIRST spatial axis and ACROSS
    the batch: reg[b',t] = sum_b sum_{a,i,j} (1 - mask[b,i,j]) seq[b',t,a,i,j] / sum seq[b',t].  (For B = 1: the occupied mass outside
    the union of K blobs in the (i,j) projection.)  Restated with that broadcasting spelled out."""
    B, T = seq.shape[:2]
    G = seq.shape[3]
    width = 2.0 * (sigma * 4.0 / G) ** 2.0
    lin = torch.linspace(-1.0, 1.0, G, dtype=seq.dtype)
    out = []
    for t in range(T):
        c = kp[:, t, :, :3]                                                  # (B,K,3)
        e0 = (-(lin[None, None] - c[:, :, 0, None]).pow(2) / width).exp()    # (B,K,G) along i
        e1 = (-(lin[None, None] - c[:, :, 1, None]).pow(2) / width).exp()    # (B,K,G) along j
        m = ((torch.ones(B, c.shape[1], G, G, dtype=seq.dtype) * e0[:, :, :, None]) * e1[:, :, None, :]) * c[:, :, 2, None, None]
        mask = m.max(dim=1, keepdim=True).values                             # (B,1,G,G)
        frame = seq[:, t]                                                    # (B,1,G,G,G)
        prod = (1 - mask)[None, :, :, :, :] * frame[:, :, :, :, :]           # (B', B, G(a), G(i), G(j)): the reference's broadcast, explicit
        out.append(prod.sum(dim=(1, 2, 3, 4)) / frame.sum(dim=(1, 2, 3, 4)))
    return torch.stack(out, dim=1)


def loss_graph_consistency_v1(kp: Tensor, aff: Tensor):
    """kypt_detector_utils.py:172-225 with ver=1, all four switches on.
    Returns local (B,T), time (B,T), sparsity (1,1), intensity (1,1)=0."""
    infl = aff.max(dim=0).values[None, None]                # (1,1,K,K,1)
    pos = kp[..., :3]
    dist = (pos[:, :, :, None] - pos[:, :, None]).pow(2).sum(dim=-1, keepdim=True)
    local = (dist * infl).mean(dim=(2, 3, 4))
    tim = ((dist - dist.mean(dim=1, keepdim=True)).abs() * infl).mean(dim=(2, 3, 4))
    a = aff.squeeze(-1)
    s = (a[:, None] * a[None]).pow(2).sum(dim=1, keepdim=True) - a[:, None].pow(4)
    s = s.sum(dim=(0, 1)).mean(dim=(0, 1), keepdim=True)
    return local, tim, s, torch.zeros(1, 1)


def loss_graph_traj_v1(kp: Tensor, aff: Tensor) -> Tensor:
    """kypt_detector_utils.py:228-265 with ver=1 -> (1,1)."""
    infl = aff.squeeze(-1).max(dim=0).values[None, None]
    vel = kp[:, 1:, :, :3] - kp[:, :-1, :, :3]
    acc = vel[:, 1:] - vel[:, :-1]
    cos = torch.nn.CosineSimilarity(dim=-1, eps=1e-6)
    vc = (((-cos(vel[:, :, :, None], vel[:, :, None]) + 1) / 2) * infl).mean(dim=(0, 1))
    ac = (((-cos(acc[:, :, :, None], acc[:, :, None]) + 1) / 2) * infl).mean(dim=(0, 1))
    return (vc + ac).mean(dim=(0, 1), keepdim=True)


def affinity_v3(params: Tensor) -> Tensor:
    """kypt_detector.py:191-199: softmax over the K-1 logits of each row, then put a zero
    on the diagonal -> (N,K,K,1)."""
    N, K, _ = params.shape
    P = torch.softmax(params, dim=-1)
    W = torch.zeros(N, K, K, dtype=params.dtype)
    for k in range(K):
        W[:, k, :k] = P[:, k, :k]
        W[:, k, k + 1:] = P[:, k, k:]
    return W[..., None]


def affinity(params: Tensor, ver: int = 3) -> Tensor:
    """KyptDetector.get_affinity (kypt_detector.py:171-210) -> (N,K,K,1).  ver 3: affinity_v3; 0 / 1 / 2 take (N,K,K) parameters:
    0 row softmax (:173-174); 1 softplus, Gram matrix per neighbour, zero diagonal, rows divided by (row sum + 1e-6) (:177-182);
    2 softplus, zero diagonal, row softmax (:185-188).  Version 4 (Gumbel noise) is not restated."""
    if ver == 3:
        return affinity_v3(params)
    N, K, _ = params.shape
    eye = torch.eye(K, dtype=params.dtype)[None]
    if ver == 0:
        W = torch.softmax(params, dim=2)
    elif ver == 1:
        S = F.softplus(params)
        W = torch.stack([S[n] @ S[n].T for n in range(N)], dim=0)
        W = W * (1 - eye)
        W = W / (W.sum(dim=-1, keepdim=True) + 1e-6)
    elif ver == 2:
        W = torch.softmax(F.softplus(params) * (1 - eye), dim=2)
    else:
        raise NotImplementedError("affinity_ver %d" % ver)
    return W[..., None]


# --------------------------------------------------------------------------------------
# detector (model/kypt_detector.py)
# --------------------------------------------------------------------------------------
V2K = "kypt_detector.vox_to_kypt"
K2V = "kypt_detector.kypt_to_vox"
DEC = K2V + ".decode_voxel_from_combined_representation"


def vox_to_kypt(sd: SD, opts, seq: Tensor, taps: Optional[dict] = None):
    """VoxToKyptNet.forward for const_intensity==3, fixed sigma: kypt_detector.py:299-364."""
    B, T = seq.shape[:2]
    K, g = opts.nkeypoints, opts.grid_size // 4
    # spatio-temporal heat-map from the clip mean, once per clip (:311-316)
    st = feature_net(add_coords(seq.mean(dim=1)), sd, V2K + ".extract_spatio_temporal_features", g, taps)
    hw = V2K + ".extract_spatio_temporal_heatmaps_from_features.0"
    prev = F.leaky_relu(F.conv3d(st, sd[hw + ".weight"], sd[hw + ".bias"]), LRELU)
    if taps is not None: taps["st_heatmap"] = prev
    pw, pb = sd[V2K + ".propagate_heatmaps.0.weight"], sd[V2K + ".propagate_heatmaps.0.bias"]
    hw = V2K + ".extract_heatmaps_from_features.0"
    hms, kps, gss = [], [], []
    first = None
    # fixed_sigma = 0 (kypt_detector.py:258-260, 303-306): sigmas = sigmoid(parameter) * max_sigma, max_sigma = 2 gaussian_sigma
    sig = opts.gaussian_sigma if getattr(opts, "fixed_sigma", 1) else torch.sigmoid(sd[V2K + ".sigmas"]) * (opts.gaussian_sigma * 2.0)
    for t in range(T):
        feat = feature_net(add_coords(seq[:, t]), sd, V2K + ".extract_features", g,
                           taps if (t == 0) else None)
        if t == 0:
            first = feat
        hm = F.leaky_relu(F.conv3d(feat, sd[hw + ".weight"], sd[hw + ".bias"]), LRELU)
        pair = torch.cat([hm.reshape(B * K, 1, g, g, g), prev.reshape(B * K, 1, g, g, g)], dim=1)
        hm = F.softplus(F.conv3d(pair, pw, pb)).view(B, K, g, g, g)     # (:339-343); prev not updated for ==3
        kp = heatmap_to_keypoints(hm)
        gs = gaussian_map(kp, sig, g)                                   # K per-keypoint calls == one batched call
        hms.append(hm); kps.append(kp); gss.append(gs)
    return torch.stack(hms, 1), torch.stack(kps, 1), torch.stack(gss, 1), first


def kypt_to_vox(sd: SD, opts, gaussians: Tensor, first_feature: Tensor, first_frame: Tensor,
                taps: Optional[dict] = None) -> Tensor:
    """KyptToVoxNet.forward: kypt_detector.py:388-460.  gaussian_cat_type 'max' / 'sum' (:396-401): every one of the K Gaussian channels
    carries the maximum / the sum clipped to [0, 1] over the K maps."""
    T = gaussians.shape[1]
    cat = getattr(opts, "gaussian_cat_type", "none")
    if cat == "max":
        gaussians = gaussians.max(dim=2, keepdim=True).values.expand_as(gaussians)
    elif cat == "sum":
        gaussians = gaussians.sum(dim=2, keepdim=True).clip(0, 1).expand_as(gaussians)
    elif cat != "none":
        raise NotImplementedError("gaussian_cat_type %r" % (cat,))
    aw = K2V + ".adjust_combined_representation.0"
    out = []
    for t in range(T):
        x = add_coords(torch.cat([gaussians[:, t], first_feature, gaussians[:, 0]], dim=1))
        x = F.leaky_relu(F.conv3d(x, sd[aw + ".weight"], sd[aw + ".bias"]), LRELU)
        if taps is not None and t == 0: taps["dec_adjust"] = x
        x = F.interpolate(x, scale_factor=2.0, mode="trilinear", align_corners=False)
        x = conv_gn_lrelu(x, sd, DEC + ".1", DEC + ".2", 1, 1)
        x = conv_gn_lrelu(x, sd, DEC + ".4", DEC + ".5", 1, 1)
        if taps is not None and t == 0: taps["dec_32"] = x
        x = F.interpolate(x, scale_factor=2.0, mode="trilinear", align_corners=False)
        x = conv_gn_lrelu(x, sd, DEC + ".8", DEC + ".9", 1, 1)
        x = conv_gn_lrelu(x, sd, DEC + ".11", DEC + ".12", 1, 1)
        if taps is not None and t == 0: taps["dec_64"] = x
        x = F.conv3d(x, sd[DEC + ".14.weight"], sd[DEC + ".14.bias"])
        out.append(torch.sigmoid(10.0 * (torch.tanh(x) + first_frame - 0.5)))
    return torch.stack(out, dim=1)


def detector_forward(sd: SD, opts, seq: Tensor, affinity_on: bool = True,
                     taps: Optional[dict] = None) -> Dict[str, Tensor]:
    """KyptDetector.forward: kypt_detector.py:81-169.  ``affinity_on`` mirrors
    ``self.affinity_start`` (set by anneal(), :71-78)."""
    B, T = seq.shape[:2]
    heatmaps, keypoints, gaussians, first = vox_to_kypt(sd, opts, seq, taps)
    recon = kypt_to_vox(sd, opts, gaussians, first, seq[:, 0], taps)
    recon_loss = F.binary_cross_entropy(recon, seq, reduction="none").mean(dim=(2, 3, 4, 5))
    zeros = torch.zeros(B, T)
    if opts.vol_fit_type == "chamfer":
        vol = loss_volume_chamfer(seq, keypoints)
    elif opts.vol_fit_type == "gaussian":
        vol = loss_volume_gaussian(seq, keypoints, opts.gaussian_sigma)
    else:
        vol = zeros
    if affinity_on:
        aff = affinity(sd["kypt_detector.affinity_params"], getattr(opts, "affinity_ver", 3))
        kk = keypoints.detach() if opts.keypoints_detach else keypoints
        local, tim, spars, inten = loss_graph_consistency_v1(kk, aff)
        traj = loss_graph_traj_v1(kk, aff) if opts.graph_traj_weight > 0 else zeros
    else:
        aff = None
        local = tim = spars = inten = traj = zeros
    return dict(
        recon=recon, keypoints=keypoints, heatmaps=heatmaps, affinity=aff,
        recon_loss=recon_loss.mean(), vol_fit_reg=vol.mean(), kypt_const_loss=zeros.mean(),
        separation_loss=loss_separation(keypoints, opts.sep_sigma).mean(),
        sparsity_loss=loss_sparsity(heatmaps).mean(),
        local_const_loss=local.mean(), time_const_loss=tim.mean(),
        sparsity_const_loss=spars.mean(), intensity_const_loss=inten.mean(),
        graph_traj_loss=traj.mean(), graph_vol_loss=zeros.mean(),
        first_feature=first, gaussians=gaussians,
    )


def decode_from_keypoints(sd: SD, opts, keypoints: Tensor, first_feature: Tensor,
                          first_frame: Tensor) -> Tensor:
    """KyptDetector.decode_from_dyna: kypt_detector.py:213-241 -> (B,Tg,1,G,G,G)."""
    g = opts.grid_size // 4
    gs = torch.stack([gaussian_map(keypoints[:, t], opts.gaussian_sigma, g)
                      for t in range(keypoints.shape[1])], dim=1)
    return kypt_to_vox(sd, opts, gs, first_feature, first_frame)


# --------------------------------------------------------------------------------------
# skeleton tree from the affinity (utils/dyna_utils.py) — host, float64
# --------------------------------------------------------------------------------------
def _adjacency_lists(K: int, A: np.ndarray, W: Optional[np.ndarray]):
    """Neighbour lists in the insertion order an undirected graph container gives when
    edges are added in row-major order of the non-zeros of A (dyna_utils.py:24-32)."""
    adj: List[Dict[int, float]] = [dict() for _ in range(K)]
    rows, cols = np.where(A)
    for u, v in zip(rows.tolist(), cols.tolist()):
        w = 1 if W is None else W[u, v]      # int for the unweighted pass, np.float32 for the nudged pass
        adj[u][v] = w
        adj[v][u] = w
    return adj


def _all_pairs(K: int, adj, big: float) -> np.ndarray:
    """Single-source Dijkstra from every node with a (dist, counter) heap; distances
    accumulate from the source outward in the dtype of the edge weights (python int for
    the unweighted passes, np.float32 for the tie-nudged passes — the reference hands
    np.float32 weights to its graph library), unreachable pairs keep ``big``
    (dyna_utils.py:21-34)."""
    D = np.ones((K, K)) * big
    for s in range(K):
        dist: Dict[int, float] = {}
        seen = {s: 0}
        heap = [(0, 0, s)]
        cnt = 1
        while heap:
            d, _, v = heapq.heappop(heap)
            if v in dist:
                continue
            dist[v] = d
            for u, w in adj[v].items():
                nd = d + w
                if u in dist:
                    continue
                if u not in seen or nd < seen[u]:
                    seen[u] = nd
                    heapq.heappush(heap, (nd, cnt, u))
                    cnt += 1
        for v, d in dist.items():
            D[s, v] = d
    return D


def _n_components(K: int, adj) -> int:
    seen = set()
    n = 0
    for s in range(K):
        if s in seen:
            continue
        n += 1
        stack = [s]
        seen.add(s)
        while stack:
            v = stack.pop()
            for u in adj[v]:
                if u not in seen:
                    seen.add(u)
                    stack.append(u)
    return n


def _stable_order(values: np.ndarray) -> np.ndarray:
    """Ascending order, ties by index (the reference uses topk(largest=False) whose tie
    order is unspecified; any tie order yields the same kinematics)."""
    return np.lexsort((np.arange(values.size), values))


def build_tree(affinity: Tensor, big: float = 1e4):
    """process_affinity_glob: dyna_utils.py:6-171.  Returns
    (A (K,K) float64, order (K,) int64, order_values (K,) float64, parents (K,) int64)."""
    N, K = affinity.shape[:2]
    infl_t = affinity.detach().max(dim=0).values.squeeze(-1)          # (K,K)
    top = infl_t.topk(N, dim=-1).indices.numpy()
    infl = infl_t.numpy()
    A_bin = np.zeros((K, K), dtype=np.float32)
    A_bin[np.arange(K)[:, None], top] = 1
    A_bin = np.maximum(A_bin, A_bin.T)

    adj = _adjacency_lists(K, A_bin, None)
    D = _all_pairs(K, adj, big)
    if _n_components(K, adj) > 1:                                      # :37-67
        tot = D.sum(axis=-1)
        root = tot.argmin()
        rank = np.zeros(K)
        for r, i in enumerate(tot.copy().argsort()):
            rank[i] = r
        cand = np.where(D[root] == big)[0]
        pick = cand[0]
        for c in cand[1:]:
            if rank[pick] > rank[c]:
                pick = c
        A_bin[root, pick] = 1
        A_bin[pick, root] = 1
        adj = _adjacency_lists(K, A_bin, None)
        D = _all_pairs(K, adj, big)

    tot = D.sum(axis=-1)                                               # :70-81
    W = A_bin.copy()
    for k in range(K - 1):
        for kd in range(k + 1, K):
            if tot[k] == tot[kd]:
                ks = np.where(A_bin[k])[0]
                kds = np.where(A_bin[kd])[0]
                for n in ks:
                    if n in kds:
                        l = kd if infl[n, k] > infl[n, kd] else k
                        W[n, l] += 1e-5
                        W[l, n] += 1e-5
    D = _all_pairs(K, _adjacency_lists(K, A_bin, W), big)              # :83-97

    root = int(torch.from_numpy(D.sum(axis=-1)).topk(K, largest=False).indices[0])  # :101-102
    rank = D[root]
    parents = []
    for k in range(K):                                                 # :105-142
        if k == root:
            parents.append(k)
            continue
        nbrs = np.where(A_bin[k])[0]
        par, pd = None, -1e3
        for n in nbrs:
            rd = rank[n] - rank[k]
            if rd < 0 and rd > pd:
                pd, par = rd, n
            elif rd < 0 and rd == pd:
                if infl[k, n] > infl[k, par]:
                    pd, par = rd, n
            elif rd == 0:
                co, co_rank = None, 1e4
                for nn in np.where(A_bin[n])[0]:
                    if nn in nbrs and rank[nn] < rank[n]:
                        if co_rank > rank[nn]:
                            co, co_rank = nn, rank[nn]
                if co is not None and infl[co, n] > infl[co, k]:
                    pd, par = rd, n
        if par is None:
            par = root
            A_bin[k, par] = 1
            A_bin[par, k] = 1
        parents.append(int(par))
    parents = np.asarray(parents, dtype=np.int64)

    A = np.zeros((K, K))
    for k in range(K):
        if k != parents[k]:
            A[k, parents[k]] = 1
            A[parents[k], k] = 1
    D = _all_pairs(K, _adjacency_lists(K, A, W), big)                  # :152-169
    order = _stable_order(D[root]).astype(np.int64)
    return A, order, D[root][order], parents


# --------------------------------------------------------------------------------------
# VRNN (model/hsvrnn_bvh.py, utils/geo_utils.py)
# --------------------------------------------------------------------------------------
DYN = "dyna_module"


def _mlp(x: Tensor, sd: SD, p: str, tanh: bool = False) -> Tensor:
    h = F.leaky_relu(F.linear(x, sd[p + ".0.weight"], sd[p + ".0.bias"]), LRELU)
    y = F.linear(h, sd[p + ".2.weight"], sd[p + ".2.bias"])
    return torch.tanh(y) if tanh else y


def rot6d(p: Tensor) -> Tensor:
    """geo_utils.py:56-78: x = a/(|a|+1e-10), z = (x × b)/(|.|+1e-10), y = z × x,
    columns [x|y|z].  (...,6) -> (...,3,3)."""
    shp = p.shape[:-1]
    p = p.reshape(-1, 6)
    a, b = p[:, :3], p[:, 3:]

    def unit(v):
        return v / (torch.sqrt(v.pow(2).sum(1)) + 1e-10)[:, None]

    def cross(u, v):
        return torch.stack([u[:, 1] * v[:, 2] - u[:, 2] * v[:, 1],
                            u[:, 2] * v[:, 0] - u[:, 0] * v[:, 2],
                            u[:, 0] * v[:, 1] - u[:, 1] * v[:, 0]], dim=1)
    x = unit(a)
    z = unit(cross(x, b))
    y = cross(z, x)
    return torch.stack([x, y, z], dim=2).reshape(*shp, 3, 3)


def fk_decode(sd: SD, dec_in: Tensor, offset: Tensor, order: np.ndarray, parents: np.ndarray):
    """extract_kypt_from_latent_and_state: hsvrnn_bvh.py:255-286 (+ geo_utils.py:3-27).
    dec_in (B,H+Z), offset (B,K,3,1) -> flat keypoints (B,K*4), global R (B,K,3,3)."""
    B = dec_in.shape[0]
    K = offset.shape[1]
    raw = _mlp(dec_in, sd, DYN + ".root_intensity_decoder", tanh=True)
    root_pos = raw[:, :3]
    inten = ((raw[:, 3:] + 1) * 0.5)[..., None]
    Rl = rot6d(_mlp(dec_in, sd, DYN + ".joint_matrix_decoder").reshape(B, K, 6))
    Rg = [None] * K
    pos = torch.zeros(B, K, 3)
    root = int(order[0])
    Rg[root] = Rl[:, root]
    pos[:, root] = root_pos
    for idx in order[1:]:
        idx = int(idx); par = int(parents[idx])
        Rg[idx] = torch.bmm(Rg[par], Rl[:, idx])
    for idx in order[1:]:
        idx = int(idx); par = int(parents[idx])
        pos[:, idx] = torch.bmm(Rg[idx], offset[:, idx]).squeeze(-1) + pos[:, par]
    return torch.cat([pos, inten], dim=-1).reshape(B, -1), torch.stack(Rg, dim=1)


def bone_offsets(sd: SD, keypoints: Tensor, parents: np.ndarray) -> Tensor:
    """get_offset: hsvrnn_bvh.py:236-253 — lower median over T of pairwise distances,
    bone length = med[b,k,parent[k]], direction = unit(offset_param)."""
    pos = keypoints[..., :3]
    dist = (pos[:, :, :, None] - pos[:, :, None]).pow(2).sum(dim=-1).sqrt()
    med = dist.median(dim=1).values
    K = pos.shape[2]
    scale = torch.stack([med[:, k, int(parents[k])] for k in range(K)], dim=-1)
    op = sd[DYN + ".offset_param"]
    unit = op / (op.pow(2).sum(dim=-1, keepdim=True).sqrt() + 1e-10)
    return (unit[None] * scale[..., None])[..., None]


def _dist_params(raw: Tensor):
    mu, s = torch.chunk(raw, 2, dim=-1)
    return mu, F.softplus(s) + 1e-4


def _kl_normal(mq, sq, mp, sp):
    """torch.distributions.kl._kl_normal_normal(q, p)."""
    var_ratio = (sq / sp).pow(2)
    t1 = ((mq - mp) / sp).pow(2)
    return 0.5 * (var_ratio + t1 - 1 - var_ratio.log())


def _posterior_step(sd: SD, h: Tensor, kp_flat: Tensor, eps: Tensor, offset, order, parents):
    """One best-of-S posterior step: hsvrnn_bvh.py:99-128.  eps (S,B,Z)."""
    S = eps.shape[0]
    B = h.shape[0]
    mu, sig = _dist_params(_mlp(torch.cat([h, kp_flat], dim=-1), sd, DYN + ".extract_post_dist"))
    z = mu[None] + eps * sig[None]
    flats, Rs = [], []
    for i in range(S):
        f, R = fk_decode(sd, torch.cat([h, z[i]], dim=-1), offset, order, parents)
        flats.append(f); Rs.append(R)
    flats = torch.stack(flats, 0); Rs = torch.stack(Rs, 0)
    d = (kp_flat[None] - flats).pow(2).sum(-1)
    best = d.argmin(dim=0)
    ar = torch.arange(B)
    bz, bf, bR = z[best, ar], flats[best, ar], Rs[best, ar]
    h2 = gru_cell(sd, torch.cat([bf, bz], dim=-1), h)
    return h2, bz, bf, bR, best, d, (mu, sig)


def gru_cell(sd: SD, x: Tensor, h: Tensor) -> Tensor:
    """nn.GRUCell semantics (gate order r,z,n) written out the way ATen's CPU cell
    evaluates it: n = tanh(i_n + r * h_n), h' = (h - n) * z + n."""
    p = DYN + ".kypt_rnn_cell"
    gi = F.linear(x, sd[p + ".weight_ih"], sd[p + ".bias_ih"])
    gh = F.linear(h, sd[p + ".weight_hh"], sd[p + ".bias_hh"])
    ir, iz, in_ = gi.chunk(3, 1)
    hr, hz, hn = gh.chunk(3, 1)
    r = torch.sigmoid(ir + hr)
    zg = torch.sigmoid(iz + hz)
    n = torch.tanh(in_ + r * hn)
    return (h - n) * zg + n


def vrnn_encode(sd: SD, opts, keypoints: Tensor, order, parents, eps: Tensor) -> Dict[str, Tensor]:
    """HSVRNNBVH.encode: hsvrnn_bvh.py:67-156.  eps (T,S,B,Z) is the standard-normal draw
    of ``rsample((S,))`` at each step in t order."""
    B, T, K, _ = keypoints.shape
    h = sd[DYN + ".init_kypt_rnn_state"].expand(B, -1)
    offset = bone_offsets(sd, keypoints, parents)
    kps, Rs, zs, hs, kls, idx, dists = [], [], [], [h], [], [], []
    for t in range(T):
        pm, ps = _dist_params(_mlp(h, sd, DYN + ".extract_prior_dist"))
        flat = keypoints[:, t].reshape(B, -1)
        h2, bz, bf, bR, best, d, (qm, qs) = _posterior_step(sd, h, flat, eps[t], offset, order, parents)
        kls.append(_kl_normal(qm, qs, pm, ps))
        kps.append(bf.view(B, K, -1)); Rs.append(bR); zs.append(bz); hs.append(h2)
        idx.append(best); dists.append(d)
        h = h2
    kp_rec = torch.stack(kps, 1)
    return dict(
        kypt_recon=kp_rec[..., :4], R=torch.stack(Rs, 1), z_kypts=torch.stack(zs, 1),
        h_kypts=torch.stack(hs, 1), kl_kypt=torch.stack(kls, 1).mean(),
        kypt_recon_loss=(kp_rec - keypoints).pow(2).sum(dim=(2, 3)).mean(),
        best_idx=torch.stack(idx, 1), sample_dist=torch.stack(dists, 2), offset=offset,
    )


def vrnn_generate(sd: SD, opts, keypoints_cond: Tensor, order, parents, Ttot: int, Tcond: int,
                  eps_post: Tensor, eps_prior: Tensor) -> Dict[str, Tensor]:
    """HSVRNNBVH.generate: hsvrnn_bvh.py:158-234.  eps_post (Tcond,S,B,Z),
    eps_prior (Ttot-Tcond,B,Z)."""
    B, _, K, _ = keypoints_cond.shape
    h = sd[DYN + ".init_kypt_rnn_state"].expand(B, -1)
    offset = bone_offsets(sd, keypoints_cond, parents)
    cond, gen, hs, zs = [], [], [h], []
    for t in range(Tcond):
        flat = keypoints_cond[:, t].reshape(B, -1)
        h, bz, bf, _, _, _, _ = _posterior_step(sd, h, flat, eps_post[t], offset, order, parents)
        cond.append(bf.view(B, K, -1)); hs.append(h); zs.append(bz)
    for t in range(Tcond, Ttot):
        pm, ps = _dist_params(_mlp(h, sd, DYN + ".extract_prior_dist"))
        z = pm + eps_prior[t - Tcond] * ps
        f, _ = fk_decode(sd, torch.cat([h, z], dim=-1), offset, order, parents)
        h = gru_cell(sd, torch.cat([f, z], dim=-1), h)
        gen.append(f.view(B, K, -1)); hs.append(h); zs.append(z)
    return dict(keypoints_cond=torch.stack(cond, 1)[..., :4], keypoints_gen=torch.stack(gen, 1)[..., :4],
                h_last=h, h_seq=torch.stack(hs, 1), z_seq=torch.stack(zs, 1), offset=offset)


# --------------------------------------------------------------------------------------
# top level (model/neural_marionette.py)
# --------------------------------------------------------------------------------------
def nm_forward(sd: SD, opts, vox: Tensor, eps: Tensor, tree=None) -> Dict[str, Tensor]:
    """NeuralMarionette.forward with detector+learner active: neural_marionette.py:34-56."""
    log = detector_forward(sd, opts, vox, affinity_on=True)
    if tree is None:
        _, order, _, parents = build_tree(log["affinity"])
    else:
        order, parents = tree
    log.update(vrnn_encode(sd, opts, log["keypoints"], order, parents, eps))
    log["order"], log["parents"] = order, parents
    return log


def nm_generate(sd: SD, opts, vox: Tensor, order, parents, eps_post: Tensor, eps_prior: Tensor):
    """NeuralMarionette.generate ('dl'): neural_marionette.py:58-103."""
    Tc, T = opts.Tcond, vox.shape[1]
    det = detector_forward(sd, opts, vox[:, :Tc].contiguous(), affinity_on=True)
    g = vrnn_generate(sd, opts, det["keypoints"], order, parents, T, Tc, eps_post, eps_prior)
    gen = decode_from_keypoints(sd, opts, g["keypoints_gen"], det["first_feature"], vox[:, 0])
    return dict(gen=torch.cat([det["recon"][:, :Tc], gen], dim=1),
                keypoints=torch.cat([det["keypoints"][:, :Tc], g["keypoints_gen"]], dim=1),
                A_hats=None)


# --------------------------------------------------------------------------------------
# sampling drivers (vis_generation.py, vis_interpolation.py) — SURVEY 8(f3)
# --------------------------------------------------------------------------------------
def _prior_rows(sd: SD, h: Tensor, eps: Tensor, offset, order, parents):
    """one prior sample per row: hsvrnn_bvh.py:210-218 as used by the demo loops."""
    pm, ps = _dist_params(_mlp(h, sd, DYN + ".extract_prior_dist"))
    z = pm + eps * ps
    f, _ = fk_decode(sd, torch.cat([h, z], dim=-1), offset, order, parents)
    return f, z


def sample_generation(sd: SD, opts, cond_voxel: Tensor, Tgen: int, sample_num: int, eps_post: Tensor, eps_prior: Tensor,
                      order=None, parents=None):
    """vis_generation.py:81-136.  cond_voxel (Tcond,1,G,G,G); eps_post (Tcond,S,Z); eps_prior (Tgen,S,Z)."""
    S, K = sample_num, opts.nkeypoints
    det = detector_forward(sd, opts, cond_voxel[None])
    kp = det["keypoints"]
    if order is None:
        _, order, _, parents = build_tree(det["affinity"])
    Tc = kp.shape[1]
    h = sd[DYN + ".init_kypt_rnn_state"].expand(S, -1)
    off = bone_offsets(sd, kp, parents).expand(S, -1, -1, -1)
    cond, gen = [], []
    for t in range(Tc):                                                   # :97-115
        flat = kp[:, t].reshape(1, -1).expand(S, -1)
        mu, sig = _dist_params(_mlp(torch.cat([h, flat], dim=-1), sd, DYN + ".extract_post_dist"))
        z = mu + eps_post[t] * sig
        f, _ = fk_decode(sd, torch.cat([h, z], dim=-1), off, order, parents)
        i = int((f - flat).pow(2).sum(dim=-1).argmin())
        f, z, h = f[i][None].expand(S, -1), z[i][None].expand(S, -1), h[i][None].expand(S, -1)
        cond.append(flat[i].view(K, 4))
        h = gru_cell(sd, torch.cat([f, z], dim=-1), h)
    for t in range(Tgen):                                                 # :117-127
        f, z = _prior_rows(sd, h, eps_prior[t], off, order, parents)
        gen.append(f.view(S, K, 4))
        h = gru_cell(sd, torch.cat([f, z], dim=-1), h)
    cond_k, gen_k = torch.stack(cond, 0)[None], torch.stack(gen, 0)[None]
    vox = []
    for s in range(S):                                                    # :132-139
        full = torch.cat([cond_k, gen_k[:, :, s]], dim=1)
        vox.append(decode_from_keypoints(sd, opts, full, det["first_feature"], cond_voxel[None, 0])[0])
    vox = torch.stack(vox, 0)
    return dict(keypoints_cond=cond_k, keypoints_gen=gen_k, voxels=(vox >= 0.5).float(), voxels_raw=vox)


def sample_interpolation(sd: SD, opts, target_voxel: Tensor, sample_rate: int, sample_num: int, eps_a: Tensor, eps_b: Tensor,
                         order=None, parents=None, force_picks=None):
    """vis_interpolation.py:80-143.  eps_a / eps_b (T,S,Z): first / second normal draw of each step.
    Test aids (not in the reference): ``force_picks`` replaces the two argmin selections of every key frame (teacher forcing);
    ``margins`` reports, per key frame, the relative gap between the smallest and second smallest distance of both selections."""
    S, K = sample_num, opts.nkeypoints
    det = detector_forward(sd, opts, target_voxel[None])
    kp = det["keypoints"]
    if order is None:
        _, order, _, parents = build_tree(det["affinity"])
    T = kp.shape[1]
    h = sd[DYN + ".init_kypt_rnn_state"].expand(S, -1)
    off = bone_offsets(sd, kp, parents).expand(S, -1, -1, -1)
    selected, pending, picks, margins = [], [], [], []

    def _gap(d):
        two = torch.topk(d, 2, largest=False).values
        return float((two[1] - two[0]) / two[1].clamp_min(1e-30))
    for t in range(T):
        flat = kp[:, t].reshape(1, -1).expand(S, -1)
        if t % sample_rate == 0 or t == T - 1:                            # :99-122
            mu, sig = _dist_params(_mlp(torch.cat([h, flat], dim=-1), sd, DYN + ".extract_post_dist"))
            z = mu + eps_a[t] * sig
            f, _ = fk_decode(sd, torch.cat([h, z], dim=-1), off, order, parents)
            fc, _ = _prior_rows(sd, h, eps_b[t], off, order, parents)
            d1 = (f - flat).pow(2).sum(dim=-1)
            i1 = int(d1.argmin()) if force_picks is None else int(force_picks[len(picks)][0])
            f, z, h = f[i1][None].expand(S, -1), z[i1][None].expand(S, -1), h[i1][None].expand(S, -1)
            d2 = (fc - f).pow(2).sum(dim=-1)
            i2 = int(d2.argmin()) if force_picks is None else int(force_picks[len(picks)][1])
            margins.append((_gap(d1), _gap(d2)))
            pending.append(flat)
            selected += [s[i2].view(K, 4) for s in pending]
            pending = []
            picks.append((i1, i2))
        else:                                                             # :123-130
            f, z = _prior_rows(sd, h, eps_a[t], off, order, parents)
            pending.append(f)
        h = gru_cell(sd, torch.cat([f, z], dim=-1), h)
    sel = torch.stack(selected, 0)[None].clone()
    sel[0, :, :, -1] = sel[0, 0, :, -1]                                   # :133
    vox = decode_from_keypoints(sd, opts, sel, det["first_feature"], target_voxel[None, 0])[0]
    return dict(keypoints=sel, voxels=(vox >= 0.5).float(), voxels_raw=vox, picks=picks, margins=margins)


# ---- evaluation metrics --------------------------------------------------------------------
def voxel_chamfer_distance(gt_voxel, recon):
    """utils/eval_utils.py:29-55 without the in-place thresholding: per-frame chamfer distances (B, T) float64.
    gt_voxel, recon (B,T,1,G,G,G); occupied = non-zero / recon >= 0.5; coordinates idx / ((G-1)/2) - 1 in fp32."""
    B, T, _, *X = gt_voxel.shape
    gt = gt_voxel.squeeze(2)
    rec = (recon.squeeze(2) >= 0.5).float()
    out = torch.zeros(B, T, dtype=torch.float64)
    for b in range(B):
        for t in range(T):
            a = torch.stack(torch.where(gt[b, t]), dim=-1) / ((X[0] - 1) / 2) - 1          # (N, 3)   :42
            c = torch.stack(torch.where(rec[b, t]), dim=-1) / ((X[0] - 1) / 2) - 1         # (M, 3)   :43
            d = (a[:, None] - c[None]).pow(2).sum(dim=-1)                                  # :44
            out[b, t] = (d.min(dim=-1).values.mean() + d.min(dim=0).values.mean()).item()  # :45
    return out


def semantic_votes(keypoints, gt_keypoints):
    """utils/eval_utils.py:59-84 up to the vote matrix: closest (B*T, K') int64 and this batch's counts (K', K) int64.
    keypoints (B,T,K,4) (not modified), gt_keypoints (B,T,K',3)."""
    kypt = keypoints.clone()
    inval = torch.where(kypt[..., -1] < 0.2)                                               # :66
    kypt[inval] = torch.tensor([1e4, 1e4, 1e4, 1.0])
    det = kypt[:, :, None, :, :-1]                                                         # (B,T,1,K,3)
    B, T, Kg, _ = gt_keypoints.shape
    K = kypt.shape[2]
    d = (gt_keypoints[:, :, :, None] - det).pow(2).sum(-1)                                 # (B,T,K',K)  :77
    closest = d.min(dim=-1).indices.view(B * T, -1)                                        # :78
    counts = torch.zeros(Kg, K, dtype=torch.int64)
    for kd in range(Kg):
        counts[kd] = torch.bincount(closest[:, kd], minlength=K)
    return closest, counts


def semantic_log(counts):
    """scores_log of semantic_scores (:80-84): mean over k' of the largest vote share, float32."""
    c = counts.to(torch.float64).numpy()
    import numpy as _np
    return _np.array([(row / row.sum()).max() for row in c], dtype=_np.float32).mean()
