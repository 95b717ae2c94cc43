"""Seeded synthetic weights and occupancy clips (no datasets / checkpoints offline).

Both generators are pure numpy (PCG64) so that the very same tensors can be
re-created on the GPU box from a seed — golden fixtures only need to carry seeds
plus small expected outputs.
"""
from __future__ import annotations

import math
from typing import Dict

import numpy as np
import torch

from .spec import HotPathOptions, param_spec


def _fan_in(shape) -> int:
    n = 1
    for s in shape[1:]:
        n *= s
    return max(n, 1)


def make_state_dict(opts: HotPathOptions, seed: int = 0, variant: str = "default",
                    dtype=torch.float32) -> Dict[str, torch.Tensor]:
    """Seeded weights for every key of :func:`param_spec`.

    variant:
      ``default``  U(-b, b), b = 1/sqrt(fan_in) for conv/linear weights and biases
                   (torch's default init scale), GroupNorm affine = 1 + 0.1 N / 0.1 N,
                   random affinity logits.
      ``peaky``    like ``default`` but the heat-map heads are scaled up so the
                   detected keypoints spread over [-1, 1] instead of hugging 0.
      ``tracking`` ``peaky`` with the occupancy channel of both first layers (input channel 0 of the k5 convs) scaled by 10: the
                   keypoints follow the figure (frame-to-frame keypoint velocities ~3e-2 on the figure clips instead of 1e-5 ... 1e-2),
                   which makes the trajectory term of the graph loss well conditioned (fixture G12).
      ``winit``    what a from-scratch detector run starts from (train.py:262,268 ->
                   utils/train_utils.py:248-264 of the reference): convs inside the
                   ``*Block`` containers (both feature nets) N(0, 0.001), the other convs
                   (heads, combined-representation adjust, decoder) N(0, 0.02), conv biases 0,
                   GroupNorm affine exactly 1 / 0, VRNN at torch's default scale; the
                   affinity logits stay random (the all-ones init is a tree of ties, fixture G3).
    """
    rng = np.random.default_rng(np.random.SeedSequence([seed, 0x4E4D]))
    sd: Dict[str, torch.Tensor] = {}
    for name, shape in param_spec(opts):
        leaf = name.rsplit(".", 1)[-1]
        is_norm = len(shape) == 1 and leaf in ("weight", "bias") and _is_norm_key(name)
        if name.endswith("affinity_params"):
            a = rng.standard_normal(shape)
        elif name.endswith("init_kypt_rnn_state") or name.endswith("offset_param"):
            a = rng.standard_normal(shape)
        elif is_norm and variant == "winit":
            a = np.ones(shape) if leaf == "weight" else np.zeros(shape)
        elif is_norm:
            a = (1.0 + 0.1 * rng.standard_normal(shape)) if leaf == "weight" else 0.1 * rng.standard_normal(shape)
        elif variant == "winit" and len(shape) == 5:
            a = (0.001 if _in_block(name) else 0.02) * rng.standard_normal(shape)
        elif variant == "winit" and leaf == "bias" and not name.startswith("dyna_module"):
            a = np.zeros(shape)
        else:
            if leaf.startswith("bias"):
                # bias bound uses the fan-in of the matching weight; approximating it
                # with a fixed small bound keeps the table-free generator simple.
                b = 0.05
            else:
                b = 1.0 / math.sqrt(_fan_in(shape))
            a = rng.uniform(-b, b, size=shape)
        if variant in ("peaky", "tracking") and ("heatmaps_from_features.0.weight" in name):
            a = a * 12.0
        if variant == "tracking" and name.endswith("features.0.block.0.weight"):
            a[:, 0] = a[:, 0] * 10.0
        sd[name] = torch.from_numpy(np.ascontiguousarray(a)).to(dtype)
    return sd


def _in_block(name: str) -> bool:
    """True for the convs below a Basic3DBlock / Res3DBlock / Pool3DBlock / Upsample3DBlock, i.e. everything in the two
    feature nets (kypt_detector.py:264-272); the heads, the adjust conv and the decoder are plain nn.Conv3d."""
    return ".extract_features." in name or ".extract_spatio_temporal_features." in name


def _is_norm_key(name: str) -> bool:
    # GroupNorm affine parameters sit at these sequential indices in the reference tree
    tails = (".block.1.", ".stride_conv.1.", ".res_branch.1.", ".res_branch.4.", ".skip_con.1.")
    if any(t in name + "." for t in tails):
        return True
    for idx in (2, 5, 9, 12):
        if (".decode_voxel_from_combined_representation.%d." % idx) in name + ".":
            return True
    return False


def bernoulli_clip(B: int, T: int, G: int, p: float = 0.03, seed: int = 1) -> torch.Tensor:
    """vox = (U[0,1) < p), shape (B, T, 1, G, G, G) fp32 (SURVEY §8(d) 'bernoulli')."""
    rng = np.random.default_rng(np.random.SeedSequence([seed, 0xB0C5]))
    v = (rng.random((B, T, 1, G, G, G), dtype=np.float32) < p).astype(np.float32)
    return torch.from_numpy(v)


# ------------------------------------------------------------------------------------
# articulated "figure" clips: capsules around a random stick figure, sampled as a
# point cloud and pushed through the restated input path of the reference
# (utils/dataset_utils.py:9-31: per-episode bbox normalisation + voxelize).
# ------------------------------------------------------------------------------------
_BONES = [(-1, 0.25), (0, 0.22), (1, 0.12), (1, 0.18), (3, 0.20), (4, 0.18),
          (1, 0.18), (6, 0.20), (7, 0.18), (0, 0.30), (9, 0.30), (0, 0.30), (11, 0.30)]


def _rot(axis, ang):
    axis = axis / (np.linalg.norm(axis) + 1e-12)
    x, y, z = axis
    c, s = math.cos(ang), math.sin(ang)
    C = 1 - c
    return np.array([[c + x * x * C, x * y * C - z * s, x * z * C + y * s],
                     [y * x * C + z * s, c + y * y * C, y * z * C - x * s],
                     [z * x * C - y * s, z * y * C + x * s, c + z * z * C]])


def figure_points(T: int, npts: int, rng) -> np.ndarray:
    """(T, npts, 3) point cloud of a smoothly moving capsule figure."""
    nb = len(_BONES)
    rest = rng.standard_normal((nb, 3))
    rest /= np.linalg.norm(rest, axis=1, keepdims=True)
    axes = rng.standard_normal((nb, 3))
    freq = rng.uniform(0.5, 2.0, nb)
    phase = rng.uniform(0, 2 * math.pi, nb)
    amp = rng.uniform(0.2, 0.9, nb)
    which = rng.integers(0, nb, npts)
    u = rng.random(npts)
    off = rng.standard_normal((npts, 3))
    off /= np.linalg.norm(off, axis=1, keepdims=True)
    out = np.zeros((T, npts, 3))
    for t in range(T):
        R = [None] * nb
        head = np.zeros((nb, 3)); tail = np.zeros((nb, 3))
        for j, (par, ln) in enumerate(_BONES):
            Rl = _rot(axes[j], amp[j] * math.sin(freq[j] * 0.35 * t + phase[j]))
            R[j] = Rl if par < 0 else R[par] @ Rl
            head[j] = (0.05 * np.array([math.sin(0.2 * t), 0, math.cos(0.3 * t)])) if par < 0 else tail[par]
            tail[j] = head[j] + R[j] @ rest[j] * ln
        p = head[which] * (1 - u)[:, None] + tail[which] * u[:, None]
        out[t] = p + 0.045 * off
    return out


def episodic_normalization(pts: np.ndarray, scale: float = 1.0) -> np.ndarray:
    """Restates utils/dataset_utils.py:9-19 of the reference (zero translation):
    shift by the episode's bbox minimum, scale by the largest bbox edge (+1e-5),
    map to [-1, 1); float64 arithmetic."""
    bmax = pts.max(axis=(0, 1))
    bmin = pts.min(axis=(0, 1))
    edge = (bmax - bmin).max()
    return ((pts - bmin[None, None]) * scale / (edge + 1e-5)) * 2 - 1


def voxel_indices(pts: np.ndarray, G: int) -> np.ndarray:
    """Restates the index arithmetic of utils/dataset_utils.py:25-29:
    step = 2/G per axis, idx = ((p - (-1)) / (step + 1e-5)).astype(int32)."""
    step = np.full(3, 2.0) / np.array([G, G, G])
    return ((pts[..., :3] + 1.0) / (step + 1e-5)).astype(np.int32)


def voxelize(pts: np.ndarray, G: int) -> np.ndarray:
    """Restates utils/dataset_utils.py:21-31 — scatter-set into a (G,G,G) fp32 grid."""
    idx = voxel_indices(pts, G)
    vox = np.zeros((G, G, G), dtype=np.float32)
    vox[idx[:, 0], idx[:, 1], idx[:, 2]] = 1.0
    return vox


def figure_clip(B: int, T: int, G: int, seed: int = 1, npts: int = 20000) -> torch.Tensor:
    out = np.zeros((B, T, 1, G, G, G), dtype=np.float32)
    for b in range(B):
        rng = np.random.default_rng(np.random.SeedSequence([seed, b, 0xF16]))
        pts = episodic_normalization(figure_points(T, npts, rng), scale=0.9)
        for t in range(T):
            out[b, t, 0] = voxelize(pts[t], G)
    return torch.from_numpy(out)


def make_eps(shape, seed: int) -> torch.Tensor:
    """Standard-normal noise with an explicit seed (the VRNN sampling entry points take
    eps explicitly; CPU and GPU generator streams differ, SURVEY §7 'RNG')."""
    rng = np.random.default_rng(np.random.SeedSequence([seed, 0xE95]))
    return torch.from_numpy(rng.standard_normal(shape).astype(np.float32))


def eval_inputs(seed=7, B=2, T=3, G=32, K=24, Kg=17):
    """Seeded inputs of the evaluation metrics (fixture g7 holds the reference's outputs on them)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    vox = figure_clip(B, T, G, seed=seed)                                                # (B,T,1,G,G,G)
    noise = torch.from_numpy(rng.random(vox.shape, dtype=np.float32))
    recon = (0.75 * torch.roll(vox, shifts=(1, -1), dims=(-1, -2)) + 0.35 * noise).clamp(0, 1)   # a shifted, noisy volume
    kp = torch.from_numpy(rng.uniform(-1, 1, size=(B, T, K, 4)).astype(np.float32))
    kp[..., 3] = torch.from_numpy(rng.uniform(0, 1, size=(B, T, K)).astype(np.float32))          # intensities, some < 0.2
    gt = torch.from_numpy(rng.uniform(-1, 1, size=(B, T, Kg, 3)).astype(np.float32))
    return vox, recon, kp, gt
