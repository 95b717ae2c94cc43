"""Architecture tables for the Neural Marionette hot path.

Everything the host side needs to know about the network is derived from the two
small generators below: the ordered list of (state_dict key, shape) pairs and the
hot-path option record.  The HIP library gets its layer graph from the same
description (see ``csrc/nm_net.hip``), so the table is the single source of truth
for checkpoint compatibility.

Reference for names/shapes (state_dict of the reference module, 337 tensors,
10 087 015 parameters): modules/vox_modules.py:8-120, model/kypt_detector.py:14-69,
244-297, 369-460, model/hsvrnn_bvh.py:12-65.
"""
from __future__ import annotations

from dataclasses import dataclass, field, asdict
from typing import Dict, Iterator, List, Tuple

Shape = Tuple[int, ...]

FEAT_DIM = 128          # kypt_detector.py:253 / :375
HG_WIDTHS = (32, 48, 72)  # vox_modules.py:83-87
VRNN_MLP_HIDDEN = 128   # hsvrnn_bvh.py:31,37,43,51


@dataclass
class HotPathOptions:
    """The subset of the reference's argparse Namespace the hot path reads
    (kypt_detector.py:18-68, hsvrnn_bvh.py:14-20, neural_marionette.py:11,15).
    Defaults are the values stored in pretrained/aist/opt.pickle."""
    grid_size: int = 64
    nkeypoints: int = 24
    input_dim: int = 3
    gaussian_sigma: float = 1.5
    fixed_sigma: int = 1
    const_intensity: int = 3
    affinity_ver: int = 3
    nneighbor: int = 2
    graph_loss_ver: int = 1
    gaussian_cat_type: str = "none"
    vol_fit_type: str = "chamfer"
    keypoints_graph: str = "affinity_params"
    graph_random_init: int = 0
    keypoints_detach: int = 0
    sep_sigma: float = 0.02
    nlatent_kypt: int = 128
    nhidden_kypt: int = 512
    transition_type: str = "dl"
    Ttot: int = 20
    Tcond: int = 5
    is_binarized: int = 1
    affinity_anneal: int = 0
    using_local_const: int = 1
    using_time_const: int = 1
    using_sparsity_const: int = 1
    using_intensity_const: int = 1
    graph_traj_weight: float = 1.0
    graph_vol_weight: float = 0.0
    state_mode: str = "no_cat"
    action_mode: str = "pose"

    @classmethod
    def from_any(cls, obj) -> "HotPathOptions":
        """Accept an argparse.Namespace / attribute bag / dict / HotPathOptions."""
        if isinstance(obj, cls):
            return cls(**asdict(obj))
        if obj is None:
            return cls()
        get = (lambda k, d: obj.get(k, d)) if isinstance(obj, dict) else (lambda k, d: getattr(obj, k, d))
        base = cls()
        return cls(**{k: get(k, v) for k, v in asdict(base).items()})

    def check_fast_path(self) -> None:
        """The HIP path implements the pretrained-AIST configuration family
        (SURVEY §5 'Config / flags'); anything else is rejected loudly instead of
        silently computing something different."""
        bad = []
        if self.input_dim != 3: bad.append("input_dim must be 3")
        if self.const_intensity != 3: bad.append("const_intensity must be 3")
        if self.affinity_ver not in (0, 1, 2, 3): bad.append("affinity_ver must be 0, 1, 2 or 3 (4 draws Gumbel noise: not implemented)")
        if self.graph_loss_ver != 1: bad.append("graph_loss_ver must be 1")
        if self.gaussian_cat_type not in ("none", "max", "sum"): bad.append("gaussian_cat_type must be 'none', 'max' or 'sum'")
        if self.vol_fit_type not in ("chamfer", "none", "gaussian"): bad.append("vol_fit_type must be chamfer / none / gaussian")
        if self.keypoints_graph != "affinity_params": bad.append("keypoints_graph must be 'affinity_params'")
        if not self.fixed_sigma and self.vol_fit_type == "gaussian": bad.append("fixed_sigma = 0 with vol_fit_type 'gaussian' is not implemented")
        if self.transition_type != "dl": bad.append("transition_type must be 'dl'")
        if self.grid_size % 8 != 0 or self.grid_size < 32:
            bad.append("grid_size must be a multiple of 8 and >= 32")
        if bad:
            raise NotImplementedError("neural_marionette_amd HIP path: " + "; ".join(bad))


# --------------------------------------------------------------------------------------
# state_dict layout
# --------------------------------------------------------------------------------------
def _conv(p: str, co: int, ci: int, k: int) -> Iterator[Tuple[str, Shape]]:
    yield p + ".weight", (co, ci, k, k, k)
    yield p + ".bias", (co,)


def _gn(p: str, c: int) -> Iterator[Tuple[str, Shape]]:
    yield p + ".weight", (c,)
    yield p + ".bias", (c,)


def _res(p: str, ci: int, co: int) -> Iterator[Tuple[str, Shape]]:
    yield from _conv(p + ".res_branch.0", co, ci, 3)
    yield from _gn(p + ".res_branch.1", co)
    yield from _conv(p + ".res_branch.3", co, co, 3)
    yield from _gn(p + ".res_branch.4", co)
    if ci != co:
        yield from _conv(p + ".skip_con.0", co, ci, 1)
        yield from _gn(p + ".skip_con.1", co)


def _pool(p: str, c: int) -> Iterator[Tuple[str, Shape]]:
    yield from _conv(p + ".stride_conv.0", c, c, 2)
    yield from _gn(p + ".stride_conv.1", c)


def _up(p: str, ci: int, co: int) -> Iterator[Tuple[str, Shape]]:
    yield p + ".block.0.weight", (ci, co, 2, 2, 2)   # ConvTranspose3d: (in, out, k, k, k)
    yield p + ".block.0.bias", (co,)
    yield from _gn(p + ".block.1", co)


def _hourglass(p: str, ci: int, co: int) -> Iterator[Tuple[str, Shape]]:
    w1, w2, w3 = HG_WIDTHS
    yield from _pool(p + ".encoder_pool1", ci)
    yield from _res(p + ".encoder_res1", ci, w1)
    yield from _pool(p + ".encoder_pool2", w1)
    yield from _res(p + ".encoder_res2", w1, w2)
    yield from _pool(p + ".encoder_pool3", w2)
    yield from _res(p + ".encoder_res3", w2, w3)
    yield from _res(p + ".decoder_res3", w3, w3)
    yield from _up(p + ".decoder_upsample3", w3, w2)
    yield from _res(p + ".decoder_res2", w2, w2)
    yield from _up(p + ".decoder_upsample2", w2, w1)
    yield from _res(p + ".decoder_res1", w1, w1)
    yield from _up(p + ".decoder_upsample1", w1, co)
    yield from _res(p + ".skip_res1", ci, co)
    yield from _res(p + ".skip_res2", w1, w1)
    yield from _res(p + ".skip_res3", w2, w2)


def _feature_net(p: str, cin: int, cout: int) -> Iterator[Tuple[str, Shape]]:
    c4, c2 = cout // 4, cout // 2
    yield from _conv(p + ".0.block.0", c4, 1 + cin, 5)
    yield from _gn(p + ".0.block.1", c4)
    yield from _pool(p + ".1", c4)
    yield from _res(p + ".2", c4, c2)
    yield from _pool(p + ".3", c2)
    yield from _hourglass(p + ".4", c2, c2)
    yield from _res(p + ".5", c2, cout)


def _linear(p: str, co: int, ci: int) -> Iterator[Tuple[str, Shape]]:
    yield p + ".weight", (co, ci)
    yield p + ".bias", (co,)


def param_spec(opts: HotPathOptions) -> List[Tuple[str, Shape]]:
    """Ordered (key, shape) list identical to ``NeuralMarionette(opt).state_dict()``
    of the reference for the supported configuration family."""
    K, D, Z, H = opts.nkeypoints, opts.input_dim, opts.nlatent_kypt, opts.nhidden_kypt
    F = FEAT_DIM
    out: List[Tuple[str, Shape]] = []
    d = "kypt_detector"
    out.append((d + ".affinity_params", (opts.nneighbor, K, K if opts.affinity_ver < 3 else K - 1)))      # kypt_detector.py:57-68
    v = d + ".vox_to_kypt"
    if not opts.fixed_sigma:
        out.append((v + ".sigmas", (K,)))                  # kypt_detector.py:258-260: created before the sub-modules
    out += list(_feature_net(v + ".extract_features", D, F))
    out += list(_conv(v + ".extract_heatmaps_from_features.0", K, F, 1))
    out += list(_feature_net(v + ".extract_spatio_temporal_features", D, 2 * F))
    out += list(_conv(v + ".extract_spatio_temporal_heatmaps_from_features.0", K, 2 * F, 1))
    out += list(_conv(v + ".propagate_heatmaps.0", 1, 2, 1))
    k2v = d + ".kypt_to_vox"
    out += list(_conv(k2v + ".adjust_combined_representation.0", F, F + 2 * K + D, 1))
    dec = k2v + ".decode_voxel_from_combined_representation"
    out += list(_conv(dec + ".1", F // 2, F, 3)); out += list(_gn(dec + ".2", F // 2))
    out += list(_conv(dec + ".4", F // 2, F // 2, 3)); out += list(_gn(dec + ".5", F // 2))
    out += list(_conv(dec + ".8", F // 4, F // 2, 3)); out += list(_gn(dec + ".9", F // 4))
    out += list(_conv(dec + ".11", F // 4, F // 4, 3)); out += list(_gn(dec + ".12", F // 4))
    out += list(_conv(dec + ".14", 1, F // 4, 1))
    m = "dyna_module"
    S = K * (D + 1)
    out.append((m + ".init_kypt_rnn_state", (1, H)))
    out.append((m + ".offset_param", (K, 3)))
    hid = VRNN_MLP_HIDDEN
    out += list(_linear(m + ".extract_post_dist.0", hid, H + S))
    out += list(_linear(m + ".extract_post_dist.2", 2 * Z, hid))
    out += list(_linear(m + ".extract_prior_dist.0", hid, H))
    out += list(_linear(m + ".extract_prior_dist.2", 2 * Z, hid))
    out += list(_linear(m + ".root_intensity_decoder.0", hid, H + Z))
    out += list(_linear(m + ".root_intensity_decoder.2", 3 + K, hid))
    out += list(_linear(m + ".joint_matrix_decoder.0", hid, H + Z))
    out += list(_linear(m + ".joint_matrix_decoder.2", 6 * K, hid))
    out.append((m + ".kypt_rnn_cell.weight_ih", (3 * H, S + Z)))
    out.append((m + ".kypt_rnn_cell.weight_hh", (3 * H, H)))
    out.append((m + ".kypt_rnn_cell.bias_ih", (3 * H,)))
    out.append((m + ".kypt_rnn_cell.bias_hh", (3 * H,)))
    return out


def param_count(opts: HotPathOptions) -> int:
    n = 0
    for _, shp in param_spec(opts):
        c = 1
        for s in shp:
            c *= s
        n += c
    return n


# Keys whose parameters are created with requires_grad=False in the reference
# (hsvrnn_bvh.py:64-65).
FROZEN_KEYS = ("dyna_module.offset_param",)

# order of the 11 scalar losses the detector entry point fills (kypt_detector.py:155-165)
DETECTOR_LOSS_KEYS = (
    "recon_loss", "vol_fit_reg", "kypt_const_loss", "separation_loss", "sparsity_loss",
    "local_const_loss", "time_const_loss", "sparsity_const_loss", "intensity_const_loss",
    "graph_traj_loss", "graph_vol_loss",
)
