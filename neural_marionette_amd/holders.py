"""Parameter holders: the reference's module tree without its arithmetic.

``train.py:262,268`` of the reference runs ``network.apply(weights_init)`` on a freshly built network, and
``utils/train_utils.py:248-264`` picks its targets by *class name* ('Conv…' -> N(0, 0.02); '…Block' -> every
``nn.Conv3d`` / ``nn.ConvTranspose3d`` below it -> N(0, 0.001), bias 0) and by ``isinstance`` on the torch layer
types.  For that call — and torch's default initialisation before it — to act on these shells exactly as on the
reference tree, the holders below

* are subclasses of the torch layer types (same constructors, same default init, same RNG consumption),
* are created in the reference's construction order (vox_modules.py:8-96, kypt_detector.py:244-297, 369-460,
  hsvrnn_bvh.py:12-65), and
* sit in containers carrying the reference's class and attribute names, so ``state_dict`` keeps its 337 keys.

None of them computes: ``forward`` raises.  The arithmetic of this path lives in libnm355.so only.
"""
from __future__ import annotations

import torch
from torch import nn

from . import _lib
from .spec import FEAT_DIM, HG_WIDTHS, VRNN_MLP_HIDDEN


def _refuse(self, *args, **kwargs):
    raise _lib.NmError(f"{type(self).__name__} is a parameter holder; the computation lives in libnm355.so "
                       "(call the owning KyptDetector / HSVRNNBVH / NeuralMarionette instead)")


class Conv3d(nn.Conv3d):
    forward = _refuse


class ConvTranspose3d(nn.ConvTranspose3d):
    forward = _refuse


class GroupNorm(nn.GroupNorm):
    forward = _refuse


class Linear(nn.Linear):
    forward = _refuse


class Sequential(nn.Sequential):
    """Index-named container ('0', '1', ...), never called."""
    forward = _refuse


class _Gap(nn.Module):
    """Parameter-free position of a Sequential (LeakyReLU / Softplus / Tanh / Upsample in the reference):
    keeps the indices of the layers after it."""

    def __init__(self, what: str):
        super().__init__()
        self.what = what

    def extra_repr(self):
        return self.what

    forward = _refuse


def _unit(conv: nn.Module, channels: int, act: bool) -> Sequential:
    """conv -> GroupNorm(C/16 groups) [-> LeakyReLU]: the repeating unit of vox_modules.py."""
    layers = [conv, GroupNorm(channels // 16, channels)]
    if act:
        layers.append(_Gap("LeakyReLU(0.01)"))
    return Sequential(*layers)


class Basic3DBlock(nn.Module):
    """vox_modules.py:8-19."""

    def __init__(self, in_planes: int, out_planes: int, kernel_size: int):
        super().__init__()
        self.block = _unit(Conv3d(in_planes, out_planes, kernel_size, 1, (kernel_size - 1) // 2), out_planes, True)

    forward = _refuse


class Res3DBlock(nn.Module):
    """vox_modules.py:22-47."""

    def __init__(self, in_planes: int, out_planes: int):
        super().__init__()
        first = _unit(Conv3d(in_planes, out_planes, 3, 1, 1), out_planes, True)
        second = _unit(Conv3d(out_planes, out_planes, 3, 1, 1), out_planes, False)
        self.res_branch = Sequential(*first, *second)             # indices 0,1,(2),3,4
        self.skip_con = Sequential() if in_planes == out_planes else _unit(Conv3d(in_planes, out_planes, 1, 1, 0), out_planes, False)

    forward = _refuse


class Pool3DBlock(nn.Module):
    """vox_modules.py:49-61."""

    def __init__(self, pool_size: int, planes: int):
        super().__init__()
        self.stride_conv = _unit(Conv3d(planes, planes, pool_size, pool_size, 0), planes, True)

    forward = _refuse


class Upsample3DBlock(nn.Module):
    """vox_modules.py:63-75."""

    def __init__(self, in_planes: int, out_planes: int, kernel_size: int, stride: int, output_padding: int = 0):
        super().__init__()
        if stride != 2:
            raise ValueError("Upsample3DBlock: stride must be 2")
        self.block = _unit(ConvTranspose3d(in_planes, out_planes, kernel_size, stride, 0, output_padding), out_planes, True)

    forward = _refuse


class HG(nn.Module):
    """vox_modules.py:78-96: the hourglass' sixteen blocks, in the order the reference creates them."""

    def __init__(self, input_channels: int, output_channels: int, N: int):
        super().__init__()
        w = (input_channels,) + HG_WIDTHS
        pad = {3: (N // 4) % 2, 2: (N // 2) % 2, 1: N % 2}
        for lvl in (1, 2, 3):
            setattr(self, f"encoder_pool{lvl}", Pool3DBlock(2, w[lvl - 1]))
            setattr(self, f"encoder_res{lvl}", Res3DBlock(w[lvl - 1], w[lvl]))
        for lvl in (3, 2, 1):
            setattr(self, f"decoder_res{lvl}", Res3DBlock(w[lvl], w[lvl]))
            setattr(self, f"decoder_upsample{lvl}",
                    Upsample3DBlock(w[lvl], output_channels if lvl == 1 else w[lvl - 1], 2, 2, pad[lvl]))
        for lvl in (1, 2, 3):
            setattr(self, f"skip_res{lvl}", Res3DBlock(w[lvl - 1], output_channels if lvl == 1 else w[lvl - 1]))

    forward = _refuse


def _feature_net(in_channels: int, out_channels: int, grid_size: int) -> Sequential:
    """kypt_detector.py:264-272."""
    q, h = out_channels // 4, out_channels // 2
    return Sequential(Basic3DBlock(1 + in_channels, q, 5), Pool3DBlock(2, q), Res3DBlock(q, h), Pool3DBlock(2, h),
                      HG(h, h, N=grid_size // 4), Res3DBlock(h, out_channels))


def _head(in_channels: int, out_channels: int, act: str) -> Sequential:
    """kypt_detector.py:273-280 / :295-297: 1x1x1 conv + activation."""
    return Sequential(Conv3d(in_channels, out_channels, 1, 1, 0), _Gap(act))


class VoxToKyptNet(nn.Module):
    """kypt_detector.py:244-297 (const_intensity = 3, fixed sigmas)."""

    def __init__(self, grid_size: int, nkeypoints: int, input_dim: int, sigmas, fixed_sigma: bool, const_intensity: int):
        super().__init__()
        self.grid_size, self.feat_dim, self.nkeypoints = grid_size, FEAT_DIM, nkeypoints
        self.fixed_sigma, self.const_intensity = fixed_sigma, const_intensity
        if fixed_sigma:
            self.sigmas = sigmas
        else:                                               # kypt_detector.py:258-260
            self.max_sigma = sigmas[0] * 2.0
            self.sigmas = nn.Parameter(torch.randn(nkeypoints))
        self.extract_features = _feature_net(input_dim, FEAT_DIM, grid_size)
        self.extract_heatmaps_from_features = _head(FEAT_DIM, nkeypoints, "LeakyReLU(0.01)")
        self.extract_spatio_temporal_features = _feature_net(input_dim, 2 * FEAT_DIM, grid_size)
        self.extract_spatio_temporal_heatmaps_from_features = _head(2 * FEAT_DIM, nkeypoints, "LeakyReLU(0.01)")
        self.propagate_heatmaps = _head(2, 1, "Softplus")

    forward = _refuse


class KyptToVoxNet(nn.Module):
    """kypt_detector.py:369-386, 417-460."""

    def __init__(self, grid_size: int, nkeypoints: int, input_dim: int, gaussian_cat_type: str):
        super().__init__()
        self.grid_size, self.output_map_width = grid_size, grid_size // 4
        self.feat_dim, self.nkeypoints, self.gaussian_cat_type = FEAT_DIM, nkeypoints, gaussian_cat_type
        self.adjust_combined_representation = _head(FEAT_DIM + 2 * nkeypoints + input_dim, FEAT_DIM, "LeakyReLU(0.01)")
        layers, c = [], FEAT_DIM
        for _ in range(2):                                     # two resolution doublings, channels halved at each
            layers.append(_Gap("Upsample(x2, trilinear, align_corners=False)"))
            layers += list(_unit(Conv3d(c, c // 2, 3, 1, 1), c // 2, True))
            layers += list(_unit(Conv3d(c // 2, c // 2, 3, 1, 1), c // 2, True))
            c //= 2
        layers.append(Conv3d(c, 1, 1))
        self.decode_voxel_from_combined_representation = Sequential(*layers)

    forward = _refuse


def vrnn_mlp_layers(nin: int, nout: int, tanh: bool):
    """hsvrnn_bvh.py:29-54: Linear(nin,128) -> LeakyReLU -> Linear(128,nout) [-> Tanh]."""
    layers = [Linear(nin, VRNN_MLP_HIDDEN), _Gap("LeakyReLU(0.01)"), Linear(VRNN_MLP_HIDDEN, nout)]
    if tanh:
        layers.append(_Gap("Tanh"))
    return layers
