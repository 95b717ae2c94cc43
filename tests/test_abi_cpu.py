"""CPU-side checks of the C-ABI library and the host mirror: the .so loads without a GPU,
exports every symbol include/nm355.h declares, and the module shells keep the reference's
state_dict layout.  No compute call is made here."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

from neural_marionette_amd import _lib, NeuralMarionette, HotPathOptions, param_spec, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "nm355.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(nm_[A-Za-z0-9_]+)\s*\(", hdr)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    syms = _declared_symbols()
    assert len(syms) >= 25
    for s in syms:
        assert hasattr(lib, s), f"libnm355.so does not export {s}"
        assert s in _lib.SIGNATURES, f"{s} has no ctypes signature"
    assert sorted(_lib.SIGNATURES) == syms
    assert lib.nm_abi_version() == 1


def test_host_linspace_matches_torch_bitwise():
    lib = _lib.load()
    for n in (2, 3, 8, 10, 16, 24, 40, 64, 88, 96):
        buf = (C.c_float * n)()
        assert lib.nm_host_linspace(n, buf) == 0
        assert np.array_equal(np.frombuffer(buf, dtype=np.float32), torch.linspace(-1, 1, n).numpy()), n
    assert lib.nm_host_linspace(1, (C.c_float * 1)()) != 0
    assert b"linspace" in lib.nm_last_error()


def test_state_dict_layout_and_loading():
    o = HotPathOptions()
    net = NeuralMarionette(o)
    keys = [(k, tuple(v.shape)) for k, v in net.state_dict().items()]
    assert keys == [(k, tuple(s)) for k, s in param_spec(o)]
    assert len(keys) == 337 and sum(v.numel() for v in net.state_dict().values()) == 10087015
    sd = synth.make_state_dict(o, seed=1)
    assert not net.load_state_dict(sd).missing_keys
    assert torch.equal(net.kypt_detector.affinity_params, sd["kypt_detector.affinity_params"])
    assert net.dyna_module.offset_param.requires_grad is False
    assert net.dyna_module.kypt_rnn_cell.weight_ih.shape == (1536, 224)
    # control_active / anneal bookkeeping (neural_marionette.py:18-32, kypt_detector.py:71-78)
    net.control_active({"detector": False, "learner": True})
    assert not any(p.requires_grad for p in net.kypt_detector.parameters())
    net.control_active({"detector": True, "learner": True})
    assert all(p.requires_grad for p in net.kypt_detector.parameters())
    assert net.kypt_detector.affinity_start is False
    net.anneal(0)
    assert net.kypt_detector.affinity_start is True


def test_no_cpu_fallback():
    net = NeuralMarionette(HotPathOptions(grid_size=32))
    with pytest.raises(_lib.NmError):
        net(torch.zeros(1, 2, 1, 32, 32, 32), {"detector": True, "learner": False})
    with pytest.raises(NotImplementedError):
        NeuralMarionette(HotPathOptions(const_intensity=1))


def test_options_from_namespace():
    import argparse
    ns = argparse.Namespace(grid_size=64, nkeypoints=24, Tcond=5, transition_type="dl", unknown_flag=3)
    o = HotPathOptions.from_any(ns)
    assert o.Tcond == 5 and o.gaussian_sigma == 1.5
    assert HotPathOptions.from_any({"grid_size": 96}).grid_size == 96
