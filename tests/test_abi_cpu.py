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


def test_no_cpp_exception_crosses_the_abi():
    """SURVEY 8(b) "no C++ exceptions across the ABI": nm_abi_selftest_throw raises a real std::bad_alloc (an allocation request of
    SIZE_MAX / 2 bytes), a std::length_error (vector::reserve past max_size) and a non-std exception INSIDE an entry point; each must
    come back as NM_ERR_INTERNAL with a message - under ctypes an escaping exception would abort the interpreter, so reaching the
    asserts is the test.  Then the sources: every exported entry point that returns a status or a size is a function-try-block."""
    lib = _lib.load()
    assert _lib.NM_ERR_INTERNAL == -6
    for kind, word in ((0, b"bad_alloc"), (1, b"reserve"), (2, b"unknown C++ exception")):
        rc = lib.nm_abi_selftest_throw(kind)
        assert rc == _lib.NM_ERR_INTERNAL, (kind, rc)
        msg = lib.nm_last_error()
        assert b"nm_abi_selftest_throw" in msg and word in msg, msg
    assert lib.nm_abi_selftest_throw(3) == 0
    # static half: the extern "C" definitions of every declared symbol
    src = ""
    csrc = os.path.join(ROOT, "neural_marionette_amd", "csrc")
    for f in sorted(os.listdir(csrc)):
        if f.endswith(".hip"):
            src += open(os.path.join(csrc, f)).read()
    missing = []
    for sym in _declared_symbols():
        m = re.search(r"^(?:int|size_t|int32_t)\s+" + sym + r"\s*\([^{;]*?\)\s*(try\b)?\s*\{", src, flags=re.M | re.S)
        if m is None:
            continue                                   # void / const char* accessors (nm_last_error, nm_abi_version ...): nothing that throws
        if m.group(1) is None and sym not in ("nm_abi_version",):      # (returns a constant)
            missing.append(sym)
    assert not missing, f"entry points without the exception barrier: {missing}"
    assert len(re.findall(r"catch \(\.\.\.\) \{ return nm_abi_catch\(", src)) >= 40


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


def test_fresh_network_is_initialised_like_torch_layers():
    """A from-scratch network (train.py:233 of the reference, before any checkpoint) must not be all zeros: the holders
    are torch layer subclasses (default init), found by class name / isinstance the way utils/train_utils.py:248-264
    looks for them.  (Bit equality with the reference tree: tests/test_oracle_vs_reference.py.)"""
    from torch import nn
    from neural_marionette_amd.modules import KyptDetector, HSVRNNBVH
    o = HotPathOptions(grid_size=32)
    torch.manual_seed(0)
    net = NeuralMarionette(o)
    sd = net.state_dict()
    assert torch.equal(sd["kypt_detector.affinity_params"], torch.ones(2, 24, 23))
    for k, v in sd.items():
        if v.dim() >= 2 and not k.endswith("affinity_params"):
            assert float(v.abs().max()) > 0 and float(v.std()) > 0, k
    gn = [m for m in net.modules() if isinstance(m, nn.GroupNorm)]
    assert len(gn) == 2 * 36 + 4 and all(bool((m.weight == 1).all()) and bool((m.bias == 0).all()) for m in gn)
    names = [type(m).__name__ for m in net.modules()]
    for cls, n in (("Basic3DBlock", 2), ("Pool3DBlock", 10), ("Res3DBlock", 22), ("Upsample3DBlock", 6), ("HG", 2)):
        assert names.count(cls) == n, (cls, names.count(cls))
    convs = [m for m in net.modules() if isinstance(m, (nn.Conv3d, nn.ConvTranspose3d))]
    assert len(convs) == 2 * 36 + 3 + 1 + 5 and all("Conv" in type(m).__name__ for m in convs)
    assert isinstance(net.dyna_module.kypt_rnn_cell, nn.GRUCell) and isinstance(net.dyna_module.extract_post_dist[0], nn.Linear)
    # holders never compute (no silent PyTorch path)
    with pytest.raises(_lib.NmError):
        net.kypt_detector.vox_to_kypt.extract_features(torch.zeros(1, 4, 32, 32, 32))
    with pytest.raises(_lib.NmError):
        convs[0](torch.zeros(1, 4, 8, 8, 8))
    # stand-alone halves construct with their own engines and the reference's sub-dict keys
    det, dyn = KyptDetector(o), HSVRNNBVH(o)
    assert ["kypt_detector." + k for k in det.state_dict()] + ["dyna_module." + k for k in dyn.state_dict()] == list(sd)
    assert det._eng() is not net._engine and dyn._eng().prefix == "dyna_module."


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


def test_bench_reads_the_committed_traffic_profiles():
    """bench.py's roofline.traffic comes from profiles/<round>_pmc_traffic.json + <round>_fetch_calib.json as tools/collect_evidence.sh
    leaves them - stamped with the tree id (an extra string-valued key in both).  The committed files must parse, give a number inside
    their own bounds, and a file from another round must be refused, not crash."""
    import importlib.util, json
    spec = importlib.util.spec_from_file_location("nm_bench", os.path.join(ROOT, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    path = os.path.join(ROOT, "profiles", f"{b.ROUND}_pmc_traffic.json")
    if not os.path.exists(path):
        t, det = b.pmc_traffic("conv_f16p2_kernel")
        assert t is None and "note" in det
        return
    pm = json.load(open(path))
    assert pm["round"] == b.ROUND and pm["kernels"]
    for rec in pm["kernels"][:6]:
        t, det = b.pmc_traffic(rec["kernel"])
        assert t is not None, det
        assert det["traffic_lower_bound"] <= t <= det["traffic_upper_bound"] + 1, det
        assert all(isinstance(v, float) for v in det["calibration"].values()) and len(det["calibration"]) >= 3
    t, det = b.pmc_traffic("no_such_kernel")
    assert t is None and "no record" in det["note"]
