#!/usr/bin/env python3
"""Diagnostic (never part of the product build): per-step phase stamps of conv_f16p (NM_DIAG build, GPU box)."""
import ctypes as C, sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neural_marionette_amd import _lib
lib = _lib.load()
cfg = _lib.NmConfig(device=0, grid_size=64, nkeypoints=24, nlatent=128, nhidden=512, nneighbor=2, gaussian_sigma=1.5,
                    sep_sigma=0.02, vol_fit_chamfer=1, use_graph_traj=1)
ctx = _lib.Context(cfg); ctx.bind_stream()
raw = C.CDLL(_lib.LIB_PATH)
for (Cin, Cout, size, N) in [(32, 32, 64, 16), (64, 64, 32, 16)]:
    x = torch.randn(N, size, size, size, Cin, device="cuda")
    w = torch.randn(Cout, Cin, 3, 3, 3, device="cuda") * 0.05
    b = torch.zeros(Cout, device="cuda")
    sc = torch.ones(N, Cin, device="cuda"); sh = torch.zeros(N, Cin, device="cuda")
    out = torch.empty(N, size, size, size, Cout, device="cuda")
    gam = torch.ones(Cout, device="cuda"); bet = torch.zeros(Cout, device="cuda")
    gsc = torch.zeros(N, Cout, device="cuda"); gsh = torch.zeros(N, Cout, device="cuda")
    stamps = torch.zeros(256 * 64 * 8 * 16 + 64, dtype=torch.int64, device="cuda")
    def run():
        _lib.check(lib.nm_op_conv3d(ctx.handle, x.data_ptr(), N, size, size, size, Cin, sc.data_ptr(), sh.data_ptr(), 0.01,
                                    w.data_ptr(), b.data_ptr(), Cout, 3, 1, 1, out.data_ptr(), Cout // 16, gam.data_ptr(),
                                    bet.data_ptr(), gsc.data_ptr(), gsh.data_ptr(), 0), "conv")
    raw.nm_diag_set_stamps(C.c_void_p(0)); run(); torch.cuda.synchronize()
    raw.nm_diag_set_stamps(C.c_void_p(stamps.data_ptr()))
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record(); run(); t1.record(); torch.cuda.synchronize()
    raw.nm_diag_set_stamps(C.c_void_p(0))
    s = stamps[: 256 * 64 * 8 * 16].view(256, 64, 8, 16).cpu().numpy().astype(np.float64)
    C16 = Cin // 16
    print(f"Cin={Cin} Cout={Cout} size={size}: {t0.elapsed_time(t1)*1e3:.0f} us")
    first = s[:, 0, 0, 0]; lastv = s[:, :, 0, 7].max(axis=1)
    span = (lastv - first)[first > 0]
    print(f"  per-block span (step 0 start .. last stamped step end): mean {span.mean():.0f} ticks, max {span.max():.0f}; kernel {t0.elapsed_time(t1)*1e3:.0f} us -> {span.max()/(t0.elapsed_time(t1)*1e3):.0f} ticks/us; steps stamped {int((s[0,:,0,7]>0).sum())}")
    v = s[:, 2:40, 0, :]; ok = v[:, :, 7] > 0
    d = lambda a, b_: ((v[:, :, b_] - v[:, :, a])[ok]).mean()
    nxt = (s[:, 3:41, 0, 0] - s[:, 2:40, 0, 7])[ok & (s[:, 3:41, 0, 0] > 0)].mean()
    print(f"  mfma wave 0: g0 {d(0,1):6.0f} bar {d(1,2):5.0f} | g1 {d(2,3):6.0f} bar {d(3,4):5.0f} | g2 {d(4,5):6.0f} bar {d(5,6):5.0f} | tail {d(6,7):5.0f} next {nxt:5.0f} | step {d(0,7)+nxt:6.0f}")
    v = s[:, 2:40, 4, :]; ok = v[:, :, 8] > 0
    print(f"  producer 4 : issue {d(0,1):6.0f} cvt0-3 {d(1,2):6.0f} bar {d(2,3):5.0f} | cvt4-9 {d(3,4):6.0f} wait {d(4,5):5.0f} bar {d(5,6):5.0f} | g2 {d(6,7):6.0f} bar {d(7,8):5.0f}")
