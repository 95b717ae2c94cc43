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
    for wv in (0, 3):
        v = s[:, 2:40, wv, :]            # skip the first steps
        ok = v[:, :, 12] > 0
        d = lambda a, b_: ((v[:, :, b_] - v[:, :, a])[ok]).mean()
        nxt = (s[:, 3:41, wv, 0] - s[:, 2:40, wv, 12])[ok & (s[:, 3:41, wv, 0] > 0)].mean()
        print(f"  wave {wv}: g0 mfma {d(0,1):6.0f} wait {d(1,2):5.0f} bar {d(2,3):5.0f} | g1 mfma {d(3,4):6.0f} wait {d(4,5):5.0f} bar {d(5,6):5.0f} |"
              f" g2 mfma {d(6,7):6.0f} wait {d(7,8):5.0f} bar {d(8,9):5.0f} | write {d(9,10):5.0f} epi(avg over steps) {d(10,11):5.0f} bar {d(11,12):5.0f} | to next step {nxt:5.0f} | step {d(0,12)+nxt:6.0f}")
