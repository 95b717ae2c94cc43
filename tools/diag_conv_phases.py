#!/usr/bin/env python3
"""Diagnostic (never part of the product build): per-phase cycle shares of conv_f16s blocks.
Build with `make -C neural_marionette_amd/csrc clean all DIAGFLAGS=-DNM_DIAG`, run on the GPU box."""
import ctypes as C, sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neural_marionette_amd import _lib

lib = _lib.load()
cfg = _lib.NmConfig(device=0, grid_size=64, nkeypoints=24, nlatent=128, nhidden=512, nneighbor=2, gaussian_sigma=1.5,
                    sep_sigma=0.02, vol_fit_chamfer=1, use_graph_traj=1)
ctx = _lib.Context(cfg); ctx.bind_stream()
raw = C.CDLL(_lib.LIB_PATH)
CASES = [(32, 32, 64, 16, 0), (64, 32, 32, 16, 1), (64, 64, 32, 16, 0), (128, 64, 16, 16, 1)]
if len(sys.argv) > 1:
    CASES = CASES[: int(sys.argv[1])]
for (Cin, Cout, size, N, up2) in CASES:
    od = size * (2 if up2 else 1)
    x = torch.randn(N, size, size, size, Cin, device="cuda")
    w = torch.randn(Cout, Cin, 3, 3, 3, device="cuda") * 0.05
    b = torch.zeros(Cout, device="cuda")
    sc = torch.ones(N, Cin, device="cuda"); sh = torch.zeros(N, Cin, device="cuda")
    out = torch.empty(N, od, od, od, Cout, device="cuda")
    gam = torch.ones(Cout, device="cuda"); bet = torch.zeros(Cout, device="cuda")
    gsc = torch.zeros(N, Cout, device="cuda"); gsh = torch.zeros(N, Cout, device="cuda")
    nblk = N * (od // 8) * (od // 8) * (od // 4) * max(1, Cout // 64)
    stamps = torch.zeros(nblk * 64 + 64, dtype=torch.int64, device="cuda")    # [item][wave][16]
    def run():
        _lib.check(lib.nm_op_conv3d(ctx.handle, x.data_ptr(), N, size, size, size, Cin, sc.data_ptr(), sh.data_ptr(), 0.01,
                                    w.data_ptr(), b.data_ptr(), Cout, 3, 1, 1, out.data_ptr(), Cout // 16, gam.data_ptr(),
                                    bet.data_ptr(), gsc.data_ptr(), gsh.data_ptr(), up2), "conv")
    raw.nm_diag_set_stamps(C.c_void_p(0)); run(); torch.cuda.synchronize()
    raw.nm_diag_set_stamps(C.c_void_p(stamps.data_ptr()))
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record(); run(); t1.record(); torch.cuda.synchronize()
    raw.nm_diag_set_stamps(C.c_void_p(0))
    s4 = stamps[: nblk * 64].view(nblk, 4, 16).cpu().numpy().astype(np.float64)
    s = s4[:, 0, :]
    ok = s[:, 10] > 0
    s = s[ok]
    d = lambda a, b_: (s[:, b_] - s[:, a]).mean()
    wl = s4[s4[:, 0, 10] > 0]
    blk_span = (wl[:, :, 10].max(axis=1) - wl[:, :, 0].min(axis=1)).mean()
    mf = ((wl[:, :, 4] - wl[:, :, 3]) + (wl[:, :, 8] - wl[:, :, 7])).mean() / 2
    print(f"   all waves: block span {blk_span:8.0f}, mean mfma phase/chunk {mf:8.0f}")
    for wv in range(4):
        d = lambda a, b_: (wl[:, wv, b_] - wl[:, wv, a]).mean()
        print(f"   wave {wv}: wait-top {d(0,1):7.0f} stage0 {d(1,2):7.0f} bar {d(2,3):6.0f} mfma0 {d(3,4):7.0f} | stage1 {d(5,6):7.0f} bar {d(6,7):6.0f} mfma1 {d(7,8):7.0f} | tail {d(8,9):7.0f} epi {d(9,10):7.0f}")
    mx = np.maximum.reduce([wl[:, wv, 4] - wl[:, wv, 3] for wv in range(4)]).mean()
    print(f"   slowest wave's mfma0 per item: {mx:8.0f}")
    print(f"Cin={Cin} Cout={Cout} size={size} up2={up2} N={N}: {t0.elapsed_time(t1)*1e3:.0f} us total, blocks={len(s)}")
    print(f"   prologue {d(0,1):8.0f} | stage0 issue+convert {d(1,2):8.0f} barrier {d(2,3):7.0f} mfma0 {d(3,4):8.0f} |"
          f" stage1 {d(5,6):8.0f} barrier {d(6,7):7.0f} mfma1 {d(7,8):8.0f} | epilogue {d(9,10):8.0f} | block total {d(0,10):8.0f} cycles (100 MHz ticks x?)")
