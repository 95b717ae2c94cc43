#!/usr/bin/env python3
"""Diagnostic (never part of the product build): per-step phase stamps of conv_f16p2 (libnm355_stamps.so: make -C neural_marionette_amd/csrc stamps).  MFMA wave 0: [0] step start, [1]/[2] before/after the barrier that ends tap group 0,
[3]/[4] group 1, [5]/[6] group 2.  Producer wave 4: [0] step start, [1] weight loads of group 1 landed, [2]/[3] barrier 0, [4] pieces 4-9
converted, [5]/[6] barrier 1, [7]/[8] barrier 2."""
import ctypes as C, sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neural_marionette_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), "libnm355_stamps.so")
lib = _lib.load()
cfg = _lib.NmConfig(device=0, grid_size=64, nkeypoints=24, nlatent=128, nhidden=512, nneighbor=2, gaussian_sigma=1.5, sep_sigma=0.02, vol_fit_chamfer=1, use_graph_traj=1)
ctx = _lib.Context(cfg); ctx.bind_stream()
raw = C.CDLL(_lib.LIB_PATH)
for (Cin, Cout, size, N) in [(64, 64, 32, 64), (32, 64, 32, 64), (32, 32, 64, 16)]:
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
    print(f"Cin={Cin} Cout={Cout} size={size} N={N}: {t0.elapsed_time(t1)*1e3:.0f} us (whole op incl. weight packing)")
    m = s[:, 4:40, 0, :]; okm = (m[:, :, 6] > 0) & (m[:, :, 0] > 0)
    dm = lambda a, b_: ((m[:, :, b_] - m[:, :, a])[okm]).mean()
    nxt = (s[:, 5:41, 0, 0] - s[:, 4:40, 0, 6])[okm & (s[:, 5:41, 0, 0] > 0)].mean()
    print(f"  MFMA wave 0 (clock64 ticks): group0 {dm(0,1):7.0f} barrier {dm(1,2):6.0f} | group1 {dm(2,3):7.0f} barrier {dm(3,4):6.0f} | group2 {dm(4,5):7.0f} barrier {dm(5,6):6.0f} | to next step {nxt:6.0f} | step {dm(0,6)+nxt:7.0f}")
    C16 = Cin // 16
    if C16 > 1:
        first = (np.arange(4, 40) % C16 == 0)[None, :] & okm
        rest = (np.arange(4, 40) % C16 != 0)[None, :] & okm
        gm = lambda sel, a, b_: ((m[:, :, b_] - m[:, :, a])[sel]).mean()
        print(f"    first step of a brick: groups {gm(first,0,1):6.0f} {gm(first,2,3):6.0f} {gm(first,4,5):6.0f} | other steps: {gm(rest,0,1):6.0f} {gm(rest,2,3):6.0f} {gm(rest,4,5):6.0f}")
    pr = s[:, 4:40, 4, :]; okp = (pr[:, :, 8] > 0) & (pr[:, :, 0] > 0)
    dp = lambda a, b_: ((pr[:, :, b_] - pr[:, :, a])[okp]).mean()
    print(f"  producer 4: wait weights {dp(0,1):6.0f} cvt 0-3 + store {dp(1,2):6.0f} barrier {dp(2,3):6.0f} | cvt 4-9 {dp(3,4):6.0f} wait+store {dp(4,5):6.0f} barrier {dp(5,6):6.0f} | group 2 {dp(6,7):6.0f} barrier {dp(7,8):6.0f}")
    e = s[:, 4:40, 0, :]; oke = (e[:, :, 8] > 0) & (e[:, :, 13] > 0) & (e[:, :, 7] > 0)
    if oke.any() and not (e[:, :, 9][oke] > 0).any():      # deferred epilogue (NM355_P2_DEFER=1): only the take is exposed
        print(f"  deferred epilogue: exposed take (combine + bias into the pending set) {((e[:, :, 13] - e[:, :, 8])[oke]).mean():6.0f}")
    elif oke.any():
        de = lambda a, b_: ((e[:, :, b_] - e[:, :, a])[oke]).mean()
        print(f"  epilogue of a finished brick: stores nt0 {de(8,9):6.0f} sums nt0 {de(9,10):6.0f} | stores nt1 {de(10,11):6.0f} sums nt1 {de(11,12):6.0f} | barrier {de(12,13):6.0f} | reduce + part {de(13,7):6.0f} | total {de(8,7):6.0f}")
