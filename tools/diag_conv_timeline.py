#!/usr/bin/env python3
"""Diagnostic (never part of the product build): per-wave phase timeline of conv_f16s workgroups sharing one CU.
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
CASES = [(32, 32, 64, 16, 0), (64, 32, 32, 16, 1), (64, 64, 32, 16, 0)]
which = int(sys.argv[1]) if len(sys.argv) > 1 else 0
Cin, Cout, size, N, up2 = CASES[which]
od = size * (2 if up2 else 1)
x = torch.randn(N, size, size, size, Cin, device="cuda")
w = torch.randn(Cout, Cin, 3, 3, 3, device="cuda") * 0.05
b = torch.zeros(Cout, device="cuda")
sc = torch.ones(N, Cin, device="cuda"); sh = torch.zeros(N, Cin, device="cuda")
out = torch.empty(N, od, od, od, Cout, device="cuda")
gam = torch.ones(Cout, device="cuda"); bet = torch.zeros(Cout, device="cuda")
gsc = torch.zeros(N, Cout, device="cuda"); gsh = torch.zeros(N, Cout, device="cuda")
nitem = N * (od // 8) * (od // 8) * (od // 4)
ngy = max(1, Cout // 64)
stamps = torch.zeros(nitem * 4 * 16 + 64, dtype=torch.int64, device="cuda")
def run():
    _lib.check(lib.nm_op_conv3d(ctx.handle, x.data_ptr(), N, size, size, size, Cin, sc.data_ptr(), sh.data_ptr(), 0.01,
                                w.data_ptr(), b.data_ptr(), Cout, 3, 1, 1, out.data_ptr(), Cout // 16, gam.data_ptr(),
                                bet.data_ptr(), gsc.data_ptr(), gsh.data_ptr(), up2), "conv")
raw.nm_diag_set_stamps(C.c_void_p(0)); run(); torch.cuda.synchronize()
raw.nm_diag_set_stamps(C.c_void_p(stamps.data_ptr())); run(); torch.cuda.synchronize()
raw.nm_diag_set_stamps(C.c_void_p(0))
s = stamps[: nitem * 64].view(nitem, 4, 16).cpu().numpy()
hw = s[:, :, 15]; xcc = s[:, :, 14] & 0xF; blk = s[:, :, 13]
wave_slot = hw & 0xF; simd = (hw >> 4) & 3; cu = (hw >> 8) & 0xF; sh_ = (hw >> 12) & 1; se = (hw >> 13) & 7
key = (xcc * 8 + se) * 32 + sh_ * 16 + cu
print("distinct CU keys:", len(np.unique(key[:, 0])), "blocks:", len(np.unique(blk[:, 0])))
# blocks per CU
import collections
cu_blocks = collections.defaultdict(set)
for i in range(nitem):
    cu_blocks[int(key[i, 0])].add(int(blk[i, 0]))
cnt = collections.Counter(len(v) for v in cu_blocks.values())
print("blocks per CU histogram:", dict(cnt))
k0 = sorted(cu_blocks.keys())[3]
bl = sorted(cu_blocks[k0])
print("CU", k0, "blocks", bl)
t0 = min(int(s[i, :, 0].min()) for i in range(nitem) if int(blk[i, 0]) in bl)
names = ["start", "top", "staged0", "bar0", "mfma0end", "top1", "staged1", "bar1", "mfma1end", "mfmaend", "epi"]
for b_ in bl:
    items = [i for i in range(nitem) if int(blk[i, 0]) == b_][:3]
    for i in items:
        for wv in range(4):
            r = s[i, wv]
            print(f"blk {b_:4d} item {i:6d} wave {wv} simd {int(simd[i, wv])} slot {int(wave_slot[i, wv])}: " +
                  " ".join(f"{names[j]}={int(r[j]) - t0:7d}" for j in (0, 1, 2, 3, 4, 6, 7, 8, 9, 10)))
# aggregate: per SIMD of this CU, fraction of time with 0 / 1 / 2 waves inside an MFMA phase
ev = []
for i in range(nitem):
    if int(key[i, 0]) != k0: continue
    for wv in range(4):
        r = s[i, wv]
        if r[10] == 0: continue
        for a, b2 in ((3, 4), (7, 8)):
            ev.append((int(simd[i, wv]), int(r[a]), int(r[b2])))
for sd in range(4):
    iv = [(a, b2) for (q, a, b2) in ev if q == sd]
    if not iv: continue
    lo = min(a for a, _ in iv); hi = max(b2 for _, b2 in iv)
    pts = sorted([(a, 1) for a, _ in iv] + [(b2, -1) for _, b2 in iv])
    occ = {0: 0, 1: 0, 2: 0, 3: 0}; cur = 0; last = lo
    for t, d in pts:
        occ[min(cur, 3)] += t - last; last = t; cur += d
    tot = hi - lo
    print(f"SIMD {sd}: span {tot} cycles; waves in MFMA phase: 0 -> {occ[0]/tot:.2f}, 1 -> {occ[1]/tot:.2f}, 2 -> {occ[2]/tot:.2f}")
