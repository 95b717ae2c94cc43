"""Upper bound for a Winograd form of conv_f16p2 (DESIGN 4 "Why there is no Winograd kernel"), measured instead of priced: the
diagnostic build (libnm355_diag.so, make diag) carries two instantiations of conv_f16p2_kernel with the RESOURCE PROFILE of a 1-D
F(2,3) kernel along z and none of its arithmetic (results are wrong on purpose):
  NM355_P2_WEMU=1   a step is 4 position groups x 9 taps x 6 MFMAs (216 for 324 = 1.5 x fewer), one A tile + four B reads per tap,
                    four weight groups (36 transformed tap matrices for 27) and four workgroup barriers per step;
  NM355_P2_WEMU=3   as 1, and the producers convert 13 pieces per step for 10 (8 transformed planes for 6 raw ones).
Left out in the emulation's favour: the extra global loads of the 1.33 x larger tile, the transform's additions, the output transform.
Layers: the two conv_f16p2 shapes of the forward (64 -> 64 @32^3, 128 -> 128 @16^3, 64 frames), split-fp16 mode, through nm_op_conv3d.
usage: diag_winograd_emu.py   (spawns itself per variant, NM355_P2_WEMU is read at the first launch)"""
import os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, os.path.join(HERE, ".."))
    from neural_marionette_amd import _lib
    _lib.LIB_PATH = os.path.join(HERE, "..", "neural_marionette_amd", "libnm355_diag.so")
    import torch
    cfg = dict(device=0, grid_size=64, nkeypoints=24, nlatent=128, nhidden=512, nneighbor=2, gaussian_sigma=1.5, sep_sigma=0.02, vol_fit_chamfer=1, use_graph_traj=1)
    c = _lib.Context(_lib.NmConfig(**cfg)); c.bind_stream(); lib = c.lib
    for (Cin, Cout, size, n) in ((64, 64, 32, 64), (128, 128, 16, 64)):
        x = torch.randn(n, size, size, size, Cin, device="cuda"); w = torch.randn(Cout, Cin, 3, 3, 3, device="cuda") * 0.05; b = torch.randn(Cout, device="cuda")
        sc = 1 + 0.1 * torch.randn(n, Cin, device="cuda"); sh = 0.1 * torch.randn(n, Cin, device="cuda")
        gam = torch.ones(Cout, device="cuda"); bet = torch.zeros(Cout, device="cuda")
        out = torch.zeros(n, size, size, size, Cout, device="cuda"); gsc = torch.zeros(n, Cout, device="cuda"); gsh = torch.zeros(n, Cout, device="cuda")
        def run():
            _lib.check(lib.nm_op_conv3d(c.handle, x.data_ptr(), n, size, size, size, Cin, sc.data_ptr(), sh.data_ptr(), 0.01, w.data_ptr(), b.data_ptr(), Cout, 3, 1, 1,
                                        out.data_ptr(), Cout // 16, gam.data_ptr(), bet.data_ptr(), gsc.data_ptr(), gsh.data_ptr(), 0), "conv")
        for _ in range(3): run()
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(20): run()
            e.record(); torch.cuda.synchronize()
            best = min(best, a.elapsed_time(e) / 20)
        print("  %3d->%3d @%d^3 x%d: %.3f ms per layer call (conv + GroupNorm statistics), best of 3 x 20" % (Cin, Cout, size, n, best))
    sys.exit(0)
names = {0: "conv_f16p2 as shipped (324 MFMAs, 3 weight groups, 10 pieces per step)", 1: "emulation: 216 MFMAs, 4 weight groups + barriers per step",
         3: "emulation: as 1 + 13 converted pieces per step"}
for d in (0, 1, 3, 0, 1, 3):
    env = dict(os.environ); env["NM355_P2_WEMU"] = str(d)
    print("NM355_P2_WEMU=%d (%s)" % (d, names[d]), flush=True)
    subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env)
