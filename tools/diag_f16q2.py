"""Ablations of conv_f16q2_kernel (diagnostic build libnm355_diag.so, -DNM_Q2_DIAG; NM355_Q2_DBG read at the first launch, so one
process per variant): bf16 64 -> 64 @32^3 x 64 frames through nm_op_conv3d.  usage: diag_f16q2.py   (spawns itself per variant)"""
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
        _lib.check(lib.nm_set_conv_mode(c.handle, 4), "mode"); _lib.check(lib.nm_op_set_storage16(c.handle, 1, 1), "set16")
        x = torch.randn(n, size, size, size, Cin, device="cuda").to(torch.bfloat16); w = torch.randn(Cout, Cin, 3, 3, 3, device="cuda") * 0.05; b = torch.randn(Cout, device="cuda")
        sc = 1 + 0.1 * torch.randn(n, Cin, device="cuda"); sh = 0.1 * torch.randn(n, Cin, device="cuda")
        gam = torch.ones(Cout, device="cuda"); bet = torch.zeros(Cout, device="cuda")
        out = torch.zeros(n, size, size, size, Cout, device="cuda", dtype=torch.bfloat16); gsc = torch.zeros(n, Cout, device="cuda"); gsh = torch.zeros(n, Cout, device="cuda")
        def run():
            _lib.check(lib.nm_op_conv3d(c.handle, x.data_ptr(), n, size, size, size, Cin, sc.data_ptr(), sh.data_ptr(), 0.01, w.data_ptr(), b.data_ptr(), Cout, 3, 1, 1,
                                        out.data_ptr(), Cout // 16, gam.data_ptr(), bet.data_ptr(), gsc.data_ptr(), gsh.data_ptr(), 0), "conv")
        for _ in range(3): run()
        torch.cuda.synchronize()
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20): run()
        e.record(); torch.cuda.synchronize()
        print("  %3d->%3d @%d^3 x%d: %.3f ms" % (Cin, Cout, size, n, a.elapsed_time(e) / 20))
    sys.exit(0)
names = {8: "input pieces loaded, not converted", 9: "input pieces converted, not loaded", 6: "producers idle + no operand reads in the tap loop", 7: "producers idle + no epilogue", 0: "full kernel", 1: "no MFMAs", 2: "producers idle (barriers only)", 3: "no epilogue stores", 4: "no weight copies", 5: "no input staging"}
for d in [int(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else "0,1,2,3,5,6,7,0".split(","))]:
    env = dict(os.environ); env["NM355_Q2_DBG"] = str(d)
    print("NM355_Q2_DBG=%d (%s)" % (d, names[d]), flush=True)
    subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env)
