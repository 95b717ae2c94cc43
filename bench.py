#!/usr/bin/env python3
"""bench.py — voxel-frames/s of the Neural Marionette hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): AIST++-shaped synthetic clips, 64^3 occupancy grid,
T = 16, B = 4 clips per GPU, full NeuralMarionette.forward (keypoint detector incl. all 11
losses + HSVRNNBVH.encode), fp32, seeded random weights of the reference architecture.  A
"step" is one forward over one batch of B*T = 64 voxel-frames whose inputs are already
resident in HBM.  N > 1: independent clips per rank (weak scaling, no data-path collective);
launched by `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...`.

Rank 0 prints ONE JSON line with the throughput, the roofline of the dominant kernel
(HIP-event timed inside the timed region, algorithmic FLOPs) and the CPU baseline (the
oracle — a port of the reference on PyTorch-CPU ops — timed on the host cores, N = 1 only).
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None,
                    help="ranks (one per GPU); default: WORLD_SIZE under a torch.distributed launcher, else 1")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the secondary measurements (exact-fp32 forward, training step)")
    ap.add_argument("--workload", choices=["forward", "train"], default="forward",
                    help="forward (default, BASELINE configs[1]): full NeuralMarionette.forward; train (configs[2] shape, fp32): one "
                         "detector-mode training step = forward + backward + gradient all-reduce + Adam")
    ap.add_argument("--conv-mode", choices=["split16", "fp32", "f16", "bf16"], default="split16",
                    help="split16: fp32-equivalent 3x f16 MFMA products (default); fp32: exact fp32 MFMA everywhere; f16: products of fp16-rounded "
                         "operands, fp32 storage; bf16: f16 arithmetic + bfloat16 storage of the training path (BASELINE config 3 as named)")
    ap.add_argument("--ranks-on-one-gpu", type=int, default=0, metavar="R",
                    help="DIAGNOSTIC (verdict r5 item 7b): R processes, one context each, ALL on device 0, gloo for the timing collectives - what R "
                         "ranks sharing the host's launch path (and one GPU) cost: the aggregate voxel-frames/s against the one-process figure. "
                         "Forces --no-extras --no-cpu-baseline; not a scaling number")
    ap.add_argument("--dist-selftest", action="store_true",
                    help="run only the multi-rank plumbing of this file (rendezvous, barrier, all-reduce of ones, MAX-reduce of the elapsed "
                         "time) and print it as one JSON line; backend gloo when no GPU is visible (tests/test_sharding_cpu.py)")
    return ap.parse_args(argv)


def launch_ranks(args, argv) -> int:
    """`python bench.py --gpus N` outside a torch.distributed launcher: start the N ranks as a CHILD
    `python -m torch.distributed.run` (one process per GPU, rendezvous on 127.0.0.1) and relay rank 0's JSON line.  This runs
    before torch is imported, so the parent never initialises the GPU (a process that has must not be replaced or forked).
    Returns the exit code: the child's, or 3 if the line does not show N ranks in the all-reduce."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    # --standalone: torch.distributed.run picks the rendezvous port itself (a bind-then-close probe here could lose it to another
    # process before the child binds); --local-addr: the container hostname may not resolve
    nproc = args.ranks_on_one_gpu or args.gpus
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1",
           f"--nproc-per-node={nproc}", os.path.abspath(__file__), *argv]
    child = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True)
    line = None
    for ln in child.stdout:
        if ln.startswith("{"):
            line = ln.strip()
        else:
            sys.stderr.write(ln)
    rc = child.wait()
    if rc != 0:
        return rc
    if line is None:
        sys.stderr.write("bench.py: the ranks printed no JSON line\n")
        return 3
    print(line, flush=True)
    d = json.loads(line).get("distributed", {})
    if args.ranks_on_one_gpu:
        return 0 if d.get("ranks_seen_by_allreduce") == args.ranks_on_one_gpu else 3
    if d.get("world_size") != args.gpus or d.get("ranks_seen_by_allreduce") != args.gpus:
        sys.stderr.write(f"bench.py: asked for {args.gpus} ranks, the all-reduce saw {d.get('ranks_seen_by_allreduce')} "
                         f"(world_size {d.get('world_size')})\n")
        return 3
    return 0


if __name__ == "__main__" and "WORLD_SIZE" not in os.environ:
    _a = parse_args()
    if (_a.gpus or 1) > 1 or _a.ranks_on_one_gpu > 1:
        sys.exit(launch_ranks(_a, sys.argv[1:]))

import torch  # noqa: E402  (after the launcher: the parent of an N-rank run never loads it)

G, T, B_PER_GPU, S = 64, 16, 4, 10
FP32_MFMA_PEAK_TFLOPS = 157.3        # MI355X_MICROARCH.md: dense fp32 matrix peak (v_mfma_f32_32x32x2_f32)
F16_MFMA_PEAK_TFLOPS = 2500.0        # MI355X_MICROARCH.md: dense f16/bf16 matrix peak (v_mfma_f32_32x32x16_f16)


def cpu_baseline(sd, opts, net, dev):
    """Oracle (kind 'port') on the bench's own shape (BASELINE config 2: B = 4 clips of 64^3 x T=16, full forward =
    detector + losses + VRNN encode).  Thread count: calibrated on a DETECTOR-dominated sample (one 64^3 clip, T = 4: the conv
    stacks are >95 % of the CPU time, as on the full shape; a 2-frame clip over-weights the VRNN's thousands of tiny ops and
    picks too few threads) over {8, 16, 32, 64, 128, os.cpu_count()} in ascending order (counts behind one that is already 3x slower than
    the best are not timed); the two best counts are then both timed on the full shape
    and the faster one gets the remaining runs (median of three).  `cores` = the count used, `host_threads` = os.cpu_count()."""
    from neural_marionette_amd import synth
    from oracle import nm_oracle as O
    ncpu = os.cpu_count() or 1
    default = torch.get_num_threads()
    nb = B_PER_GPU
    vox = synth.figure_clip(nb, T, G, seed=1001)
    eps = synth.make_eps((T, S, nb, opts.nlatent_kypt), seed=1002)
    small_v, small_e = vox[:1, :4].contiguous(), eps[:4, :, :1].contiguous()
    calib, skipped = {}, []
    with torch.no_grad():
        for thr in sorted({min(ncpu, c) for c in (8, 16, 32, 64, 128, ncpu)} | {default}):
            # ascending; once a count is more than 3x slower than the best so far the larger ones are not timed (oversubscribed pools
            # only get worse: 256 threads took 129 s for this 1 s sample on the round-4 evidence host, profiles/r04_bench.json.log)
            if calib and calib[max(calib)] > 3.0 * min(calib.values()):      # (the largest count timed so far)
                skipped.append(thr)
                continue
            torch.set_num_threads(thr)
            O.nm_forward(sd, opts, small_v[:, :1], small_e[:1])           # warm this pool size
            t0 = time.perf_counter()
            O.nm_forward(sd, opts, small_v, small_e)
            calib[thr] = time.perf_counter() - t0
        cand = sorted(calib, key=calib.get)[:2]
        runs = {}
        ref = None
        for thr in cand:                                                   # one full-shape run at each of the two best counts
            torch.set_num_threads(thr)
            t0 = time.perf_counter()
            ref = O.nm_forward(sd, opts, vox, eps)
            runs[thr] = [time.perf_counter() - t0]
        best_thr = min(runs, key=lambda k: runs[k][0])
        torch.set_num_threads(best_thr)
        times = runs[best_thr]
        t_all = time.perf_counter()
        # two more runs at the winner (median of three) unless a run exceeds 40 s on this host (then what fits 90 s more)
        while len(times) < 3 and (time.perf_counter() - t_all) + times[-1] < 90.0:
            t0 = time.perf_counter()
            ref = O.nm_forward(sd, opts, vox, eps)
            times.append(time.perf_counter() - t0)
    times = sorted(times)
    med = times[len(times) // 2] if len(times) % 2 else 0.5 * (times[len(times) // 2 - 1] + times[len(times) // 2])
    # parity of the GPU path on exactly this sample (keypoint L2 vs the CPU reference port)
    with torch.no_grad():
        out = net(vox.to(dev), {"detector": True, "learner": True}, eps=eps.to(dev))
    torch.cuda.synchronize(dev)
    d = (out["keypoints"][..., :3].cpu() - ref["keypoints"][..., :3]).double()
    l2 = d.pow(2).sum(-1).sqrt().max().item()
    lat = max((out[k].cpu().double() - ref[k].double()).abs().max().item() for k in ("z_kypts", "h_kypts"))
    parity = dict(kypt_l2=l2, latent_linf=lat, kypt_recon_linf=(out["kypt_recon"].cpu().double() - ref["kypt_recon"].double()).abs().max().item(),
                  best_idx_equal=bool((out["best_idx"].cpu().long() == ref["best_idx"].long()).all()))
    return dict(value=nb * T / med, unit="voxel-frames/s", cores=best_thr, threads_used=best_thr, host_threads=ncpu, **host_cpu_limits(),
                kind="port", runs_s=[round(t, 3) for t in times],
                thread_calibration_s={str(k): round(v, 3) for k, v in sorted(calib.items())},
                thread_calibration_not_timed=[int(k) for k in skipped],
                full_shape_first_run_s={str(k): round(v[0], 3) for k, v in runs.items()},
                sample=f"{'median' if len(times) >= 3 else 'mean'} of {len(times)} x oracle.nm_forward on {nb} clips of 64^3 x T=16 (the bench shape) "
                       f"(detector + losses + VRNN encode), torch {torch.__version__} CPU ops, "
                       f"{best_thr} threads (calibrated on one 64^3 x T=4 clip, the two best counts both timed on the full shape; "
                       f"os.cpu_count() = {ncpu})"), parity


def host_cpu_limits():
    """What the process may actually use of the host: the scheduler affinity mask and the cgroup CPU quota (a 256-thread host whose
    container is limited to a few cores explains a CPU baseline near an 8-core machine's)."""
    try:
        aff = len(os.sched_getaffinity(0))
    except Exception:
        aff = None
    quota = None
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                quota = f.read().strip()
            if path.endswith("cfs_quota_us"):
                with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                    quota = quota + " " + f.read().strip()
            break
        except Exception:
            continue
    cores = None
    if quota:
        q = quota.split()
        if q[0] not in ("max", "-1") and len(q) > 1 and float(q[1]) > 0:
            cores = float(q[0]) / float(q[1])
    model = None
    try:
        with open("/proc/cpuinfo") as f:
            model = next((ln.split(":", 1)[1].strip() for ln in f if ln.startswith("model name")), None)
    except Exception:
        pass
    return dict(affinity_cpus=aff, cgroup_cpu_max=quota, cgroup_cpu_cores=cores, cpu_model=model)


STEP_TFLOP = 99.15e-3 * B_PER_GPU * T          # algorithmic TFLOP of one forward step on one GPU
# PMC summaries are only read from this round's files under profiles/ (tools/collect_evidence.sh <round> writes them; the round can be
# overridden with NM355_ROUND for a re-run of an older tree)
ROUND = os.environ.get("NM355_ROUND", "r06")


def prof_families(lib, h, _lib):
    """{kernel family: (event-timed ms total, algorithmic flops total, launches)} of the context's current profiler window."""
    fam = {}
    for v in range(16):
        ms, fl, n = C.c_double(), C.c_double(), C.c_int64()
        _lib.check(lib.nm_prof_read(h, v, C.byref(ms), C.byref(fl), C.byref(n)), "prof_read")
        if n.value:
            fam[lib.nm_prof_kernel_name(v).decode()] = (ms.value, fl.value, n.value)
    return fam


# FETCH_SIZE tallies 64 B per fabric read request (profiles/<round>_fetch_calib.txt, tools/calib/): contiguous streams leave the L2 as
# 128-B requests (counter = bytes / 2 - MI355X_MICROARCH.md's x2), 64-B runs at a larger pitch as 64-B requests (counter = bytes).
# Dominant read stream of each conv family -> the factor applied to its raw counter:
FETCH_FACTOR = (
    ("conv_up2c", 2.0, "staging loads 32 B per lane, 128 contiguous bytes per voxel (pattern pair32)"),
    ("conv_pool_f16", 2.0, "32 contiguous bytes per lane (pattern pair32)"),
    ("conv_f16", 1.0, "staging loads 16-B pieces, four lanes per (voxel, 16-channel chunk): 64-B runs at the voxel pitch (pattern seg64)"),
    ("wgrad16", 1.0, "staging loads 16-B pieces in 64-B runs at the voxel pitch (pattern seg64)"),
)


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from THIS round's committed PMC passes (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in
    separate runs, reduced by tools/pmc_traffic_all.py; a PMC pass cannot run inside this process).  WRITE_SIZE is exact; FETCH_SIZE
    counts 64 B per request whatever the request's size, so the read bytes lie between the raw counter and twice it: `traffic` applies
    the factor calibrated for the kernel's dominant read stream (FETCH_FACTOR; calibration file of the same round), and the detail
    carries both bounds.  Without this round's calibration file the figure is withheld (null) - an uncalibrated number is worse than
    none.  A file written in another round is refused."""
    path = os.path.join(ROOT, "profiles", f"{ROUND}_pmc_traffic.json")
    try:
        pm = json.load(open(path))
    except Exception:
        return None, dict(file=None, note=f"no profiles/{ROUND}_pmc_traffic.json yet (tools/pmc_traffic_all.py writes it)")
    if pm.get("round") != ROUND:
        return None, dict(file=os.path.basename(path), note=f"refused: written in round {pm.get('round')!r}, not {ROUND}")
    base = kernel.split("<")[0]
    rec = next((r for r in pm.get("kernels", []) if r["kernel"] == kernel), None) or \
        next((r for r in pm.get("kernels", []) if r["kernel"].split("<")[0] == base), None)
    if rec is None:
        return None, dict(file=os.path.basename(path), note=f"no record for {kernel}")
    det = dict(file=os.path.basename(path), kernel=rec["kernel"], launches=rec["launches"], fetch_bytes_raw=rec["fetch_bytes_raw"],
               write_bytes=rec["write_bytes"], traffic_lower_bound=rec["fetch_bytes_raw"] + rec["write_bytes"],
               traffic_upper_bound=2 * rec["fetch_bytes_raw"] + rec["write_bytes"])
    try:
        cal = json.load(open(os.path.join(ROOT, "profiles", f"{ROUND}_fetch_calib.json")))
    except Exception:
        det["note"] = f"no profiles/{ROUND}_fetch_calib.json: FETCH_SIZE uncalibrated this round, traffic withheld (bounds above)"
        return None, det
    fac, why = next(((f, w) for pre, f, w in FETCH_FACTOR if base.startswith(pre)), (2.0, "contiguous 16 B per lane (pattern contig16)"))
    factors = {k: v["factor"] for k, v in cal.items() if isinstance(v, dict) and "factor" in v}     # (the file also carries its tree id)
    if not factors:
        det["note"] = f"profiles/{ROUND}_fetch_calib.json holds no calibration records: traffic withheld (bounds above)"
        return None, det
    det.update(fetch_factor=fac, fetch_pattern=why, calibration=factors, calibration_file=f"{ROUND}_fetch_calib.json")
    return fac * rec["fetch_bytes_raw"] + rec["write_bytes"], det


def extra_measurements(net, vox, eps, acts, dev, barrier, dist_on, world):
    """Secondary numbers in the same JSON line, measured AFTER the timed region (the headline is untouched):
    `fp32_exact` - the same forward with every conv on the exact fp32 MFMA path (the reference's literal arithmetic);
    `train` - BASELINE configs[2]'s per-GPU shape (64^3, T = 16, B = 4 clips per GPU): detector-mode training step = training
    forward + backward of the 11 weighted losses + bucketed gradient all-reduce (RCCL when N > 1) + fused Adam, fp32-equivalent."""
    from neural_marionette_amd.train import DetectorTrainer
    from neural_marionette_amd import _lib
    out = {}

    def timed(fn, warm, steps):
        for _ in range(warm):
            fn()
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        barrier()
        dt = time.perf_counter() - t0
        if dist_on:
            import torch.distributed as dist
            tt = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        return dt / steps

    def fwd():
        with torch.no_grad():
            return net(vox, acts, eps=eps)
    # the same forward on the second synthetic generator SURVEY 8(d) names: Bernoulli(p = 0.03) occupancy - no empty 4x8x8 brick, so
    # the inference path's sparse first layer (data-dependent) does all of its work
    from neural_marionette_amd import synth as _synth
    vb = _synth.bernoulli_clip(B_PER_GPU, T, G, p=0.03, seed=1).to(dev)

    def fwd_b():
        with torch.no_grad():
            return net(vb, acts, eps=eps)
    ms = timed(fwd_b, 2, 10) * 1e3
    out["forward_bernoulli"] = dict(value=world * B_PER_GPU * T / (ms * 1e-3), unit="voxel-frames/s", ms_per_step=ms, steps=10,
                                    workload="the headline forward on Bernoulli(p=0.03) occupancy clips instead of figure clips (no empty brick: "
                                             "the sparse first layer and its pool conv run dense)")
    del vb
    net.set_conv_mode("fp32")
    ms = timed(fwd, 1, 4) * 1e3
    out["fp32_exact"] = dict(value=world * B_PER_GPU * T / (ms * 1e-3), unit="voxel-frames/s", ms_per_step=ms, steps=4,
                             dtype="f32 (v_mfma_f32_32x32x2_f32 for every conv: bit-exact fp32 fma chains)",
                             frac_of_fp32_mfma_peak=STEP_TFLOP / (ms * 1e-3) / FP32_MFMA_PEAK_TFLOPS)
    net.set_conv_mode("split16")
    saved = {k: v.detach().clone() for k, v in net.state_dict().items()}      # the training steps below move the weights
    free0 = torch.cuda.mem_get_info(dev)[0]
    net.train()
    tr = DetectorTrainer(net, lr=4e-4)
    step = lambda: tr.step(vox, sync=False)
    ms = timed(step, 2, 5) * 1e3
    free1 = torch.cuda.mem_get_info(dev)[0]
    mem = (C.c_size_t * 4)()
    _lib.check(net._engine.ctx.lib.nm_ctx_memory(net._engine.ctx.handle, mem), "ctx_memory")
    out["train_memory_gb"] = dict(inference_workspace=mem[0] / 1e9, training_arena=mem[1] / 1e9, weight_gradient_side_block=mem[2] / 1e9,
                                  weights_and_packs=mem[3] / 1e9, note="ctx-owned device memory after the fp32-equivalent training steps (nm_ctx_memory)")
    out["train"] = dict(value=world * B_PER_GPU * T / (ms * 1e-3), unit="voxel-frames/s", ms_per_step=ms, steps=5, warmup=2,
                        workload="detector-mode training step (train.py:376-412): forward with retained activations + backward of the 11 "
                                 "AIST-weighted losses + gradient all-reduce (2 bucket chunks) + fused Adam; 64^3, T=16, B=4 clips per GPU",
                        dtype="f32 storage, conv products as 3x f16-split MFMA (fp32-equivalent)", n_gpus=world,
                        hbm_gb_taken_by_training=(free0 - free1) / 2 ** 30,
                        algorithmic_tflop_per_step=3.0 * STEP_TFLOP,
                        frac_of_f16_mfma_peak=3.0 * STEP_TFLOP / (ms * 1e-3) / F16_MFMA_PEAK_TFLOPS)
    # the same step in the reduced-precision conv mode (BASELINE configs[2] names bf16): products of fp16-rounded operands, fp32
    # accumulation / storage / master weights - see include/nm355.h nm_set_conv_mode 3
    with torch.no_grad():
        net.load_state_dict(saved)
    net.set_conv_mode("f16")
    tr = DetectorTrainer(net, lr=4e-4)
    step = lambda: tr.step(vox, sync=False)
    ms = timed(step, 2, 5) * 1e3
    out["train_f16"] = dict(value=world * B_PER_GPU * T / (ms * 1e-3), unit="voxel-frames/s", ms_per_step=ms, steps=5, warmup=2,
                            workload="the `train` step in conv mode 'f16'",
                            dtype="f32 storage and accumulation, conv products of fp16-rounded operands (1 f16 MFMA per product; gradients "
                                  "within 1e-2 global L2 of fp64, tests/test_train_detector_gpu.py::test_detector_gradients_f16_mode)",
                            n_gpus=world, algorithmic_tflop_per_step=3.0 * STEP_TFLOP,
                            frac_of_f16_mfma_peak=3.0 * STEP_TFLOP / (ms * 1e-3) / F16_MFMA_PEAK_TFLOPS)
    with torch.no_grad():
        net.load_state_dict(saved)
    # BASELINE configs[2] in its NAMED precision: the 'f16' arithmetic with bfloat16 storage of every activation / activation gradient of
    # >= 32^3 voxels per frame that the training forward keeps (conv mode 4; fp32 master weights, GroupNorm statistics, accumulators, Adam)
    # (a context of its own: a context's arenas only grow, and this mode's point is the halved training arena)
    from neural_marionette_amd import NeuralMarionette as _NM
    net_b = _NM(net.options)
    net_b.load_state_dict(saved)
    net_b = net_b.to(dev).train()
    net_b.anneal(1)
    net_b.set_conv_mode("bf16")
    tr = DetectorTrainer(net_b, lr=4e-4)
    step = lambda: tr.step(vox, sync=False)
    ms = timed(step, 2, 5) * 1e3
    mem2 = (C.c_size_t * 4)()
    lib_b, h_b = net_b._engine.ctx.lib, net_b._engine.ctx.handle
    _lib.check(lib_b.nm_ctx_memory(h_b, mem2), "ctx_memory")
    # its own roofline: three more steps with every matrix-core launch on either stream bracketed by HIP events (durations include the
    # contention between the step's three queues); one MFMA per algorithmic product in this mode, so issued = algorithmic
    tr.bucket.time_collectives = True
    _lib.check(lib_b.nm_prof_enable(h_b, 2), "prof_enable")
    for _ in range(3):
        step()
    torch.cuda.synchronize(dev)
    _lib.check(lib_b.nm_prof_enable(h_b, 0), "prof_enable")
    fam_b = prof_families(lib_b, h_b, _lib)
    ar_ms = tr.bucket.last_allreduce_ms()
    top_b = [dict(kernel=nm, ms_per_step=m3 / 3.0, launches_per_step=n3 / 3.0, algorithmic_tflop_per_step=fl3 / 3.0 / 1e12,
                  achieved_tflops=fl3 / (m3 * 1e-3) / 1e12, frac_of_f16_mfma_peak=fl3 / (m3 * 1e-3) / 1e12 / F16_MFMA_PEAK_TFLOPS)
             for nm, (m3, fl3, n3) in sorted(fam_b.items(), key=lambda kv: -kv[1][0])[:3] if m3 > 0]
    roof_b = None
    if top_b:
        roof_b = dict(bound="mfma", kernel=top_b[0]["kernel"], achieved=top_b[0]["achieved_tflops"], peak=F16_MFMA_PEAK_TFLOPS, unit="TFLOP/s",
                      frac=top_b[0]["frac_of_f16_mfma_peak"], top3=top_b,
                      matrix_core_ms_per_step_all_families=sum(v[0] for v in fam_b.values()) / 3.0,
                      note="event-timed launches of the step's conv / weight-gradient families on all three queues (durations include their "
                           "mutual contention); one f16 MFMA per algorithmic product in this mode: issued = algorithmic")
    del tr, net_b
    out["train_bf16"] = dict(value=world * B_PER_GPU * T / (ms * 1e-3), unit="voxel-frames/s", ms_per_step=ms, steps=5, warmup=2,
                             workload="the `train` step in conv mode 'bf16' (BASELINE configs[2]: 64^3, T=16, B=4 clips per GPU, bf16)",
                             dtype="bfloat16 storage of the training path's activations and activation gradients (>= 32^3 voxels per frame), conv products of "
                                   "fp16-rounded operands with fp32 accumulation, fp32 master weights / GroupNorm statistics / Adam "
                                   "(tests/test_storage16_gpu.py states the gradient bounds)",
                             n_gpus=world, algorithmic_tflop_per_step=3.0 * STEP_TFLOP,
                             frac_of_f16_mfma_peak=3.0 * STEP_TFLOP / (ms * 1e-3) / F16_MFMA_PEAK_TFLOPS,
                             training_arena_gb=mem2[1] / 1e9, weight_gradient_side_block_gb=mem2[2] / 1e9,
                             roofline=roof_b, allreduce_ms=ar_ms,
                             allreduce_note=("event pair on the bucket's side stream around the step's two gradient all-reduce chunks (RCCL)" if ar_ms is not None
                                             else "no collective was issued (one rank without a process group): measured when N > 1"),
                             note="memory: ctx-owned arenas of this mode's own context (nm_ctx_memory); the fp32-storage figures are in train_memory_gb")
    with torch.no_grad():
        net.load_state_dict(saved)
    # the second regime of train.py (pretrained_mode 1): detector frozen and run forward only, the VRNN trained through its BPTT kernels
    net.set_conv_mode("split16")
    from neural_marionette_amd.train import LearnerTrainer
    lt = LearnerTrainer(net, lr=4e-4)
    lstep = lambda: lt.step(vox, eps=eps, sync=False)
    ms = timed(lstep, 2, 5) * 1e3
    out["train_learner"] = dict(value=world * B_PER_GPU * T / (ms * 1e-3), unit="voxel-frames/s", ms_per_step=ms, steps=5, warmup=2,
                                workload="learner-mode training step (train.py:376-412, pretrained_mode 1): detector forward (frozen) + HSVRNNBVH.encode "
                                         "forward and backward through time + gradient all-reduce + Adam; 64^3, T=16, B=4 clips per GPU",
                                dtype="f32 (VRNN), detector forward fp32-equivalent split-fp16", n_gpus=world)
    with torch.no_grad():
        net.load_state_dict(saved)
    # the same step with the frozen detector run WITHOUT its voxel decoder and losses (LearnerTrainer(lean=True), nm_detector_keypoints):
    # the learner's loss reads only keypoints and affinity; losses / gradients / updated weights bit-identical to the step above
    # (tests/test_network_gpu.py::test_lean_learner_step_is_bit_identical_to_the_full_one).  Reported beside, not instead of, the
    # reference-shaped step, which executes the whole detector forward under no_grad as neural_marionette.py:45-47 does.
    lt2 = LearnerTrainer(net, lr=4e-4, lean=True)
    lstep2 = lambda: lt2.step(vox, eps=eps, sync=False)
    ms = timed(lstep2, 2, 5) * 1e3
    out["train_learner_lean"] = dict(value=world * B_PER_GPU * T / (ms * 1e-3), unit="voxel-frames/s", ms_per_step=ms, steps=5, warmup=2,
                                     workload="learner-mode training step with a keypoints-only detector pass (no voxel decoder, no detector losses: "
                                              "the learner's loss does not read them); same weight update as train_learner",
                                     dtype="f32 (VRNN), detector encoder fp32-equivalent split-fp16", n_gpus=world)
    with torch.no_grad():
        net.load_state_dict(saved)
    net.control_active({"detector": True, "learner": True})
    net.set_conv_mode("f16")
    net.eval()
    ms = timed(fwd, 1, 4) * 1e3
    out["f16_forward"] = dict(value=world * B_PER_GPU * T / (ms * 1e-3), unit="voxel-frames/s", ms_per_step=ms, steps=4,
                              dtype="conv products of fp16-rounded operands, f32 elsewhere (outside the 1e-4 parity contract: keypoints "
                                    "within 2e-3 of the CPU reference)",
                              frac_of_f16_mfma_peak=STEP_TFLOP / (ms * 1e-3) / F16_MFMA_PEAK_TFLOPS)
    net.set_conv_mode("split16")
    # BASELINE configs[3] (D-FAUST-shaped 96^3, T = 8, B = 2: the large-grid stress case) and configs[4] (generation rollout latency,
    # Tcond = 5 posterior + 64 prior steps, B = 1) - secondary, rank 0's GPU only, each with its own context
    if world == 1:
        try:
            out.update(other_configs(dev, timed))
        except Exception as e:                                  # (never at the expense of the headline line)
            out["other_configs_error"] = repr(e)
    return out


def other_configs(dev, timed):
    from neural_marionette_amd import NeuralMarionette, HotPathOptions, synth
    res = {}
    o4 = HotPathOptions(grid_size=96)
    n4 = NeuralMarionette(o4); n4.load_state_dict(synth.make_state_dict(o4, seed=7, variant="peaky")); n4 = n4.to(dev).eval(); n4.anneal(1)
    v4 = synth.figure_clip(2, 8, 96, seed=3).to(dev)
    e4 = synth.make_eps((8, 10, 2, o4.nlatent_kypt), seed=4).to(dev)
    acts = {"detector": True, "learner": True}

    def f4():
        with torch.no_grad():
            return n4(v4, acts, eps=e4)
    ms = timed(f4, 2, 5) * 1e3
    res["config4_96cubed"] = dict(value=16 / (ms * 1e-3), unit="voxel-frames/s (96^3 frames)", ms_per_step=ms, steps=5,
                                  workload="full NeuralMarionette.forward, 96^3, T=8, B=2 (parity: tests/test_network_gpu.py::test_config4_96cubed_vs_reference_fixture)",
                                  voxels_per_s=16 * 96 ** 3 / (ms * 1e-3))
    del n4, v4
    o5 = HotPathOptions(grid_size=32, Tcond=5)
    sd5 = synth.make_state_dict(o5, seed=21, variant="default")
    n5 = NeuralMarionette(o5); n5.load_state_dict(sd5); n5 = n5.to(dev).eval(); n5.anneal(1)
    K, Z, Tc, Tt = o5.nkeypoints, o5.nlatent_kypt, 5, 69
    kp = (torch.rand(1, Tc, K, 4, generator=torch.Generator().manual_seed(1)) * 1.6 - 0.8).to(dev)
    ep = synth.make_eps((Tc, 10, 1, Z), seed=50).to(dev); er = synth.make_eps((Tt - Tc, 1, Z), seed=51).to(dev)
    with torch.no_grad():
        aff = n5.kypt_detector.get_affinity()

    def f5():
        with torch.no_grad():
            return n5.dyna_module.generate(kp, aff, Ttot=Tt, Tcond=Tc, eps_post=ep, eps_prior=er)
    ms = timed(f5, 3, 20) * 1e3
    res["config5_rollout"] = dict(value=ms * 1e3 / Tt, unit="us per generated step", higher_is_better=False, ms_per_rollout=ms, steps_per_rollout=Tt,
                                  workload="HSVRNNBVH.generate, Tcond=5 posterior + 64 prior steps, B=1, K=24 (vis_generation.py:81-136)")
    return res


def dist_selftest(world, rank, local):
    """The multi-rank plumbing of main() without the GPU work: rendezvous, barrier, all-reduce of ones, MAX-reduce of a time."""
    import torch.distributed as dist
    on_gpu = torch.cuda.is_available()
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    dev = torch.device("cuda", local) if on_gpu else torch.device("cpu")
    if on_gpu:
        torch.cuda.set_device(dev)
        dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    dist.barrier()
    t0 = time.perf_counter()
    seen = torch.ones(1, device=dev)
    dist.all_reduce(seen)
    tt = torch.tensor([time.perf_counter() - t0 + rank], device=dev, dtype=torch.float64)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    if rank == 0:
        print(json.dumps(dict(selftest="distributed", n_gpus=world, max_over_ranks_checks=bool(tt.item() >= world - 1),
                              distributed=dict(world_size=world, ranks_seen_by_allreduce=int(seen.item()), backend=dist.get_backend(),
                                               device_count=torch.cuda.device_count()))), flush=True)
    dist.barrier()
    dist.destroy_process_group()


def main():
    args = parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    one_gpu = args.ranks_on_one_gpu > 0
    if one_gpu:                           # diagnostic: every rank on device 0
        if world != args.ranks_on_one_gpu:
            raise SystemExit(f"bench.py: --ranks-on-one-gpu {args.ranks_on_one_gpu} but the launcher started WORLD_SIZE={world} ranks")
        args.gpus, local = world, 0
        args.no_extras = args.no_cpu_baseline = True
    if args.gpus is None:                 # `torchrun --nproc-per-node N bench.py` without --gpus: the launcher's rank count is the request
        args.gpus = world
    if world != args.gpus:                # an explicit --gpus that the launcher did not honour is an error
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    if args.dist_selftest:
        return dist_selftest(world, rank, local)
    # under a torch.distributed launcher the collectives of the timing protocol run even with ONE rank (real RCCL on a one-GPU box:
    # tests/test_bench_contract_gpu.py::test_forward_line_under_torchrun_one_rank)
    dist_on = world > 1 or ("RANK" in os.environ and "WORLD_SIZE" in os.environ and "MASTER_ADDR" in os.environ)
    if dist_on:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local)
        if one_gpu:
            dist.init_process_group("gloo")          # (RCCL refuses two ranks on one device; the timing collectives carry host scalars)
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback exists for the product path)")
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)

    from neural_marionette_amd import NeuralMarionette, HotPathOptions, synth, _lib
    opts = HotPathOptions(grid_size=G)
    sd = synth.make_state_dict(opts, seed=42, variant="peaky")
    net = NeuralMarionette(opts)
    net.load_state_dict(sd)
    net = net.to(dev).eval()
    net.anneal(1)
    net.set_conv_mode(args.conv_mode)
    acts = {"detector": True, "learner": True}
    vox = synth.figure_clip(B_PER_GPU, T, G, seed=1 + rank).to(dev)           # resident in HBM
    eps = synth.make_eps((T, S, B_PER_GPU, opts.nlatent_kypt), seed=100 + rank).to(dev)

    def barrier():
        torch.cuda.synchronize(dev)
        if dist_on:
            dist.barrier()
            torch.cuda.synchronize(dev)

    if args.workload == "train":
        from neural_marionette_amd.train import DetectorTrainer
        net.train()
        trainer = DetectorTrainer(net, lr=4e-4)
        step = lambda: trainer.step(vox, sync=False)
    else:
        def step():                      # inference forward, as the reference runs it outside training (train.py:441, vis_*.py)
            with torch.no_grad():
                return net(vox, acts, eps=eps)
    for _ in range(args.warmup):
        step()
    eng = net._engine
    lib, h = eng.ctx.lib, eng.ctx.handle
    _lib.check(lib.nm_prof_enable(h, 3), "prof_enable")      # conv launches of >= 20 GFLOP on the ctx stream, bracketed by HIP events
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    _lib.check(lib.nm_prof_enable(h, 0), "prof_enable")
    cdev = torch.device("cpu") if one_gpu else dev      # where the collectives' scalars live
    if dist_on:
        tt = torch.tensor([dt], device=cdev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    fam_main = prof_families(lib, h, _lib) if rank == 0 else {}        # timed region, launches on the ctx stream (clean durations)
    # the same families including the launches the library puts on its side stream (clip-mean net / VRNN beside the frame stack):
    # three more steps after the timed region, every conv launch on either stream bracketed by events (durations include contention)
    fam_all = {}
    if rank == 0:
        _lib.check(lib.nm_prof_enable(h, 2), "prof_enable")
        for _ in range(3):
            step()
        torch.cuda.synchronize(dev)
        _lib.check(lib.nm_prof_enable(h, 0), "prof_enable")
        fam_all = prof_families(lib, h, _lib)

    # what each rank saw (the driver can check that RCCL really spanned N devices)
    ranks_seen = world
    if dist_on:
        seen = torch.ones(1, device=cdev)
        dist.all_reduce(seen)
        ranks_seen = int(seen.item())

    extra = {}
    if args.workload == "forward" and not args.no_extras:
        extra = extra_measurements(net, vox, eps, acts, dev, barrier, dist_on, world)

    if rank == 0:
        frames = world * B_PER_GPU * T * args.steps
        # dominant kernel = the conv family with the largest event-timed total per step over BOTH streams (what a rocprof summary
        # ranks by); its `achieved` comes from its launches on the ctx stream inside the timed region (clean durations)
        # f16 matrix-core families; THREE products per algorithmic product only in conv mode 1 (split-fp16) - the one-product modes
        # (3 'f16' / 4 'bf16': conv_f16q2, conv_f16r and the SINGLE instantiations of the others) issue one (advisor finding, round 5)
        is_f16 = lambda nm: nm.startswith("conv_f16") or nm.startswith("conv_pool_f16") or nm.startswith("conv_up2c") or nm.startswith("wgrad16")
        is_split = lambda nm: eng.conv_mode == 1 and is_f16(nm) and not (nm.startswith("conv_f16q2") or nm.startswith("conv_f16r"))
        ranked = sorted(fam_all.items(), key=lambda kv: -kv[1][0])
        top3 = []
        for nm, (ms_a, fl_a, n_a) in ranked[:3]:
            ms_m, fl_m, n_m = fam_main.get(nm, (0.0, 0.0, 0))
            pk = F16_MFMA_PEAK_TFLOPS if (is_f16(nm) and eng.conv_mode != 0) else FP32_MFMA_PEAK_TFLOPS
            top3.append(dict(kernel=nm, ms_per_step_all_streams=ms_a / 3.0, launches_per_step_all_streams=n_a / 3.0,
                             ms_per_step_ctx_stream=ms_m / args.steps, launches_per_step_ctx_stream=n_m / args.steps,
                             achieved_tflops_ctx_stream=(fl_m / (ms_m * 1e-3) / 1e12 if ms_m else None),
                             frac_of_peak=(fl_m / (ms_m * 1e-3) / 1e12 / pk if ms_m else None), peak_tflops=pk))
        best = next(((nm, *fam_main[nm]) for nm, _ in ranked if nm in fam_main), None)
        roof = None
        if best:
            name, ms, fl, n = best
            ach = fl / (ms * 1e-3) / 1e12
            try:
                traffic, traffic_detail = pmc_traffic(name)
            except Exception as e:                                  # an auxiliary file must never take the bench line down
                traffic, traffic_detail = None, dict(file=None, note=f"profiles/{ROUND}_pmc_traffic.json could not be read: {e!r}")
            split = is_split(name)
            on_f16 = is_f16(name) and eng.conv_mode != 0
            peak = F16_MFMA_PEAK_TFLOPS if on_f16 else FP32_MFMA_PEAK_TFLOPS
            roof = dict(bound="mfma", achieved=ach, peak=peak, unit="TFLOP/s", frac=ach / peak, traffic=traffic,
                        traffic_detail=traffic_detail, top3=top3,
                        kernel=name, launches=n, avg_launch_ms=ms / n, kernel_time_share=ms * 1e-3 / dt,
                        note=("achieved = algorithmic fp32 conv FLOPs (2*voxels*Cout*Cin*k^3) / event-timed launch time; "
                              + ("this kernel issues 3 f16 MFMA products per algorithmic product (hi/lo operand split, "
                                 "f32 accumulate), so issued = 3 x achieved; peak = dense f16 MFMA"
                                 if split else ("one f16 MFMA product per algorithmic product (operands rounded to fp16); peak = dense f16 MFMA"
                                                if on_f16 else "peak = dense fp32 MFMA"))))
            if split:
                roof["issued"] = 3.0 * ach
                roof["frac_issued"] = 3.0 * ach / peak
                roof["split_ceiling_tflops"] = peak / 3.0          # three MFMA products per algorithmic product: what 100 % MFMA issue would give
                roof["frac_of_split_ceiling"] = ach / (peak / 3.0)
                roof["algorithmic_vs_fp32_mfma_peak"] = ach / FP32_MFMA_PEAK_TFLOPS
        cpu, parity = (None, {})
        if world == 1 and not args.no_cpu_baseline and args.workload == "forward":
            cpu, parity = cpu_baseline(sd, opts, net, dev)
        line = dict(
            metric="voxel-frames/sec (64^3, T=16)", value=frames / dt, unit="voxel-frames/s",
            n_gpus=1 if one_gpu else world,
            **(dict(diagnostic=f"{world} processes (one nm_ctx each) sharing device 0: aggregate of all of them; NOT a scaling figure", ranks_on_one_gpu=world) if one_gpu else {}), steps=args.steps, warmup=args.warmup, ms_per_step=dt / args.steps * 1e3,
            higher_is_better=True, scaling="weak", vs_baseline=None,
            dtype=("f32 (conv products as 3x f16-split MFMA with f32 accumulate, fp32-equivalent; everything else f32)"
                   if eng.conv_mode == 1 else ("f16 conv products (operands rounded to fp16, 1 MFMA per product), f32 accumulation and storage"
                                               if eng.conv_mode == 3 else ("bf16 storage of the training path (activations / activation gradients >= 32^3 voxels per frame), "
                                                                            "f16 conv products, f32 accumulation, master weights, statistics and optimizer"
                                                                            if eng.conv_mode == 4 else "f32"))),
            data="synthetic",
            config=dict(workload=("AIST++-shaped synthetic clips 64^3 T=16 B=4/GPU, full NeuralMarionette.forward "
                                  "(detector + 11 losses + HSVRNNBVH.encode, best-of-10), fp32, random-init weights") if args.workload == "forward" else
                                 ("AIST++-shaped synthetic clips 64^3 T=16 B=4/GPU, detector-mode training step (forward + backward of the "
                                  "11 weighted losses + flat-bucket gradient all-reduce + Adam), fp32, random-init weights"),
                        grid=G, T=T, clips_per_gpu=B_PER_GPU, global_clips=world * B_PER_GPU, conv_mode=args.conv_mode,
                        parallelism=f"clip-sharded x{world} (no data-path collective)"),
            roofline=roof, cpu_baseline=cpu, kypt_l2_vs_cpu=parity.get("kypt_l2"), latent_linf_vs_cpu=parity.get("latent_linf"),
            parity_vs_cpu=parity or None,
            step_roofline=dict(algorithmic_tflop_per_step=STEP_TFLOP, achieved=STEP_TFLOP * world / (dt / args.steps),
                               unit="TFLOP/s", peak=F16_MFMA_PEAK_TFLOPS * world if eng.conv_mode else FP32_MFMA_PEAK_TFLOPS * world,
                               frac=STEP_TFLOP / (dt / args.steps) / (F16_MFMA_PEAK_TFLOPS if eng.conv_mode else FP32_MFMA_PEAK_TFLOPS),
                               note="whole forward step: 99.15 GFLOP per voxel-frame (BASELINE.md, the reference's dense fp32 conv "
                                    "arithmetic) x 64 frames; the split-fp16 kernels issue 3 f16 MFMA products per algorithmic product"
                               ) if args.workload == "forward" else None,
            distributed=dict(world_size=world, ranks_seen_by_allreduce=ranks_seen, backend=(dist.get_backend() if dist_on else None),
                             device_count=torch.cuda.device_count(), device=torch.cuda.get_device_name(dev)),
            **extra,
        )
        if args.workload == "train":
            mem = (C.c_size_t * 4)()
            _lib.check(lib.nm_ctx_memory(h, mem), "ctx_memory")
            line["train_memory_gb"] = dict(inference_workspace=mem[0] / 1e9, training_arena=mem[1] / 1e9, weight_gradient_side_block=mem[2] / 1e9,
                                           weights_and_packs=mem[3] / 1e9, note="ctx-owned device memory after this mode's training steps (nm_ctx_memory)")
        print(json.dumps(line), flush=True)
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
