"""N>1 path on CPU: world_size-2 gloo group; clip partition, gather and loss averaging give
the single-process result (the per-clip computation is replaced by a deterministic stand-in,
the HIP forward itself needs the GPU)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from neural_marionette_amd.dist import clip_shard, gather_clips, mean_losses


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _per_clip(vox):                       # stand-in for the per-clip forward: (b,T,1,G,G,G) -> (b,T,3)
    return torch.stack([vox.sum(dim=(2, 3, 4, 5)), vox.mean(dim=(2, 3, 4, 5)), vox.amax(dim=(2, 3, 4, 5))], dim=-1)


def _worker(rank, world, port, n_clips, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = torch.Generator().manual_seed(0)
    vox = (torch.rand(n_clips, 3, 1, 8, 8, 8, generator=g) < 0.2).float()
    a, b = clip_shard(n_clips, rank, world)
    local = _per_clip(vox[a:b])
    full = gather_clips(local, n_clips)
    loss = mean_losses({"recon_loss": local[..., 1].mean()}, b - a, n_clips)
    if rank == 0:
        q.put((full, loss["recon_loss"], _per_clip(vox), _per_clip(vox)[..., 1].mean()))
    dist.barrier()
    dist.destroy_process_group()


def test_clip_shard_partition():
    for n in (1, 4, 5, 7, 32):
        for w in (1, 2, 3, 8):
            spans = [clip_shard(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_world2_gloo_matches_single_process():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    n_clips = 5                           # uneven split 3 + 2
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_clips, q)) for r in range(2)]
    for p in procs: p.start()
    full, loss, ref, ref_loss = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert torch.equal(full, ref)
    assert abs(float(loss) - float(ref_loss)) < 1e-6


def _grad_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from neural_marionette_amd.train import allreduce_mean_
    g = torch.Generator().manual_seed(100 + rank)
    grads = [torch.randn(7, 5, generator=g), torch.randn(11, generator=g), torch.randn(3, 2, 2, generator=g)]
    mine = [t.clone() for t in grads]
    allreduce_mean_(mine)
    if rank == 0:
        q.put((grads, mine))
    dist.barrier()
    dist.destroy_process_group()


def test_gradient_bucket_allreduce_world2():
    """the flat-bucket gradient all-reduce of the learner training step (RCCL on the GPU box, gloo here)"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_grad_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs: p.start()
    g0, avg = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    gen = torch.Generator().manual_seed(101)
    g1 = [torch.randn(7, 5, generator=gen), torch.randn(11, generator=gen), torch.randn(3, 2, 2, generator=gen)]
    for a, b, c in zip(g0, g1, avg):
        assert torch.allclose(c, (a + b) / 2, atol=1e-7)


# ---- the real DetectorTrainer.step control flow (bucket views, chunked all-reduce, averaging, Adam) with the three device
# ---- operations replaced by CPU stand-ins: world size 2 over gloo ------------------------------------------------------------
def _stub_grad(name, shape, rank):
    g = torch.Generator().manual_seed((sum(name.encode()) * 7919 + 13 * rank) % (2 ** 31))
    return torch.randn(*shape, generator=g)


def _trainer_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from neural_marionette_amd import NeuralMarionette, HotPathOptions
    from neural_marionette_amd.train import DetectorTrainer

    class CpuStandIn(DetectorTrainer):
        def _forward_backward(self, vox, named, bucket):
            for n, p in named:
                bucket.views[n].copy_(_stub_grad(n, p.shape, rank))
            if getattr(self, "poison", False) and rank == 1:
                bucket.flat[bucket.flat.numel() // 2] = float("inf")          # ONE rank's gradient overflowed
            bucket.reduce_chunk(0)                   # as the HIP step does once the decoder's gradients exist
            return torch.arange(11, dtype=torch.float32) * (rank + 1)

        def _adam(self, params, grads, m, v, ok=None):        # torch.optim.Adam's update, written out (ok == 0: nm_adam_step_multi_ok's no-op)
            if ok is not None and float(ok) == 0.0:
                return
            b1, b2 = self.betas
            for p, g, mm, vv in zip(params, grads, m, v):
                mm.mul_(b1).add_(g, alpha=1 - b1); vv.mul_(b2).addcmul_(g, g, value=1 - b2)
                denom = (vv / (1 - b2 ** self.t)).sqrt_().add_(self.eps)
                p.data.addcdiv_(mm / (1 - b1 ** self.t), denom, value=-self.lr)

    torch.manual_seed(3)
    net = NeuralMarionette(HotPathOptions(grid_size=32))
    before = {k: v.clone() for k, v in net.state_dict().items()}
    tr = CpuStandIn(net, lr=1e-2)
    out = tr.step(torch.zeros(1, 2, 1, 32, 32, 32))
    assert len(tr.bucket.chunks) == 2 and tr.bucket.chunks[0][1] == tr.bucket.chunks[1][0] > 0
    assert tr.bucket.chunks[1][1] == tr.bucket.flat.numel() == sum(p.numel() for p in net.kypt_detector.parameters())
    dec = [n for n in tr.bucket.views if n.startswith("kypt_detector.kypt_to_vox.")]
    assert sum(tr.bucket.views[n].numel() for n in dec) == tr.bucket.chunks[0][1]
    worst = 0.0
    for n, p in net.kypt_detector.named_parameters():
        name = "kypt_detector." + n
        g = (_stub_grad(name, p.shape, 0) + _stub_grad(name, p.shape, 1)) / 2
        assert torch.allclose(p.grad, g, atol=1e-7), name                      # .grad = the bucket view holding the mean gradient
        assert p.grad.data_ptr() == tr.bucket.views[name].data_ptr()
        want = before[name] - 1e-2 * g / (g.abs() + 1e-8)                        # first Adam step: m_hat / (sqrt(v_hat) + eps)
        worst = max(worst, float((p.detach() - want).abs().max()))
    # a step in which ONE rank's gradient holds an inf: the sum carries it to every rank, the bucket's finiteness flag is 0 on both and
    # the optimizer leaves parameters and moments exactly as they are (no zeroed-gradient pseudo step); check=True raises instead
    after1 = {k: v.clone() for k, v in net.state_dict().items()}
    mom1 = {i: (m.clone(), v.clone()) for i, (m, v) in tr.state.items()}
    tr.poison = True
    tr.step(torch.zeros(1, 2, 1, 32, 32, 32))
    for k, v in net.state_dict().items():
        assert torch.equal(v, after1[k]), k
    for i, (m, v) in tr.state.items():
        assert torch.equal(m, mom1[i][0]) and torch.equal(v, mom1[i][1])
    raised = False
    try:
        tr.step(torch.zeros(1, 2, 1, 32, 32, 32), check=True)
    except Exception as e:
        raised = "non-finite" in str(e)
    assert raised, "step(check=True) must raise on a non-finite gradient bucket"
    tr.poison = False
    tr.step(torch.zeros(1, 2, 1, 32, 32, 32))          # and training goes on
    assert any(not torch.equal(v, after1[k]) for k, v in net.state_dict().items())
    if rank == 0:
        q.put((worst, float(out["loss"]), sorted(out)))
    dist.barrier()
    dist.destroy_process_group()


def test_detector_trainer_step_world2_with_cpu_stand_ins():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_trainer_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs: p.start()
    worst, loss, keys = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert worst < 1e-6
    from neural_marionette_amd.train import DETECTOR_LOSS_WEIGHTS
    from neural_marionette_amd.spec import DETECTOR_LOSS_KEYS
    assert abs(loss - sum(DETECTOR_LOSS_WEIGHTS[k] * i for i, k in enumerate(DETECTOR_LOSS_KEYS))) < 1e-3
    assert "loss" in keys and "recon_loss" in keys


# ---- bench.py --gpus N outside a launcher: the file starts its own N ranks (child torch.distributed.run, before torch is imported)
def _bench(*args, env=None):
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), *args], capture_output=True, text=True, timeout=600, cwd=root, env=e)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    return r, [json.loads(ln) for ln in lines]


def test_bench_gpus2_starts_two_ranks_gloo():
    """`python bench.py --gpus 2 ...` (the form the driver uses for N = 1) must yield n_gpus 2 / world_size 2: here with the
    plumbing-only --dist-selftest on gloo (no GPU in this container), the same launcher / rendezvous / collectives the bench uses."""
    r, lines = _bench("--gpus", "2", "--dist-selftest")
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(lines) == 1, r.stdout
    d = lines[0]
    assert d["n_gpus"] == 2 and d["distributed"]["world_size"] == 2 and d["distributed"]["ranks_seen_by_allreduce"] == 2
    assert d["distributed"]["backend"] == "gloo" and d["max_over_ranks_checks"] is True


def test_bench_world_size_mismatch_is_an_error():
    """Under an external launcher WORLD_SIZE must equal --gpus (a run that silently degenerated to another rank count)."""
    r, lines = _bench("--gpus", "4", "--dist-selftest", env={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and not lines
    assert "WORLD_SIZE=1" in (r.stderr + r.stdout)


def test_detector_trainer_rejects_unknown_loss_names_and_tracks_weight_edits():
    from neural_marionette_amd import NeuralMarionette, HotPathOptions
    from neural_marionette_amd.train import DetectorTrainer
    import pytest
    net = NeuralMarionette(HotPathOptions(grid_size=32))
    with pytest.raises(KeyError):
        DetectorTrainer(net, weights={"recon_loss": 1.0, "no_such_loss": 2.0})
    tr = DetectorTrainer(net)
    w0 = tr._weight_vector(torch.device("cpu")).clone()
    tr.weights["recon_loss"] = 7.0                        # a per-epoch schedule edits the dict in place
    w1 = tr._weight_vector(torch.device("cpu"))
    assert w0[tr.loss_keys.index("recon_loss")] == 100.0 and w1[tr.loss_keys.index("recon_loss")] == 7.0
    # frozen parameters stay in the gradient bucket (the library writes every detector gradient), only Adam skips them
    net.kypt_detector.affinity_params.requires_grad = False
    names = [n for n, _ in tr._named()]
    assert "kypt_detector.affinity_params" in names and len(names) == len(list(net.kypt_detector.parameters()))


def test_bench_adopts_world_size_when_gpus_is_not_given():
    """`torchrun --nproc-per-node N bench.py` without --gpus: the launcher's rank count is the request (only an explicit mismatch fails)."""
    import socket
    sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
    r, lines = _bench("--dist-selftest", env={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(lines) == 1 and lines[0]["n_gpus"] == 1 and lines[0]["distributed"]["world_size"] == 1


# ---- multi-GPU readiness without the node (round 5): the launcher with EIGHT ranks, and the trainer's clip-weighted mean on an uneven
# ---- split over four ranks --------------------------------------------------------------------------------------------------------
def test_bench_gpus8_dist_selftest_gloo():
    """`python bench.py --gpus 8 --dist-selftest`: the 8-rank launch the driver's scaling run uses (child torch.distributed.run on
    127.0.0.1, rendezvous, barrier, all-reduce of ones = 8, MAX-reduce over ranks), on gloo here."""
    r, lines = _bench("--gpus", "8", "--dist-selftest", env={"OMP_NUM_THREADS": "1"})
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(lines) == 1, r.stdout
    d = lines[0]
    assert d["n_gpus"] == 8 and d["distributed"]["world_size"] == 8 and d["distributed"]["ranks_seen_by_allreduce"] == 8
    assert d["distributed"]["backend"] == "gloo" and d["max_over_ranks_checks"] is True


def _clip_grad(name, shape, clip):
    g = torch.Generator().manual_seed((sum(name.encode()) * 104729 + 31 * clip) % (2 ** 31))
    return torch.randn(*shape, generator=g)


def _uneven_worker(rank, world, port, n_clips, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    from neural_marionette_amd import NeuralMarionette, HotPathOptions
    from neural_marionette_amd.train import DetectorTrainer
    a, b = clip_shard(n_clips, rank, world)

    class CpuStandIn(DetectorTrainer):
        def _forward_backward(self, vox, named, bucket):
            # what the HIP step produces: the gradient of THIS rank's clip mean, times the rank's share of the global mean (the
            # library's backward is linear in the dL/dloss vector, which step() scales by self._rank_scale)
            for n, p in named:
                local = sum(_clip_grad(n, p.shape, c) for c in range(a, b)) / (b - a)
                bucket.views[n].copy_(local * self._rank_scale)
            bucket.reduce_chunk(0)
            return torch.full((11,), float(rank))

        def _adam(self, params, grads, m, v, ok=None):
            if ok is not None and float(ok) == 0.0:
                return
            b1, b2 = self.betas
            for p, g, mm, vv in zip(params, grads, m, v):
                mm.mul_(b1).add_(g, alpha=1 - b1); vv.mul_(b2).addcmul_(g, g, value=1 - b2)
                denom = (vv / (1 - b2 ** self.t)).sqrt_().add_(self.eps)
                p.data.addcdiv_(mm / (1 - b1 ** self.t), denom, value=-self.lr)

    torch.manual_seed(3)
    net = NeuralMarionette(HotPathOptions(grid_size=32))
    tr = CpuStandIn(net, lr=1e-2)
    tr.step(torch.zeros(b - a, 2, 1, 32, 32, 32), global_clips=n_clips)
    assert abs(tr._rank_scale - (b - a) * world / n_clips) < 1e-12
    worst = 0.0
    for n, p in net.kypt_detector.named_parameters():
        name = "kypt_detector." + n
        want = sum(_clip_grad(name, p.shape, c) for c in range(n_clips)) / n_clips      # the single-process gradient of the 6-clip batch
        worst = max(worst, float((p.grad - want).abs().max()))
    if rank == 0:
        q.put(worst)
    dist.barrier()
    dist.destroy_process_group()


def test_detector_trainer_uneven_clip_split_world4_is_the_clip_weighted_mean():
    """B = 6 clips over 4 ranks (2 + 2 + 1 + 1, dist.clip_shard): with ``global_clips`` the bucket's sum / world is the CLIP-weighted
    mean - the gradient a single process computes on all 6 clips (every loss is a mean over clips, train.py:376-412) - not the mean of
    the ranks' local means."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_uneven_worker, args=(r, 4, port, 6, q)) for r in range(4)]
    for p in procs: p.start()
    worst = q.get(timeout=600)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert worst < 1e-6, worst
