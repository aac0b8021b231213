"""-m gpu: the training-step kernels around the translator: fused weighted cross-entropy
against the oracle / torch, FusedAdam against torch.optim.Adam and AdamW, and the whole step inside one hipGraph."""
import copy

import pytest
import torch

from tests.util import hhi_args, rel_err, seeded_feats, seeded_state_dict

pytestmark = pytest.mark.gpu
CE_W = [0.266, 0.734]


@pytest.mark.parametrize("B,C,weighted", [(256, 2, True), (3000, 2, True), (5, 7, True), (64, 3, False), (1, 2, True)])
def test_weighted_ce_matches_oracle_and_torch(egx_lib, cuda, B, C, weighted):
    from egot2_amd import functional as F_egx
    from oracle import translator_ref as ref
    g = torch.Generator().manual_seed(B * 31 + C)
    logits = (torch.randn(B, C, generator=g) * 3).to(cuda).requires_grad_(True)
    target = torch.randint(0, C, (B,), generator=g).to(cuda)
    w = torch.rand(C, generator=g) + 0.1 if weighted else None
    loss = F_egx.weighted_cross_entropy(logits, target, None if w is None else w.to(cuda))
    (loss * 1.7).backward()
    l64 = logits.detach().cpu().double().requires_grad_(True)
    want = ref.weighted_ce(l64, target.cpu(), list(w.double()) if w is not None else [1.0] * C)
    (want * 1.7).backward()
    assert abs(loss.item() - want.item()) < 1e-5 * max(1.0, abs(want.item()))
    assert (logits.grad.cpu().double() - l64.grad).abs().max().item() < 1e-6
    t = torch.nn.functional.cross_entropy(logits.detach(), target, weight=None if w is None else w.to(cuda))
    assert abs(loss.item() - t.item()) < 1e-5 * max(1.0, abs(t.item()))


def test_cross_entropy_module_mirrors_reference_criterion(egx_lib, cuda):
    from egot2_amd.train import CrossEntropyLoss
    crit = CrossEntropyLoss(weight=torch.FloatTensor(CE_W)).to(cuda)
    assert list(crit.state_dict().keys()) == ["weight"]        # `criterion.weight` in the reference's Lightning ckpt
    ref = torch.nn.CrossEntropyLoss(weight=torch.FloatTensor(CE_W)).to(cuda)
    x = torch.randn(32, 2, device=cuda)
    y = torch.randint(0, 2, (32,), device=cuda)
    assert abs(crit(x, y).item() - ref(x, y).item()) < 1e-6
    with pytest.raises(ValueError):
        crit(x, y.int())


def _ttm(cuda, seed=5):
    from egot2_amd import hhi_ttm
    m = hhi_ttm.TaskFusionMFTransformer3Task(hhi_args(dropout=0.0))
    m.pos_embed.dropout.p = 0.0
    m.load_state_dict(seeded_state_dict(m, seed))
    return m.to(cuda).train()


@pytest.mark.parametrize("adamw,wd,lr", [(False, 0.0, 5e-4), (True, 1e-4, 1e-4), (False, 1e-2, 1e-3)])
def test_fused_adam_matches_torch(egx_lib, cuda, adamw, wd, lr):
    """Reference optimizers: Adam(lr=5e-4) HHI/tasks/ttm/video_task_2loader.py:62-64, AdamW(1e-4, wd=1e-4)
    HOI/tasks/multitask/video_task.py:624-626. Both models take their gradients from the same HIP path."""
    from egot2_amd.train import FusedAdam
    a, b = _ttm(cuda), _ttm(cuda)
    opt_a = FusedAdam(a.parameters(), lr=lr, weight_decay=wd, adamw=adamw)
    opt_b = (torch.optim.AdamW if adamw else torch.optim.Adam)(b.parameters(), lr=lr, weight_decay=wd)
    w = torch.tensor(CE_W, device=cuda)
    for step in range(6):
        feats = [f.to(cuda) for f in seeded_feats(100 + step, [(8, 15, 256)] * 3)]
        target = torch.randint(0, 2, (8,), generator=torch.Generator().manual_seed(step)).to(cuda)
        for m, opt in ((a, opt_a), (b, opt_b)):
            opt.zero_grad(set_to_none=True)
            torch.nn.functional.cross_entropy(m.forward_features(*feats), target, weight=w).backward()
            opt.step()
    # Elements whose gradient is rounding noise (e.g. the key bias, to which softmax is invariant) move by +-lr per
    # step in either run, so the model-level bound is the largest possible drift; exact update arithmetic is pinned
    # by test_adam_kernel_semantics below.
    for (n, pa), (_, pb) in zip(a.named_parameters(), b.named_parameters()):
        assert (pa - pb).abs().max().item() <= 2.0 * lr * 6, n
        assert (pa - pb).abs().median().item() < 1e-6 + 1e-2 * lr, n
    # the optimizer state is visible per parameter (torch's state_dict layout) and the translator's parameters now
    # live in one flat buffer (a single launch per step)
    st = opt_a.state[a.ln.weight]
    assert st["exp_avg"].shape == a.ln.weight.shape and int(st["step"].item()) == 6
    assert len({p.untyped_storage().data_ptr() for p in a.parameters()}) == 1
    sd = opt_a.state_dict()
    assert len(sd["state"]) == len(list(a.parameters()))


@pytest.mark.parametrize("adamw,wd", [(False, 0.0), (False, 0.05), (True, 0.05)])
def test_adam_kernel_semantics(egx_lib, cuda, adamw, wd):
    """Identical gradients into FusedAdam and torch.optim: parameters must agree to fp32 rounding. Covers stand-alone
    tensors, sizes that are not multiples of 4 and gradients that are unaligned views of a shared buffer."""
    from egot2_amd.train import FusedAdam
    g = torch.Generator().manual_seed(11)
    shapes = [(7,), (128, 33), (1, 3, 128), (5,), (2048, 128)]
    pa = [torch.nn.Parameter(torch.randn(s, generator=g).to(cuda)) for s in shapes]
    pb = [torch.nn.Parameter(p.detach().clone()) for p in pa]
    oa = FusedAdam(pa, lr=1e-2, betas=(0.9, 0.99), eps=1e-8, weight_decay=wd, adamw=adamw)
    ob = (torch.optim.AdamW if adamw else torch.optim.Adam)(pb, lr=1e-2, betas=(0.9, 0.99), eps=1e-8, weight_decay=wd)
    for step in range(8):
        flat = torch.randn(3 + sum(torch.Size(s).numel() + 1 for s in shapes[:3]), generator=g).to(cuda)
        off = 3                                            # the first three gradients are unaligned views of `flat`
        for i, (x, y) in enumerate(zip(pa, pb)):
            if i < 3:
                n = x.numel()
                x.grad = flat[off:off + n].view(x.shape)
                off += n + 1
            else:
                x.grad = torch.randn(x.shape, generator=g).to(cuda)
            y.grad = x.grad.detach().clone()
        oa.step()
        ob.step()
    for x, y in zip(pa, pb):
        assert torch.allclose(x, y, rtol=2e-5, atol=2e-6), (x - y).abs().max().item()


@pytest.mark.parametrize("compute", ["f32", "f32s"])
def test_whole_training_step_in_one_graph(egx_lib, cuda, compute):
    """forward + fused CE + backward + FusedAdam captured once and replayed: parameters must follow the eager run."""
    from egot2_amd.train import CrossEntropyLoss, FusedAdam
    ref_model, m = _ttm(cuda, 3).set_compute(compute), _ttm(cuda, 3).set_compute(compute)
    crit = CrossEntropyLoss(torch.FloatTensor(CE_W)).to(cuda)
    feats = [f.to(cuda) for f in seeded_feats(77, [(16, 15, 256)] * 3)]
    target = torch.randint(0, 2, (16,), generator=torch.Generator().manual_seed(1)).to(cuda)

    def one_step(model, opt):
        opt.zero_grad(set_to_none=True)
        loss = crit(model.forward_features(*feats), target)
        loss.backward()
        opt.step()
        return loss

    opt_r = FusedAdam(ref_model.parameters(), lr=5e-4)
    for _ in range(5):
        one_step(ref_model, opt_r)

    opt = FusedAdam(m.parameters(), lr=5e-4)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        one_step(m, opt)                      # eager step 1: builds the flat parameter buffer before capture
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        loss = one_step(m, opt)
    for _ in range(3):                        # capture does not execute: steps 2..4 by replay, then compare at 4 vs 5?
        graph.replay()
    torch.cuda.synchronize()
    # eager 1 + 3 replays = 4 steps; bring the reference to the same count
    ref4 = _ttm(cuda, 3).set_compute(compute)
    opt4 = FusedAdam(ref4.parameters(), lr=5e-4)
    for _ in range(4):
        one_step(ref4, opt4)
    for (n, pa), (_, pb) in zip(m.named_parameters(), ref4.named_parameters()):
        # atomically accumulated gradients differ in the last bits between runs; noise-level gradients (key bias)
        # then move by +-lr per step in either run
        assert (pa - pb).abs().max().item() <= 2.0 * 5e-4 * 4, n
        assert (pa - pb).abs().median().item() < 1e-5, n
    assert int(opt._step_dev.item()) == 4 and torch.isfinite(loss).item()


def test_graphed_step_helper_follows_the_eager_loop(egx_lib, cuda):
    """train.GraphedStep (the bench's replayed step packaged for a training script): three DIFFERENT batches fed to the captured step must
    leave the parameters where the eager loop over the same batches leaves them (p = 0: no masks to match), the returned loss must be the
    current batch's, and a batch of another shape is refused with an error instead of being run on stale buffers."""
    from egot2_amd.train import CrossEntropyLoss, FusedAdam, GraphedStep
    crit = CrossEntropyLoss(torch.FloatTensor(CE_W)).to(cuda)
    batches = [([f.to(cuda) for f in seeded_feats(300 + i, [(12, 15, 256)] * 3)],
                torch.randint(0, 2, (12,), generator=torch.Generator().manual_seed(i)).to(cuda)) for i in range(4)]
    warm = 2

    def loss_of(model):
        return lambda f, y: crit(model.forward_features(*f), y)

    ref = _ttm(cuda, 3).set_compute("f32s")
    opt_r = FusedAdam(ref.parameters(), lr=5e-4)
    ref_losses = []
    # the helper's `warm` warm-up steps on the example batch leave NO trace (round 6: parameters, Adam moments / step count and the dropout seed are
    # snapshotted and restored around them): the eager loop is just the batches that are fed to the captured step
    for f, y in batches[1:]:
        opt_r.zero_grad(set_to_none=True)
        loss = loss_of(ref)(f, y)
        loss.backward()
        opt_r.step()
        ref_losses.append(loss.item())

    m = _ttm(cuda, 3).set_compute("f32s")
    before = [p.detach().clone() for p in m.parameters()]
    opt_m = FusedAdam(m.parameters(), lr=5e-4)
    step = GraphedStep(loss_of(m), example_inputs=batches[0], params=list(m.parameters()), optimizer=opt_m, warmup=warm)
    torch.cuda.synchronize()
    assert all(torch.equal(a, b) for a, b in zip(before, m.parameters())), "the warm-up steps must not leave updates behind"
    assert int(opt_m._step_dev.item()) == 0 and all(float(bk.exp_avg.abs().max()) == 0.0 for bk in opt_m._buckets.values())
    got = [step(f, y).item() for f, y in batches[1:]]
    torch.cuda.synchronize()
    for a, b in zip(got, ref_losses):
        assert abs(a - b) < 2e-3 * max(1.0, abs(b)), (got, ref_losses)
    for (n, pa), (_, pb) in zip(m.named_parameters(), ref.named_parameters()):
        assert (pa - pb).abs().max().item() <= 2.0 * 5e-4 * 3, n
        assert (pa - pb).abs().median().item() < 2e-5, n
    assert all(p.grad is not None for p in m.parameters() if p.requires_grad)
    with pytest.raises(ValueError, match="one GraphedStep per batch shape"):
        step([f[:5] for f in batches[1][0]], batches[1][1][:5])


def test_graphed_step_draws_fresh_masks_and_refuses_a_host_seed(egx_lib, cuda):
    """GraphedStep on a model WITH dropout: after enable_device_seed() every replay draws fresh masks (the same batch gives different losses from
    replay to replay — no optimizer, so the weights stay put), and without the device-resident seed the capture is refused by the library instead
    of baking one mask into the graph."""
    from egot2_amd import hhi_ttm, _lib
    from egot2_amd.train import CrossEntropyLoss, GraphedStep
    crit = CrossEntropyLoss(torch.FloatTensor(CE_W)).to(cuda)
    feats = [f.to(cuda) for f in seeded_feats(17, [(24, 15, 256)] * 3)]
    target = torch.randint(0, 2, (24,), generator=torch.Generator().manual_seed(3)).to(cuda)

    def build():
        m = hhi_ttm.TaskFusionMFTransformer3Task(hhi_args(dropout=0.5))
        m.load_state_dict(seeded_state_dict(m, 21))
        return m.to(cuda).set_compute("f32s").train()

    m = build().enable_device_seed()
    step = GraphedStep(lambda f, y: crit(m.forward_features(*f), y), (feats, target), list(m.parameters()))
    losses = [step(feats, target).item() for _ in range(6)]
    torch.cuda.synchronize()
    assert len({round(v, 6) for v in losses}) >= 5, losses                 # fresh masks: (almost surely) six different losses
    assert all(abs(v) < 20 for v in losses)

    m2 = build()                                                           # host seed
    with pytest.raises(_lib.EgxError, match="cannot be captured in a hipGraph"):
        GraphedStep(lambda f, y: crit(m2.forward_features(*f), y), (feats, target), list(m2.parameters()))
    torch.cuda.synchronize()


def test_staged_backward_and_overlapped_allreduce_layout(egx_lib, cuda):
    """egx_defer_small: the backward stops before the grouped small weight gradients, run_deferred() finishes it, and the
    result equals the one-shot backward; the late gradients (dW_proj, dW_in, dW_o) sit first in the flat buffer so that
    ddp.allreduce_gradients_overlapped can exchange the rest while they are still being computed."""
    from egot2_amd import ddp, functional as F_egx
    a, b = _ttm(cuda, 8), _ttm(cuda, 8)
    feats = [f.to(cuda) for f in seeded_feats(31, [(6, 15, 256)] * 3)]
    target = torch.randint(0, 2, (6,), generator=torch.Generator().manual_seed(3)).to(cuda)
    w = torch.tensor(CE_W, device=cuda)
    torch.nn.functional.cross_entropy(a.forward_features(*feats), target, weight=w).backward()
    b.egx_defer_small = True
    torch.nn.functional.cross_entropy(b.forward_features(*feats), target, weight=w).backward()
    late_names = ("proj_lam.weight", "proj_ttm.weight", "proj_asd.weight", "transformer_encoder.layers.0.self_attn.in_proj_weight",
                  "transformer_encoder.layers.0.self_attn.out_proj.weight")
    nb = dict(b.named_parameters())
    lay = F_egx.last_grad_layout
    assert lay["late_floats"] == sum(nb[n].numel() for n in late_names)
    for n in late_names:                       # not computed yet: still the zero fill of stage 1
        assert nb[n].grad.abs().max().item() == 0.0
        assert nb[n].grad.untyped_storage().data_ptr() == lay["flat"].untyped_storage().data_ptr()
        assert nb[n].grad.storage_offset() < lay["late_floats"]
    assert nb["ln.weight"].grad.storage_offset() >= lay["late_floats"]
    calls = []
    n_coll = ddp.allreduce_gradients_overlapped(lambda: (calls.append(1), F_egx.run_deferred()))   # no process group: just finishes
    assert calls == [1] and n_coll == 0
    for (n, pa), (_, pb) in zip(a.named_parameters(), b.named_parameters()):
        assert torch.allclose(pa.grad, pb.grad, rtol=1e-4, atol=1e-6), n


def test_fused_adam_resume_continues_from_saved_moments(egx_lib, cuda):
    """state_dict -> fresh FusedAdam -> load_state_dict -> continue must equal the uninterrupted run (the
    moments and the bias-correction step used to restart from zero after a resume). Checked against torch.optim.Adam
    fed the same gradients, for a fresh optimizer and for one whose buckets already exist."""
    from egot2_amd.train import FusedAdam
    g = torch.Generator().manual_seed(21)
    shapes = [(33,), (64, 16), (5,)]
    mk = lambda: [torch.nn.Parameter(torch.randn(s, generator=torch.Generator().manual_seed(7 + i)).to(cuda)) for i, s in enumerate(shapes)]  # noqa: E731
    pa, pb = mk(), mk()
    grads = [[torch.randn(s, generator=g).to(cuda) for s in shapes] for _ in range(7)]

    def feed(ps, opt, k):
        flat = torch.cat([x.reshape(-1) for x in grads[k][:2]])       # first two gradients share one flat buffer
        ps[0].grad = flat[:33].view(33)
        ps[1].grad = flat[33:].view(64, 16)
        ps[2].grad = grads[k][2].clone()
        opt.step()

    ob = torch.optim.Adam(pb, lr=1e-2)
    oa = FusedAdam(pa, lr=1e-2)
    for k in range(3):
        feed(pa, oa, k)
        feed(pb, ob, k)
    sd = copy.deepcopy(oa.state_dict())
    assert int(sd["state"][0]["step"].item()) == 3
    weights = [p.detach().clone() for p in pa]
    # (1) fresh process: new parameters + new optimizer, nothing stepped yet
    pc = [torch.nn.Parameter(w.clone()) for w in weights]
    oc = FusedAdam(pc, lr=1e-2)
    oc.load_state_dict(sd)
    # (2) same process: an optimizer that already owns buckets, rolled back to the checkpoint
    for k in range(3, 5):
        feed(pa, oa, k)
    oa.load_state_dict(sd)
    with torch.no_grad():
        for p, w in zip(pa, weights):
            p.copy_(w)
    for k in range(3, 7):
        feed(pb, ob, k)
        feed(pc, oc, k)
        feed(pa, oa, k)
    for x, y, z in zip(pa, pb, pc):
        assert torch.allclose(z, y, rtol=2e-5, atol=2e-6), (z - y).abs().max().item()
        assert torch.allclose(x, y, rtol=2e-5, atol=2e-6), (x - y).abs().max().item()
    assert int(oc._step_dev.item()) == 7 and int(oc.state[pc[0]]["step"].item()) == 7


def test_weighted_ce_ignore_index_and_double_backward(egx_lib, cuda):
    """Labels outside [0, C) (F.cross_entropy's ignore_index = -100 included) carry no loss / weight / gradient, and a
    second backward through a retained graph returns the same gradient instead of None."""
    from egot2_amd import functional as F_egx
    g = torch.Generator().manual_seed(2)
    z = torch.randn(40, 3, generator=g).to(cuda).requires_grad_(True)
    y = torch.randint(0, 3, (40,), generator=g)
    y[::7] = -100
    y = y.to(cuda)
    w = torch.tensor([0.2, 0.5, 1.3], device=cuda)
    loss = F_egx.weighted_cross_entropy(z, y, w)
    want = torch.nn.functional.cross_entropy(z.detach().double().requires_grad_(True), y, weight=w.double())
    assert abs(loss.item() - want.item()) < 1e-5
    (g1,) = torch.autograd.grad(loss, z, retain_graph=True)
    (g2,) = torch.autograd.grad(loss, z)
    zr = z.detach().double().requires_grad_(True)
    torch.nn.functional.cross_entropy(zr, y, weight=w.double()).backward()
    assert torch.equal(g1, g2) and (g1.double() - zr.grad).abs().max().item() < 1e-6
    assert g1[::7].abs().max().item() == 0.0


def test_run_ttm_synth_plumbing_entry(egx_lib, cuda):
    """BASELINE.json configs[0] (the run_ttm.py plumbing entry): registry -> build_model(args) ->
    forward -> weighted CE -> FusedAdam on synthetic features, for both registered TTM translators; the loss goes down."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("run_ttm_synth", os.path.join(root, "tools", "run_ttm_synth.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    for name in ("TaskFusionMFTransformer2Task", "TaskFusionMFTransformer3Task"):
        losses = mod.main(["--model", name, "--num_layers", "1", "--hidden_dim", "128", "--dropout", "0.1", "--steps", "24",
                           "--lr", "2e-3", "--dtype", "f32s" if name.endswith("3Task") else "f32"])
        assert all(l == l for l in losses) and sum(losses[-4:]) < sum(losses[:4])


def _init_group(rank, world, port, backend):
    """gloo: every rank shares cuda:0 and the buffers travel through the host (pins layout and logic on a one-GPU box);
    nccl: one rank per device, RCCL over xGMI — the real thing, wherever >= 2 GPUs are visible."""
    import os
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    if backend == "nccl":
        torch.cuda.set_device(rank)
        dev = torch.device("cuda", rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dev = torch.device("cuda:0")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    return dev


def _hip_ddp_worker(rank, world, port, q, overlapped, backend="gloo", egx_comm=False):
    """One rank of the N-rank data-parallel step on the REAL HIP backward (see _init_group). egx_comm: the exchange through the
    C ABI's own RCCL communicator (egx_allreduce) instead of torch.distributed."""
    import torch.distributed as dist
    from egot2_amd import ddp, functional as F_egx
    dev = _init_group(rank, world, port, backend)
    m = _ttm(dev, 12)
    ddp.broadcast_parameters(m)
    feats = seeded_feats(41, [(8, 15, 256)] * 3)
    y = torch.tensor([0, 1, 1, 0, 1, 0, 0, 1])
    sf = [f.to(dev) for f in ddp.shard_batch(feats, rank, world)]
    sy = ddp.shard_batch([y], rank, world)[0].to(dev)
    m.egx_defer_small = overlapped
    loss = torch.nn.functional.cross_entropy(m.forward_features(*sf), sy)       # unweighted: rank means average exactly
    loss.backward()
    params = [p for p in m.parameters() if p.grad is not None]
    ranks_seen = None
    if egx_comm:
        def bcast(idb):
            box = [idb]
            dist.broadcast_object_list(box, src=0)
            return box[0]
        comm = ddp.EgxComm(rank, world, bcast)
        ranks_seen = comm.size
        n = comm.allreduce_gradients(params)
        torch.cuda.synchronize()
        comm.close()
    elif overlapped:
        n = ddp.allreduce_gradients_overlapped(F_egx.run_deferred, params)
    else:
        n = ddp.allreduce_gradients(params)
    torch.cuda.synchronize()
    one_storage = len({p.grad.untyped_storage().data_ptr() for p in params}) == 1
    if rank == 0:
        res = (n, one_storage, {k: p.grad.cpu().numpy().copy() for k, p in m.named_parameters()})
        q.put(res + (ranks_seen,) if egx_comm else res)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
@pytest.mark.parametrize("overlapped", [False, True])
def test_two_rank_hip_backward_allreduce_equals_single_process(egx_lib, cuda, overlapped, world):
    """the N-rank leg on the HIP backward itself (not the stock module): rank-sharded clips, ONE flat
    gradient buffer per rank straight out of the fused backward, one all-reduce (or the overlapped early / late pair),
    result == the single-process gradient on the concatenated batch."""
    import os
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 39500 + os.getpid() % 2000 + (7 if overlapped else 0) + 13 * world
    procs = [ctx.Process(target=_hip_ddp_worker, args=(r, world, port, q, overlapped)) for r in range(world)]
    for p in procs:
        p.start()
    n, one_storage, grads = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert one_storage and n == (2 if overlapped else 1)      # every gradient of the translator lives in one flat buffer
    m = _ttm(cuda, 12)
    feats = [f.to(cuda) for f in seeded_feats(41, [(8, 15, 256)] * 3)]
    y = torch.tensor([0, 1, 1, 0, 1, 0, 0, 1], device=cuda)
    torch.nn.functional.cross_entropy(m.forward_features(*feats), y).backward()
    for k, p in m.named_parameters():
        assert torch.allclose(p.grad.cpu(), torch.from_numpy(grads[k]), rtol=2e-3, atol=2e-6), k


def _lta_small(dev):
    from types import SimpleNamespace as NS
    from egot2_amd import hoi_lta
    cfg = NS(FORECASTING=NS(NUM_INPUT_CLIPS=4, NUM_ACTIONS_TO_PREDICT=3),
             MODEL=NS(TRANSLATION_HEADS=8, TRANSLATION_LAYERS=3, TRANSLATION_INPUT_FEATURES=256, TRANSLATION_DROPOUT=0.0,
                      NUM_CLASSES=[5, 7], DROPOUT_RATE=0.0, HEAD_ACT="softmax"), TEST=NS(NO_ACT=False))
    m = hoi_lta.TaskFusionMFTransformerLTA4Task(cfg)
    m.load_state_dict(seeded_state_dict(m, 77))
    return m.to(dev).set_compute("bf16").train()


def _lta_loss(m, feats):
    o = m.forward_features(*feats)
    return o[0].mean() + (o[1] * o[1]).mean()      # means over equal shards: the rank average is the full-batch gradient


_LTA_SHAPES = [(8, 4, 8192), (8, 4, 8192), (8, 4, 256), (8, 4, 2048)]


def _wide_bucket_worker(rank, world, port, q, backend="gloo"):
    """One rank of the N-rank step on the WIDE bf16 backward with the per-layer bucketed exchange (see _init_group)."""
    import torch.distributed as dist
    from egot2_amd import ddp, functional as F_egx
    dev = _init_group(rank, world, port, backend)
    m = _lta_small(dev)
    ddp.broadcast_parameters(m)
    sf = [f.to(dev) for f in ddp.shard_batch(seeded_feats(43, _LTA_SHAPES), rank, world)]
    params = [p for p in m.parameters() if p.requires_grad]
    seen = []
    with ddp.BucketedExchange(params) as ex:
        inner = F_egx.bucket_hook

        def spy(flat, lo, hi):          # record what the backward announces, then do the exchange
            seen.append((flat.data_ptr(), lo, hi))
            inner(flat, lo, hi)
        F_egx.bucket_hook = spy
        _lta_loss(m, sf).backward()
    torch.cuda.synchronize()
    enc = m.transformer.layers
    # layout: the slices were announced in address order (last layer first), and layer l's gradients sit inside slice L - 1 - l
    lay_ok = all(a[1] <= b[1] for a, b in zip(seen, seen[1:])) and len({s[0] for s in seen}) == 1
    base = seen[0][0]
    for l, layer in enumerate(enc):
        lo, hi = seen[len(enc) - 1 - l][1:]
        for p in layer.parameters():
            off = (p.grad.data_ptr() - base) // 4
            lay_ok = lay_ok and lo <= off and off + p.grad.numel() <= hi
    if rank == 0:
        q.put((ex.collectives, len(seen), lay_ok, {k: p.grad.float().cpu().numpy().copy() for k, p in m.named_parameters() if p.grad is not None}))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_two_rank_wide_bucketed_exchange_equals_single_process(egx_lib, cuda, world):
    """VERDICT r2 missing #1: gradient exchange overlapped with the backward on the WIDE path (configs[3] / [4]): the encoder's
    backward announces one slice of the flat buffer per layer, last layer first (egx_config.bucket_cb), every slice is
    all-reduced as it is announced, the head's torch gradients follow, and the result equals the single-process gradient on the
    concatenated batch."""
    import os
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 41500 + os.getpid() % 2000 + 13 * world
    procs = [ctx.Process(target=_wide_bucket_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    ncoll, nseen, lay_ok, grads = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert nseen == 4 and lay_ok            # three encoder layers + the token-preparation remainder, laid out last-layer-first
    assert nseen + 1 <= ncoll <= nseen + 3  # + the MultiTaskHead gradients torch computed (views of the stacked dW / db: one buffer each)
    m = _lta_small(cuda)
    _lta_loss(m, [f.to(cuda) for f in seeded_feats(43, _LTA_SHAPES)]).backward()
    for k, p in m.named_parameters():
        if p.grad is None:
            continue
        ref, got = p.grad.float().cpu(), torch.from_numpy(grads[k])
        assert (got - ref).norm().item() <= 2e-2 * ref.norm().item() + 1e-6, k      # bf16 operands: the batch split changes roundings


def _action_head(dev):
    """Stand-in for the trainable SlowFast head whose output is the LTA translator's (unprojected) action stream."""
    torch.manual_seed(5)
    return torch.nn.Linear(256, 256).to(dev)


def _wide_featgrad_worker(rank, world, port, q):
    """the action features of the LTA translators come from a TRAINABLE head, so the encoder's backward
    returns d(feature). That activation gradient must stay out of the exchanged flat buffer (it is per-sample, and autograd
    hands it upstream while the collectives run)."""
    import os
    import torch.distributed as dist
    from egot2_amd import ddp, functional as F_egx
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda:0")
    m, head = _lta_small(dev), _action_head(dev)
    ddp.broadcast_parameters(m)
    sf = [f.to(dev) for f in ddp.shard_batch(seeded_feats(43, _LTA_SHAPES), rank, world)]
    params = [p for p in m.parameters() if p.requires_grad] + list(head.parameters())
    seen = []
    act = head(sf[2])
    act.retain_grad()
    with ddp.BucketedExchange(params) as ex:
        inner = F_egx.bucket_hook

        def spy(flat, lo, hi):
            seen.append((flat.data_ptr() + 4 * lo, flat.data_ptr() + 4 * hi))
            inner(flat, lo, hi)
        F_egx.bucket_hook = spy
        _lta_loss(m, [sf[0], sf[1], act, sf[3]]).backward()
    torch.cuda.synchronize()
    a = act.grad.data_ptr()
    outside = all(not (lo <= a < hi) for lo, hi in seen)
    if rank == 0:
        q.put((outside, act.grad.float().cpu().numpy().copy(),
               {k: p.grad.float().cpu().numpy().copy() for k, p in list(m.named_parameters()) + [("acthead." + k, p) for k, p in head.named_parameters()]
                if p.grad is not None}))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_bucketed_exchange_keeps_feature_gradients_private(egx_lib, cuda):
    import os
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 43500 + os.getpid() % 2000
    procs = [ctx.Process(target=_wide_featgrad_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    outside, dact, grads = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert outside, "d(feature) lies inside a slice of the flat buffer that was all-reduced"
    m, head = _lta_small(cuda), _action_head(cuda)
    feats = [f.to(cuda) for f in seeded_feats(43, _LTA_SHAPES)]
    act = head(feats[2])
    act.retain_grad()
    _lta_loss(m, [feats[0], feats[1], act, feats[3]]).backward()
    # rank 0's d(feature) is the PER-SAMPLE gradient of its own shard (x world: its loss is a mean over half the batch), untouched by the exchange
    ref = 2.0 * act.grad[:4].float().cpu()
    got = torch.from_numpy(dact)
    assert (got - ref).norm().item() <= 3e-2 * ref.norm().item() + 1e-7
    named = dict(list(m.named_parameters()) + [("acthead." + k, p) for k, p in head.named_parameters()])
    for k, g in grads.items():
        ref, got = named[k].grad.float().cpu(), torch.from_numpy(g)
        assert (got - ref).norm().item() <= 3e-2 * ref.norm().item() + 1e-6, k


@pytest.mark.parametrize("compute,impl", [("f32", "fused"), ("bf16", "fused"), ("f32s", "fused"), ("f32", "generic"), ("bf16", "generic")])
def test_deterministic_mode_gives_bit_identical_gradients(egx_lib, cuda, compute, impl):
    """same inputs + same dropout seed => bit-identical logits and gradients across runs with
    `set_deterministic()`: the fused backward sums its split-K slabs and per-clip partial rows in a fixed order instead of
    fp32 atomics; the shape-generic backward routes its split-K weight gradients through slabs and its LayerNorm / bias /
    pooled-head parameter gradients through partial buffers with ordered sums. The result must still match the default
    (atomic) mode to accumulation-order noise. B = 300 clips: more than one round of workgroups, several token splits per
    weight-gradient problem."""
    B = 300
    feats = [f.to(cuda) for f in seeded_feats(55, [(B, 15, 256)] * 3)]
    target = (torch.arange(B, device=cuda) * 7) % 2
    w = torch.tensor(CE_W, device=cuda)

    def run(det):
        torch.manual_seed(4321)
        from egot2_amd import hhi_ttm
        m = hhi_ttm.TaskFusionMFTransformer3Task(hhi_args(num_layers=2, dropout=0.5))
        m.load_state_dict(seeded_state_dict(m, 9))
        m = m.to(cuda).set_compute(compute, impl).set_deterministic(det).train()
        m._egx_step = 0
        logits = m.forward_features(*feats)
        torch.nn.functional.cross_entropy(logits, target, weight=w).backward()
        torch.cuda.synchronize()
        return logits.detach().clone(), {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}

    a, b, c = run(True), run(True), run(False)
    assert torch.equal(a[0], b[0])
    for k in a[1]:
        assert torch.equal(a[1][k], b[1][k]), k
        assert torch.allclose(a[1][k], c[1][k], rtol=2e-3, atol=1e-6), k


@pytest.mark.parametrize("M,K,C,weighted", [(3840, 128, 2, True), (45, 128, 2, True), (1, 64, 2, False), (1000, 256, 5, True), (70000, 128, 2, True)])
def test_fused_linear_cross_entropy_matches_torch(egx_lib, cuda, M, K, C, weighted):
    """egx_linear_ce_*: Linear + weighted CE (+ softmax scores and the correct-frame count of lossAV) in one launch each way
    against torch's F.linear / F.cross_entropy autograd, including ignored labels and a non-unit upstream gradient."""
    from egot2_amd import functional as F_egx
    g = torch.Generator().manual_seed(M + K)
    x = torch.randn(M, K, generator=g).to(cuda).requires_grad_(True)
    W = (torch.randn(C, K, generator=g) * 0.2).to(cuda).requires_grad_(True)
    b = torch.randn(C, generator=g).to(cuda).requires_grad_(True)
    y = torch.randint(0, C, (M,), generator=g)
    if M > 10:
        y[3] = -100                                   # F.cross_entropy's default ignore_index
    y = y.to(cuda)
    w = torch.tensor([1.0, 4.0, 0.5, 2.0, 3.0][:C], device=cuda) if weighted else None
    loss, logits, probs, pred, correct = F_egx.linear_cross_entropy(x, W, b, y, w)
    (loss * 1.7).backward()
    xr, Wr, br = (t.detach().clone().requires_grad_(True) for t in (x, W, b))
    zr = torch.nn.functional.linear(xr, Wr, br)
    lr = torch.nn.functional.cross_entropy(zr, y, weight=w)
    (lr * 1.7).backward()
    torch.cuda.synchronize()
    assert torch.allclose(logits, zr, rtol=1e-5, atol=1e-5)
    assert abs(loss.item() - lr.item()) < 1e-5 * max(1.0, abs(lr.item()))
    assert torch.allclose(probs, torch.softmax(zr, dim=-1), rtol=1e-5, atol=1e-6)
    assert torch.equal(pred, torch.round(torch.softmax(zr, dim=-1))[:, 1])
    assert correct.item() == (pred == y).sum().item()
    assert torch.allclose(x.grad, xr.grad, rtol=1e-4, atol=1e-7)
    assert rel_err(W.grad, Wr.grad) < 1e-5 and rel_err(b.grad, br.grad) < 1e-5
    # fixed-order sums: a second run is bit-identical
    x2, W2, b2 = (t.detach().clone().requires_grad_(True) for t in (x, W, b))
    l2 = F_egx.linear_cross_entropy(x2, W2, b2, y, w)[0]
    (l2 * 1.7).backward()
    assert torch.equal(l2, loss) and torch.equal(W2.grad, W.grad) and torch.equal(b2.grad, b.grad)


def test_lossAV_mirror_matches_the_stock_modules(egx_lib, cuda):
    """hhi_asd.lossAV against nn.Linear + nn.CrossEntropyLoss(weight=[1, 4]) + softmax / round / count as written in
    HHI/tasks/asd/loss.py:11-30, with a shared state_dict (same parameter and buffer names)."""
    from egot2_amd import hhi_asd
    torch.manual_seed(3)
    ours = hhi_asd.lossAV(128).to(cuda)
    fc = torch.nn.Linear(128, 2).to(cuda)
    crit = torch.nn.CrossEntropyLoss(weight=torch.FloatTensor([1, 4])).to(cuda)
    assert set(ours.state_dict()) == {"criterion.weight", "FC.weight", "FC.bias"}
    fc.load_state_dict({"weight": ours.FC.weight.detach(), "bias": ours.FC.bias.detach()})
    x = torch.randn(705, 1, 128, device=cuda)
    y = torch.randint(0, 2, (705,), device=cuda)
    xo, xs = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    nloss, score, label, num = ours(xo, y)
    nloss.backward()
    z = fc(xs.squeeze(1))
    ref = crit(z, y)
    ref.backward()
    assert abs(nloss.item() - ref.item()) < 1e-5
    assert torch.allclose(score, torch.softmax(z, -1), atol=1e-6)
    assert torch.equal(label, torch.round(torch.softmax(z, -1))[:, 1]) and num.item() == (label == y).sum().item()
    assert torch.allclose(xo.grad, xs.grad, rtol=1e-4, atol=1e-8)
    assert rel_err(ours.FC.weight.grad, fc.weight.grad) < 1e-5 and rel_err(ours.FC.bias.grad, fc.bias.grad) < 1e-5
    import numpy as np
    assert np.allclose(ours(x), z[:, 1].detach().cpu().numpy(), atol=1e-5)


@pytest.mark.parametrize("extra", [["--force-dist"], ["--force-dist", "--graph-collectives"], ["--force-dist", "--no-overlap"],
                                   ["--config", "c5hhi", "--batch", "32", "--force-dist"],
                                   # round 4: the bucketed exchange CAPTURED with the step (device-resident dropout seed on the wide path)
                                   ["--config", "c5hhi", "--batch", "32", "--force-dist", "--graph-collectives"]])
def test_bench_distributed_code_paths_on_one_rank(cuda, extra):
    """The RCCL code paths of bench.py (staged-backward overlap, collectives captured inside the step's hipGraph, single
    collective, per-layer buckets on the wide path) with a one-rank process group: the JSON line must carry the exchange fields.
    An N > 1 run needs a multi-GPU node; this pins that the paths start, capture and time without one."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["MASTER_PORT"] = str(43500 + os.getpid() % 2000 + len(extra))
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--steps", "3", "--warmup", "1", "--trials", "1", "--no-cpu-baseline", "--no-roofline",
           "--no-optimizer-line", "--no-native-line"] + extra
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["value"] > 0 and out["allreduce_us"] > 0
    if "c5hhi" in extra:
        assert out["overlap"] == "bucketed" and out["collectives_per_step"] >= 4      # 3 encoder layers + remainder + the decoder's buffer
        assert out["config"]["launch"] == ("one hipGraph replay per step" if "--graph-collectives" in extra else "eager")
    elif "--no-overlap" in extra:
        assert out["overlap"] == "none"
    else:
        assert out["overlap"] == "staged"


# ---- RCCL with peers: one rank per visible device. Skipped on the one-GPU boxes of this pool; the moment a node
# with >= 2 GPUs runs the suite these are the first N > 1 RCCL runs of the library.
def _rccl_world():
    n = torch.cuda.device_count()
    return 8 if n >= 8 else 4 if n >= 4 else 2 if n >= 2 else 0


needs_peers = pytest.mark.skipif(_rccl_world() < 2, reason="needs >= 2 GPUs: one RCCL rank per device")


def _spawn(worker, world, port, *args):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=worker, args=(r, world, port, q) + args) for r in range(world)]
    for p in procs:
        p.start()
    res = q.get(timeout=600)
    for p in procs:
        p.join(timeout=180)
        assert p.exitcode == 0
    return res


def _ttm_single_process_grads(cuda):
    m = _ttm(cuda, 12)
    feats = [f.to(cuda) for f in seeded_feats(41, [(8, 15, 256)] * 3)]
    y = torch.tensor([0, 1, 1, 0, 1, 0, 0, 1], device=cuda)
    torch.nn.functional.cross_entropy(m.forward_features(*feats), y).backward()
    return {k: p.grad.cpu() for k, p in m.named_parameters()}


@needs_peers
@pytest.mark.parametrize("mode", ["single", "staged", "egx_comm"])
def test_rccl_one_rank_per_device_fused_path_equals_single_process(egx_lib, cuda, mode):
    """N ranks over RCCL / xGMI on the per-clip kernels: the clips sharded by rank, the flat gradient buffer exchanged by one
    all-reduce, by the staged pair overlapped with the backward tail (bench.py's default for N > 1), or through the C ABI's own
    communicator (egx_allreduce); the result is the single-process gradient of the whole batch."""
    import os
    world = _rccl_world()
    port = 45500 + os.getpid() % 2000 + {"single": 0, "staged": 3, "egx_comm": 5}[mode]
    res = _spawn(_hip_ddp_worker, world, port, mode == "staged", "nccl", mode == "egx_comm")
    assert res[1] and res[0] == (2 if mode == "staged" else 1)
    if mode == "egx_comm":
        assert res[3] == world          # what RCCL itself reports for the communicator
    ref = _ttm_single_process_grads(cuda)
    for k, g in ref.items():
        assert torch.allclose(g, torch.from_numpy(res[2][k]), rtol=2e-3, atol=2e-6), k


@needs_peers
def test_rccl_one_rank_per_device_wide_bucketed_exchange_equals_single_process(egx_lib, cuda):
    """N ranks over RCCL on the wide bf16 path: per-layer buckets all-reduced while the backward runs (the configurations meant
    for 8 GPUs, BASELINE.json configs[3] / [4])."""
    import os
    world = _rccl_world()
    ncoll, nseen, lay_ok, grads = _spawn(_wide_bucket_worker, world, 47500 + os.getpid() % 2000, "nccl")
    assert nseen == 4 and lay_ok and nseen + 1 <= ncoll <= nseen + 3
    m = _lta_small(cuda)
    _lta_loss(m, [f.to(cuda) for f in seeded_feats(43, _LTA_SHAPES)]).backward()
    for k, p in m.named_parameters():
        if p.grad is not None:
            ref, got = p.grad.float().cpu(), torch.from_numpy(grads[k])
            assert (got - ref).norm().item() <= 2e-2 * ref.norm().item() + 1e-6, k


def test_egx_allreduce_one_rank_communicator(egx_lib, cuda):
    """The C ABI's RCCL entry points on one GPU: a one-rank communicator (ncclCommInitRank with nranks = 1), the exchange of a real
    backward's flat gradient buffer through egx_allreduce (average over one rank = identity), the library RCCL was resolved from."""
    from egot2_amd import ddp
    assert egx_lib.egx_comm_library(), "RCCL was not resolved"
    comm = ddp.EgxComm(0, 1)
    assert comm.size == 1
    t = torch.arange(1000, device=cuda, dtype=torch.float32)
    comm.allreduce_(t)
    comm.allreduce_(t, average=False)
    tb = torch.ones(257, device=cuda, dtype=torch.bfloat16)
    comm.allreduce_(tb)
    torch.cuda.synchronize()
    assert torch.equal(t.cpu(), torch.arange(1000, dtype=torch.float32)) and torch.equal(tb.cpu(), torch.ones(257, dtype=torch.bfloat16))
    m = _ttm(cuda, 12)
    feats = [f.to(cuda) for f in seeded_feats(41, [(8, 15, 256)] * 3)]
    y = torch.tensor([0, 1, 1, 0, 1, 0, 0, 1], device=cuda)
    torch.nn.functional.cross_entropy(m.forward_features(*feats), y).backward()
    before = {k: p.grad.clone() for k, p in m.named_parameters()}
    assert comm.allreduce_gradients(m.parameters()) == 1        # every gradient of the translator lives in one flat buffer
    torch.cuda.synchronize()
    assert all(torch.equal(p.grad, before[k]) for k, p in m.named_parameters())
    comm.close()


def test_captured_step_with_collectives_on_the_c_abi_communicator_replays_200_times(egx_lib, cuda):
    """a step WITH its gradient exchange captured as one hipGraph, the collectives on egx_allreduce (ddp.use_egx_comm: RCCL
    through the C ABI on streams this process owns — no ProcessGroupNCCL work objects, no watchdog thread) instead of torch.distributed, whose
    watchdog aborted one captured run in six in round 5 (hipErrorCapturedEvent). One-rank communicator (this pool hands out one GPU): the staged
    exchange of the per-clip path (first collective on the side stream as a parallel branch of the graph, under the grouped small weight
    gradients; second on the capture stream) and the single exchange, each replayed 200 times; the averaged-over-one-rank gradients must equal
    the un-exchanged backward's and stay finite throughout."""
    from egot2_amd import ddp, functional as F_egx
    from egot2_amd.train import CrossEntropyLoss
    crit = CrossEntropyLoss(torch.FloatTensor(CE_W)).to(cuda)
    comm = ddp.EgxComm(0, 1)
    ddp.use_egx_comm(comm)
    try:
        for staged in (True, False):
            from egot2_amd import hhi_ttm
            m = hhi_ttm.TaskFusionMFTransformer3Task(hhi_args(dropout=0.3))
            m.load_state_dict(seeded_state_dict(m, 5))
            m = m.to(cuda).set_compute("f32s").train().enable_device_seed()
            m.egx_defer_small = staged
            params = [p for p in m.parameters() if p.requires_grad]
            feats = [f.to(cuda) for f in seeded_feats(77, [(40, 15, 256)] * 3)]
            y = torch.randint(0, 2, (40,), generator=torch.Generator().manual_seed(7)).to(cuda)
            one = F_egx.unit_grad(cuda)
            counts = []

            def step():
                for p in params:
                    p.grad = None
                loss = crit(m.forward_features(*feats), y)
                loss.backward(gradient=one)
                if staged:
                    counts.append(ddp.allreduce_gradients_overlapped(F_egx.run_deferred, params, force=True))
                else:
                    counts.append(ddp.allreduce_gradients(params, force=True))
                return loss

            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(3):
                    step()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            assert counts[-1] >= (2 if staged else 1)
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, capture_error_mode="thread_local"):
                loss = step()
            seen = set()
            for i in range(200):
                gr.replay()
                if i % 50 == 49:
                    torch.cuda.synchronize()
                    assert all(torch.isfinite(p.grad).all().item() for p in params)
                    seen.add(round(loss.item(), 6))
            torch.cuda.synchronize()
            assert len(seen) >= 3, seen         # fresh dropout masks replay after replay: the graph is doing real steps
            # p = 0 in eval-equivalent terms is not available here (dropout recipe); compare one replay with the un-exchanged eager backward from the same seed
            m._egx_seed_dev.fill_(1234)
            gr.replay()
            torch.cuda.synchronize()
            got = {k: p.grad.clone() for k, p in m.named_parameters()}
            m._egx_seed_dev.fill_(1234)
            m.egx_defer_small = False
            ddp.use_egx_comm(None)
            for p in params:
                p.grad = None
            crit(m.forward_features(*feats), y).backward()
            torch.cuda.synchronize()
            ddp.use_egx_comm(comm)
            for k, p in m.named_parameters():
                # (+ a floor for gradients that cancel to ~0, the head bias: their relative error under another atomic-add order is noise)
                assert (got[k] - p.grad).norm().item() <= 2e-5 * p.grad.norm().item() + 2e-7 * p.grad.numel() ** 0.5, k
    finally:
        ddp.use_egx_comm(None)
        comm.close()
