"""CPU suite for the host side: C-ABI surface, registry semantics, error behaviour, data-parallel gradient exchange
(gloo, world_size 2), bench arithmetic."""
import os

import numpy as np
import re
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol(egx_lib):
    """Every function include/egot2x.h declares is exported by libegot2x.so and bound in egot2_amd/_lib.py."""
    from egot2_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "egot2x.h")).read()
    declared = set(re.findall(r"\b(egx_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"egx_config", "egx_segment", "egx_layer"}
    assert declared, "no declarations parsed"
    for name in sorted(declared):
        assert hasattr(egx_lib, name), f"{name} declared in egot2x.h but not exported"
    assert declared <= set(_lib.SIGNATURES), f"unbound symbols: {declared - set(_lib.SIGNATURES)}"
    assert egx_lib.egx_abi_version() == _lib.EGX_ABI_VERSION


def test_workspace_query_and_config_errors(egx_lib):
    """Host-only entry points: workspace sizing and argument validation (no GPU needed)."""
    import ctypes as C
    from egot2_amd._lib import Config, Segment
    cfg = Config(128, 4, 2048, 1, 3, 1e-5, 0, 0, 0.0, 0.0, 0.0)
    segs = (Segment * 3)()
    for s in segs:
        s.T, s.d_in, s.proj_w = 15, 256, 1   # non-null marker
    sv, sc = C.c_size_t(), C.c_size_t()
    assert egx_lib.egx_encoder_workspace(C.byref(cfg), segs, 256, C.byref(sv), C.byref(sc)) == 0
    assert sv.value > 256 * 45 * 128 * 4 and sc.value > 0
    bad = Config(132, 5, 2048, 1, 3, 1e-5, 0, 0, 0.0, 0.0, 0.0)   # d_model not divisible by heads
    assert egx_lib.egx_encoder_workspace(C.byref(bad), segs, 256, C.byref(sv), C.byref(sc)) != 0
    assert b"n_heads" in egx_lib.egx_last_error()
    assert egx_lib.egx_encoder_workspace(C.byref(cfg), segs, 0, C.byref(sv), C.byref(sc)) != 0   # empty batch


def test_workspace_sizes_above_2_gib_are_not_truncated(egx_lib):
    """C4 at B=256 (HOI LTA 4-task, S=128, d=768, 4 layers) keeps > 4 GiB of intermediates: the size must come back
    intact (an unqualified max() in hipcc host code resolves to max(int, int) and used to return 0 here)."""
    import ctypes as C
    from egot2_amd._lib import Config, Segment
    cfg = Config(768, 8, 2048, 4, 4, 1e-5, 0, 0, 0.1, 0.0, 0.0)
    segs = (Segment * 4)()
    for s, k in zip(segs, (8192, 8192, 768, 2048)):
        s.T, s.d_in, s.proj_w = 32, k, (0 if k == 768 else 1)
    sv, sc = C.c_size_t(), C.c_size_t()
    assert egx_lib.egx_encoder_workspace(C.byref(cfg), segs, 256, C.byref(sv), C.byref(sc)) == 0
    N = 256 * 128
    assert sv.value > 4 * N * (2048 + 7 * 768) * 4 > 2**31      # per layer: hidden + x_in, qkv (3), attn_o, res1, x1 (+ res2)
    assert sc.value > N * 2048 * 4
    sv2, sc2 = C.c_size_t(), C.c_size_t()
    assert egx_lib.egx_translator_workspace(C.byref(cfg), segs, 256, C.byref(sv2), C.byref(sc2)) == 0
    assert sv2.value >= sv.value and sc2.value >= sc.value


def test_cpu_tensors_raise_not_fallback(egx_lib):
    from egot2_amd import _lib, hhi_ttm
    from tests.util import hhi_args, seeded_feats
    model = hhi_ttm.TaskFusionMFTransformer3Task(hhi_args())
    with pytest.raises(_lib.EgxError, match="no CPU fallback"):
        model.forward_features(*seeded_feats(1, [(2, 15, 256)] * 3))


def test_registry_and_constructor_protocol():
    """HHI/models/ttm/build.py:17-20 semantics and the reference's backbone/freeze quirk."""
    from argparse import Namespace
    from egot2_amd import hhi_asd, hhi_ttm
    from tests.util import hhi_args
    a = hhi_args()
    a.model = "TaskFusionMFTransformer3Task"
    assert type(hhi_ttm.build_model(a)).__name__ == "TaskFusionMFTransformer3Task"
    assert type(hhi_asd.build_model(a)) is hhi_asd.TaskFusionMFTransformer3Task
    with pytest.raises(KeyError):
        hhi_ttm.MODEL_REGISTRY.get("NoSuchModel")
    frozen = Namespace(**{**vars(hhi_args()), "nofreeze": False})
    with pytest.raises(AttributeError):   # the reference freezes self.ttm_model unconditionally
        hhi_ttm.TaskFusionMFTransformer3Task(frozen)
    with pytest.raises(ImportError):      # a checkpoint without the reference tree / a registered factory
        hhi_ttm.TaskFusionMFTransformer3Task(Namespace(**{**vars(hhi_args()), "lam_checkpoint": "x.pth"}))


def test_bench_flop_model_matches_baseline_md():
    sys.path.insert(0, ROOT)
    import bench
    fwd, bwd = bench.algorithmic_flops(256, 3, 15, 256, 128, 4, 1, 2048)
    assert abs(fwd / 1e9 - 14.610) < 0.01 and abs((fwd + bwd) / 1e9 - 43.075) < 0.02


def _ddp_worker(rank, world, port, q):
    import torch.distributed as dist
    from egot2_amd import ddp
    from oracle.stock_module import StockTTMTranslator
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(1 + rank)            # different init per rank: broadcast must fix it
    m = StockTTMTranslator(3, 128, 4, dropout=0.0, num_layers=1).train()
    m.pos_embed.dropout.p = 0.0
    ddp.broadcast_parameters(m)
    g = torch.Generator().manual_seed(5)
    feats = [torch.randn(8, 15, 256, generator=g) for _ in range(3)]
    y = torch.tensor([0, 1, 1, 0, 1, 0, 0, 1])
    sf = ddp.shard_batch(feats, rank, world)
    sy = ddp.shard_batch([y], rank, world)[0]
    loss = torch.nn.functional.cross_entropy(m(*sf), sy)      # unweighted: per-rank means average exactly
    loss.backward()
    # make most gradients views of one flat buffer, as the HIP encoder's backward does
    params = [p for p in m.parameters() if p.grad is not None]
    flat = torch.cat([p.grad.reshape(-1) for p in params[:-2]])
    off = 0
    for p in params[:-2]:
        n = p.numel()
        p.grad = flat[off:off + n].view_as(p)
        off += n
    ncoll = ddp.allreduce_gradients(params)
    if rank == 0:
        # numpy payloads: torch tensors travel by fd and would need this process alive at receive time
        q.put((ncoll, {k: p.grad.numpy().copy() for k, p in m.named_parameters()},
               {k: v.numpy().copy() for k, v in m.state_dict().items()}))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gradient_allreduce_equals_single_process():
    """world_size 2 over gloo: sharded batch + one all-reduce of the flat gradient buffer == the single-process
    gradient on the concatenated batch."""
    import torch.multiprocessing as mp
    from oracle.stock_module import StockTTMTranslator
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_ddp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    ncoll, grads, sd = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert ncoll == 2          # one flat buffer + one coalesced buffer for the stragglers
    m = StockTTMTranslator(3, 128, 4, dropout=0.0, num_layers=1).train()
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    m.pos_embed.dropout.p = 0.0
    g = torch.Generator().manual_seed(5)
    feats = [torch.randn(8, 15, 256, generator=g) for _ in range(3)]
    y = torch.tensor([0, 1, 1, 0, 1, 0, 0, 1])
    torch.nn.functional.cross_entropy(m(*feats), y).backward()
    for k, p in m.named_parameters():
        assert torch.allclose(p.grad, torch.from_numpy(grads[k]), rtol=1e-4, atol=1e-6), k


def _overlap_worker(rank, world, port, q):
    import torch.distributed as dist
    from egot2_amd import ddp, functional as F_egx
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    flat = torch.arange(40, dtype=torch.float32) * (rank + 1)        # rank 0: i, rank 1: 2i -> average 1.5 i
    late = 12
    flat[:late] = -1.0                                              # "not computed yet": finish_backward fills them in
    F_egx.last_grad_layout.clear()
    F_egx.last_grad_layout.update(flat=flat, late_floats=late)
    order = []

    def finish():
        order.append("finish")
        flat[:late] = torch.arange(late, dtype=torch.float32) * 10 * (rank + 1)
    n = ddp.allreduce_gradients_overlapped(finish)
    if rank == 0:
        q.put((n, order, flat.numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_overlapped_allreduce():
    """The overlapped exchange on gloo: the early region is reduced while `finish_backward` produces the late one, the
    late region follows, and both come out averaged over the ranks."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_overlap_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    n, order, flat = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert n == 2 and order == ["finish"]
    i = np.arange(40, dtype=np.float32)
    assert np.allclose(flat[12:], 1.5 * i[12:])
    assert np.allclose(flat[:12], 15.0 * i[:12])


def test_bench_self_launches_ranks_gloo():
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment must start 2 ranks by itself (the round-1 script
    silently ran one). The launcher plumbing is exercised on CPU through --launch-selftest --backend gloo: both ranks
    join the process group (rank_sum = 1 + 2), the timing is the max over ranks, `n_gpus` and `parallelism` say 2, and
    the JSON line is the last line of stdout."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    port = 33500 + os.getpid() % 2000
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-selftest", "--backend", "gloo",
                        "--master-port", str(port)], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = r.stdout.strip().splitlines()[-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["config"]["parallelism"] == "dp2" and out["rank_sum"] == 3.0
    assert out["max_rank_seconds"] >= 0.02          # rank 1 sleeps 20 ms: the max over ranks, not rank 0's 10 ms


def test_bench_rejects_mismatched_world_size():
    """--gpus N under a launcher that set a different WORLD_SIZE is an error, not a silently relabelled run."""
    import subprocess
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-selftest"],
                       capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE" in r.stderr


def _ddp_weighted_worker(rank, world, port, q):
    import torch.distributed as dist
    from egot2_amd import ddp
    from oracle.stock_module import StockTTMTranslator
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(3)
    m = StockTTMTranslator(3, 128, 4, dropout=0.0, num_layers=1).train()
    m.pos_embed.dropout.p = 0.0
    g = torch.Generator().manual_seed(6)
    feats = [torch.randn(8, 15, 256, generator=g) for _ in range(3)]
    y = torch.tensor([0, 0, 0, 1, 1, 1, 1, 0])             # rank 0 holds classes {0,0,0,1}, rank 1 {1,1,1,0}: unequal weight sums
    w = torch.tensor([0.266, 0.734])
    sf = ddp.shard_batch(feats, rank, world)
    sy = ddp.shard_batch([y], rank, world)[0]
    loss = torch.nn.functional.cross_entropy(m(*sf), sy, weight=w)       # bench.py's per-rank weighted loss
    loss.backward()
    ddp.allreduce_gradients(list(m.parameters()))
    wsum = w[sy].sum().item()
    if rank == 0:
        q.put(({k: p.grad.numpy().copy() for k, p in m.named_parameters()},
               {k: v.numpy().copy() for k, v in m.state_dict().items()}, wsum))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_weighted_ce_normalisation_effect():
    """SURVEY.md §8e caveat, pinned: CrossEntropyLoss(weight) divides by the sum of the target weights of the LOCAL batch,
    so averaging per-rank gradients (what DDP and bench.py do, as the reference silently does) equals the single-process
    gradient of  0.5 * (L_0 + L_1)  with each L_r normalised by its own rank's weight sum - NOT the gradient of the
    weighted CE over the concatenated batch unless both ranks hold the same class mix."""
    import torch.multiprocessing as mp
    from oracle.stock_module import StockTTMTranslator
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 35500 + os.getpid() % 2000
    procs = [ctx.Process(target=_ddp_weighted_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    grads, sd, wsum0 = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    m = StockTTMTranslator(3, 128, 4, dropout=0.0, num_layers=1).train()
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    m.pos_embed.dropout.p = 0.0
    g = torch.Generator().manual_seed(6)
    feats = [torch.randn(8, 15, 256, generator=g) for _ in range(3)]
    y = torch.tensor([0, 0, 0, 1, 1, 1, 1, 0])
    w = torch.tensor([0.266, 0.734])
    ce = torch.nn.functional.cross_entropy
    # (a) what the exchange computes: the mean of the two per-rank weighted losses
    out = m(*feats)
    (0.5 * (ce(out[:4], y[:4], weight=w) + ce(out[4:], y[4:], weight=w))).backward()
    for k, p in m.named_parameters():
        assert torch.allclose(p.grad, torch.from_numpy(grads[k]), rtol=1e-4, atol=1e-6), k
    # (b) the concatenated-batch weighted CE differs, by exactly the per-rank weight-sum ratio
    assert abs(wsum0 - (3 * 0.266 + 0.734)) < 1e-6
    m.zero_grad()
    ce(m(*feats), y, weight=w).backward()
    gk = "linear_head.1.bias"
    assert not torch.allclose(m.get_parameter(gk).grad, torch.from_numpy(grads[gk]), rtol=1e-3, atol=1e-7)


def test_overlapped_allreduce_also_exchanges_gradients_outside_the_flat_buffer():
    """allreduce_gradients_overlapped used to reduce only the recorded flat buffer. With `params` it must
    also exchange gradients living elsewhere (heads, decoder) and refuse a flat buffer that belongs to other params."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 37500 + os.getpid() % 2000
    procs = [ctx.Process(target=_overlap_params_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    n, a, b, raised = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert n == 3 and raised           # early + late regions of the flat buffer, then one coalesced loose buffer
    assert np.allclose(a, 1.5 * np.arange(8)) and np.allclose(b, 1.5 * np.ones(5))


def _overlap_params_worker(rank, world, port, q):
    import torch.distributed as dist
    from egot2_amd import ddp, functional as F_egx
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    flat = torch.arange(8, dtype=torch.float32) * (rank + 1)
    pa = torch.nn.Parameter(torch.zeros(8))
    pa.grad = flat[0:8].view(8)
    pb = torch.nn.Parameter(torch.zeros(5))
    pb.grad = torch.ones(5) * (rank + 1)                # lives outside the flat buffer (a head / decoder parameter)
    F_egx.last_grad_layout.clear()
    F_egx.last_grad_layout.update(flat=flat, late_floats=4)
    n = ddp.allreduce_gradients_overlapped(lambda: None, [pa, pb])
    raised = False
    F_egx.last_grad_layout.update(flat=torch.zeros(4), late_floats=0)      # a buffer none of the params' grads live in
    try:
        ddp.allreduce_gradients_overlapped(lambda: None, [pa, pb])
    except RuntimeError:
        raised = True
    if rank == 0:
        q.put((n, pa.grad.numpy().copy(), pb.grad.numpy().copy(), raised))
    dist.barrier()
    dist.destroy_process_group()
