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


def test_host_entry_points_sweep(egx_lib):
    """tests/host_paths.py on the product library: every implementation's workspace layout over shapes, compute modes and batches,
    the implementation / slice policy, the error paths, the decoder layouts, the RCCL binding's resolver."""
    from egot2_amd import _lib
    from tests import host_paths
    assert host_paths.exercise(host_paths.bind(_lib.LIB_PATH)) > 10000


def test_host_code_under_address_and_ub_sanitizers():
    """SURVEY.md §5 / VERDICT r4 item 9d: the same sweep against a build whose C++ orchestration (encoder.hip, wide_host.hip,
    wide_decoder.hip, comm.hip) is compiled with AddressSanitizer + UBSan on the HOST side (egot2_amd/build.py build_sanitized;
    -fno-gpu-sanitize: no GPU ASAN, no XNACK), in a subprocess with the ASAN runtime preloaded and without torch. Any report —
    an out-of-bounds read of a layout table, a signed overflow in a size computation, a misaligned access — aborts the child."""
    import subprocess
    import sys
    from egot2_amd import build as egx_build
    lib = egx_build.build_sanitized()
    syms = subprocess.run(["nm", "-D", lib], capture_output=True, text=True).stdout
    assert "__asan_init" in syms and "__ubsan_handle" in syms, "the sanitized build carries no sanitizer instrumentation"
    env = dict(os.environ, LD_PRELOAD=egx_build.asan_runtime(), ASAN_OPTIONS="detect_leaks=0:halt_on_error=1:abort_on_error=0",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1", PYTHONPATH=ROOT)
    env.pop("EGX_LIB", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "host_paths.py"), lib], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0 and "host paths ok" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])
    assert "AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-4000:]


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


def test_registry_and_constructor_protocol(monkeypatch):
    """HHI/models/ttm/build.py:17-20 semantics and the reference's backbone/freeze quirk."""
    # order-independent: other test modules put the reference HHI tree on sys.path (then `models.lam.model` resolves to
    # the real LAMBackbone and the checkpoint open fails instead); a None entry makes the import fail as it does on a
    # box without the reference tree.
    for name in ("models", "models.lam", "models.lam.model", "models.ttm", "models.ttm.model"):
        monkeypatch.setitem(sys.modules, name, None)
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


def _bucket_worker(rank, world, port, q):
    import torch.distributed as dist
    from egot2_amd import ddp, functional as F_egx
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # a "backward" that announces its flat buffer in three slices (two layers, last layer first, then the rest), one
    # parameter whose gradient lives outside any announced buffer (a task head computed by torch)
    flat = torch.zeros(48)
    ps = [torch.nn.Parameter(torch.zeros(16)) for _ in range(3)] + [torch.nn.Parameter(torch.zeros(5))]
    # precondition: a parameter that already holds a .grad would be accumulated into while the collective rewrites it
    ps[0].grad = torch.zeros(16)
    try:
        with ddp.BucketedExchange(ps):
            raise AssertionError("a stale .grad must be refused")
    except RuntimeError as e:
        assert "already hold a .grad" in str(e) and F_egx.bucket_hook is None
    ps[0].grad = None
    order = []
    with ddp.BucketedExchange(ps) as ex:
        assert F_egx.bucket_hook is not None
        for i in range(3):                                   # the "backward" creates the gradients as views of its flat buffer
            ps[i].grad = flat[16 * i:16 * (i + 1)]
        for i in range(3):                                   # slice i becomes final, is announced, then the next one is "computed"
            flat[16 * i:16 * (i + 1)] = (i + 1) * (rank + 1) * torch.arange(16, dtype=torch.float32)
            F_egx.bucket_hook(flat, 16 * i, 16 * (i + 1))
            order.append(ex.collectives)
        ps[3].grad = torch.full((5,), float(rank + 1))
    assert F_egx.bucket_hook is None
    if rank == 0:
        q.put((ex.collectives, order, flat.numpy().copy(), ps[3].grad.numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_bucketed_exchange():
    """ddp.BucketedExchange on gloo: every announced slice is all-reduced (averaged) as it is announced, the gradient that no
    backward announced is exchanged when the block is left, and the hook is removed again."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 35500 + os.getpid() % 2000
    procs = [ctx.Process(target=_bucket_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    n, order, flat, loose = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert order == [1, 2, 3] and n == 4            # three slices + one collective for the unannounced gradient
    i = np.arange(16, dtype=np.float32)
    for k in range(3):
        assert np.allclose(flat[16 * k:16 * (k + 1)], 1.5 * (k + 1) * i)
    assert np.allclose(loose, 1.5)


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


def test_bench_trusts_the_launchers_world_size():
    """Under a launcher (WORLD_SIZE set) the environment decides the world size: a harness that runs
    `torchrun --nproc-per-node=N bench.py` without repeating --gpus N must still get its JSON line, labelled with the
    ranks that actually ran; the mismatch is reported on stderr."""
    import json
    import subprocess
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-selftest"],
                       capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode == 0 and "WORLD_SIZE" in r.stderr
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["n_gpus"] == 1 and out["config"]["parallelism"] == "dp1"


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


class _StubBackbone(torch.nn.Module):
    """Factory-built stand-in for a frozen HOI backbone: one parameter (to check freezing), records how it was built."""

    def __init__(self, kind, **kw):
        super().__init__()
        self.kind, self.kw = kind, kw
        self.w = torch.nn.Parameter(torch.ones(1))
        self.head = torch.nn.Linear(2, 2)                 # "head" parameters stay trainable under freeze_backbone_params

    def forward(self, x, *a, middle=False, **k):
        B = x[0].shape[0]
        if self.kind in ("pnr", "oscc"):
            return torch.zeros(B, 16, 8192)
        if self.kind == "lta":
            return torch.zeros(x[0].shape[1], B, 2048)    # (n, B, 2048), transposed by the caller
        return torch.zeros(B, self.kw["num_classes"][0])  # SlowFast with head: (B, d)


def _with_hoi_factories(fn):
    from egot2_amd import backbones
    built = []
    keys = ("hoi_pnr", "hoi_oscc", "hoi_slowfast", "hoi_lta")
    for key in keys:
        backbones.register_backbone_factory(key, lambda _k=key[4:], **kw: built.append(_k) or _StubBackbone(_k, **kw))
    try:
        return fn(built)
    finally:
        for key in keys:
            backbones._FACTORIES.pop(key, None)


def test_hoi_constructors_build_their_backbones():
    """`Class(cfg)` of the HOI translators must build and freeze its backbones where the reference does
    (lta_models_lta_transfer.py:279-302, video_model_transfer_3task.py:23-58, video_model_builder.py:98-130), so that the
    real forward signature works on a stock instance. Backbones come from injected factories; forward() must get as far
    as the HIP call (which refuses CPU tensors) instead of dying on a missing attribute."""
    from types import SimpleNamespace as NS
    from egot2_amd import _lib, hoi_ar, hoi_lta, hoi_multitask, hoi_pnr

    def run(built):
        cfg = NS(FORECASTING=NS(NUM_INPUT_CLIPS=2, NUM_ACTIONS_TO_PREDICT=3, INPUT_OFFSET=0),
                 MODEL=NS(TRANSLATION_HEADS=8, TRANSLATION_LAYERS=1, TRANSLATION_INPUT_FEATURES=256, TRANSLATION_DROPOUT=0.0,
                          NUM_CLASSES=[5, 7], DROPOUT_RATE=0.0, HEAD_ACT="softmax", FEAT_DROPOUT_RATE=0.0, TRANSFORMER_DROPOUT_RATE=0.0),
                 TEST=NS(NO_ACT=False), DATA=NS(TASK="state_change_detection"),
                 PRETRAIN=NS(PNR_CFG="pnr.yaml", OSCC_CFG="oscc.yaml", ACTION_CFG="ar.yaml", LTA_CFG="lta.yaml", PNR_FT=True, OSCC_FT=True, ACTION_FT=True),
                 CHECKPOINT_FILE_PATH_AR="ar.ckpt", CHECKPOINT_FILE_PATH_LTA="lta.ckpt")
        m = hoi_lta.TaskFusionMFTransformerLTA4Task(cfg)
        assert built == ["pnr", "oscc", "slowfast", "lta"]
        assert m.oscc_model.kw["no_temp_pool"] is False and m.action_model.kw["num_classes"] == [256] and m.lta_model.kw["build_decoder"] is True
        assert not m.pnr_model.w.requires_grad and not m.lta_model.w.requires_grad
        assert not m.action_model.w.requires_grad and m.action_model.head.weight.requires_grad      # head stays trainable
        # xavier init ran BEFORE the backbones were attached: their parameters are untouched
        assert m.pnr_model.w.item() == 1.0
        x_lta = [torch.zeros(2, 2, 3, 8, 4, 4), torch.zeros(2, 2, 3, 32, 4, 4)]
        with pytest.raises(_lib.EgxError, match="no CPU fallback"):
            m(x_lta, torch.zeros(2, 2, 3, 16, 4, 4))                                           # real forward(x_lta, x_pnr)
        del built[:]
        m2 = hoi_lta.TaskFusionMFTransformer2Task(cfg)
        assert built == ["slowfast", "lta"] and m2.lta_model.kw["build_decoder"] is False
        with pytest.raises(_lib.EgxError, match="no CPU fallback"):
            m2(x_lta)
        del built[:]
        p = hoi_pnr.TaskFusionMFTransformer3TaskDropout(cfg)
        assert built == ["pnr", "oscc", "slowfast"] and p.oscc_model.kw["no_temp_pool"] is True
        assert p.recognition_model.kw["with_head"] is False and p.recognition_model.kw["loader"] == "recognition"
        assert not p.pnr_model.training                                                        # .eval() as the reference does under *_FT
        del built[:]
        v = hoi_pnr.TaskFusionMFTransformer(cfg)
        assert built == ["pnr", "oscc"]
        with pytest.raises(_lib.EgxError, match="no CPU fallback"):
            v([torch.zeros(2, 3, 16, 4, 4)])                                                   # real forward(x)
        del built[:]
        a3 = hoi_ar.TaskFusionMFTransformer3Task(cfg)
        a2 = hoi_ar.TaskFusionMFTransformer2TaskAR(cfg)
        assert built == ["pnr", "oscc", "slowfast", "slowfast", "lta"] and hasattr(a3, "recognition_model") and hasattr(a2, "lta_model")
        del built[:]
        args = NS(hidden_dim=256, num_heads=8, num_layers=1, dropout=0.0, pnr_cfg_file="p", oscc_cfg_file="o", action_cfg_file="a", lta_cfg_file="l")
        from egot2_amd.synth import HOI_G_VOCAB
        g = hoi_multitask.TaskTranslationPromptTransformer6Task(args, HOI_G_VOCAB)
        assert built == ["pnr", "oscc", "slowfast", "lta"] and g.recognition_model.kw["num_classes"] == [256]
        del built[:]
        g2 = hoi_multitask.TaskTranslationPromptTransformer2Task(args, HOI_G_VOCAB)
        assert built == ["pnr", "oscc"] and g2.oscc_model.kw["no_temp_pool"] is True
        with pytest.raises(_lib.EgxError, match="no CPU fallback"):
            g2([torch.zeros(2, 3, 16, 4, 4)], torch.zeros(2, 2, dtype=torch.long))             # real forward(video_pnr, target)
        return True

    assert _with_hoi_factories(run)


def test_hoi_backbone_without_reference_tree_or_factory_raises():
    """A config that names a backbone, outside the reference tree and with no factory registered: loud ImportError."""
    from types import SimpleNamespace as NS
    from egot2_amd import hoi_lta
    cfg = NS(FORECASTING=NS(NUM_INPUT_CLIPS=2, NUM_ACTIONS_TO_PREDICT=3),
             MODEL=NS(TRANSLATION_HEADS=8, TRANSLATION_LAYERS=1, TRANSLATION_INPUT_FEATURES=256, TRANSLATION_DROPOUT=0.0,
                      NUM_CLASSES=[5, 7], DROPOUT_RATE=0.0, HEAD_ACT="softmax"), TEST=NS(NO_ACT=False),
             PRETRAIN=NS(PNR_CFG="pnr.yaml", OSCC_CFG=None))
    with pytest.raises(ImportError, match="register_backbone_factory"):
        hoi_lta.TaskFusionMFTransformerLTA4Task(cfg)


def test_lossAV_mirror_has_the_reference_interface():
    """hhi_asd.lossAV (HHI/tasks/asd/loss.py:11-30): constructor argument, parameter / buffer names and the criterion's class
    weights, checked without a GPU; the fused kernels behind forward() are covered by the GPU suite."""
    import torch
    from egot2_amd import hhi_asd
    m = hhi_asd.lossAV(128)
    assert set(m.state_dict()) == {"criterion.weight", "FC.weight", "FC.bias"}
    assert tuple(m.FC.weight.shape) == (2, 128) and m.criterion.weight.tolist() == [1.0, 4.0]
    assert hhi_asd.lossAV().FC.in_features == 256
    with pytest.raises(Exception):          # the HIP path fails loudly on CPU tensors instead of falling back
        m(torch.randn(4, 1, 128), torch.zeros(4, dtype=torch.int64))


def test_feature_cache_and_sink_host_logic(tmp_path):
    """Row F4 producer side, host logic only (no kernels): the on-disk Stage-II cache round-trips packed bf16 / fp32 streams
    bit-exactly and atomically replaces files; the sink validates block shapes; the drop-in PNR / OSCC head has the reference
    head's parameter names (HOI/models/pnr/head_helper.py:338: `projection = nn.Linear(8192, num_classes)`)."""
    from egot2_amd import _lib
    from egot2_amd.feature_sink import FeatureCache, FeatureSink, PooledFeatureHead
    cache = FeatureCache(str(tmp_path / "c"))
    feats = {"pnr": torch.randn(1, 4, 64).bfloat16(), "action": torch.randn(1, 4, 16)}
    assert not cache.has("uid123:0-8")
    cache.save("uid123:0-8", feats)
    cache.save("uid123:0-8", feats)                 # overwrite is atomic (os.replace)
    got = cache.load("uid123:0-8")
    assert set(got) == set(feats) and all(torch.equal(got[k], feats[k]) and got[k].dtype == feats[k].dtype for k in feats)
    sink = FeatureSink("cpu", torch.float32)
    out = cache.load_batch(["uid123:0-8", "uid123:0-8"], sink)
    assert tuple(out["pnr"].shape) == (2, 4, 64) and out["pnr"].dtype == torch.bfloat16 and torch.equal(out["pnr"][1], feats["pnr"][0])
    sink.alloc("x", 2, 6, 8)
    sink.put("x", torch.ones(2, 3, 8), t0=3)
    assert sink.get("x")[:, 3:].eq(1).all()
    with pytest.raises(_lib.EgxError):
        sink.put("x", torch.ones(2, 4, 8), t0=3)    # runs past the stream
    with pytest.raises(_lib.EgxError):
        sink.put_pooled_map("x", torch.zeros(2, 2, 1, 2, 2), (1, 1, 1))     # CPU tensor: no fallback
    head = PooledFeatureHead([2048], 17, [[1, 7, 7]], act_func="softmax_1")
    assert set(head.state_dict()) == {"projection.weight", "projection.bias"} and tuple(head.projection.weight.shape) == (17, 8192)
