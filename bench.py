#!/usr/bin/env python3
"""Headline benchmark: clips/s, forward+backward, 3-task TTM translator (B=256 per GPU, T=15, d=128, h=4, L=1,
d_ff=2048) on synthetic backbone features — BASELINE.json configs[1].

  python bench.py --gpus N --steps K --warmup W      (N > 1: launched by torch.distributed.run, one rank per GPU)

A step = token preparation + encoder + pooled head + weighted CE + full backward (+ one RCCL all-reduce of the
flat gradient buffer when N > 1) over one batch of 256 clips per GPU (weak scaling). Inputs are resident in HBM
before the timed region. Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_TFLOPS = {"f32": 157.3, "bf16": 2500.0}   # dense MFMA peaks, /opt/skills/guides/MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0


def algorithmic_flops(B, K, T, d_in, d, h, L, d_ff, n_out=2):
    """GEMM FLOPs (2mnk), softmax/LN/elementwise excluded — BASELINE.md §2."""
    S = K * T
    N = B * S
    proj = K * 2 * B * T * d_in * d
    layer = 2 * N * d * 3 * d + 4 * B * S * S * d + 2 * N * d * d + 4 * N * d * d_ff
    head = 2 * B * d * n_out
    fwd = proj + L * layer + head
    bwd = 2 * fwd - proj  # no dX into the frozen features
    return fwd, bwd


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=256, help="clips per GPU")
    ap.add_argument("--frames", type=int, default=15)
    ap.add_argument("--layers", type=int, default=1)
    ap.add_argument("--dtype", default="f32", choices=["f32", "bf16"])
    ap.add_argument("--impl", default="auto", choices=["auto", "generic", "fused"])
    ap.add_argument("--dropout", type=float, default=0.5, help="encoder dropout (reference recipe README.md:84 uses 0.5)")
    ap.add_argument("--optimizer", action="store_true", help="run the Adam update inside the timed step (headline excludes it by default)")
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel eagerly instead of replaying one captured hipGraph per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-optimizer-line", action="store_true", help="skip the separate fwd+bwd+Adam measurement")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--overlap", action="store_true", help="world > 1: start the all-reduce of all but the late gradients while the backward's tail (grouped small weight gradients) still runs; default: one all-reduce after the backward")
    ap.add_argument("--force-dist", action="store_true", help="initialise RCCL and run the gradient all-reduce path even with one rank (self-test)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the translator has no CPU path")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    multi = world > 1 or args.force_dist          # take the distributed code path
    if multi:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=dev)
    if args.gpus != world and rank == 0:
        print(f"[bench] note: --gpus {args.gpus} but WORLD_SIZE={world}; using {world}", file=sys.stderr)

    from egot2_amd import ddp, hhi_ttm, _lib
    from tests.util import hhi_args
    lib = _lib.load()

    B, T, K, d, h, L, dff = args.batch, args.frames, 3, 128, 4, args.layers, 2048
    torch.manual_seed(0)
    model = hhi_ttm.TaskFusionMFTransformer3Task(hhi_args(hidden_dim=d, num_heads=h, dropout=args.dropout, num_layers=L))
    model = model.to(dev).set_compute(args.dtype, args.impl).train()
    ddp.broadcast_parameters(model)
    params = [p for p in model.parameters() if p.requires_grad]

    from egot2_amd.train import CrossEntropyLoss, FusedAdam
    g = torch.Generator().manual_seed(1234 + rank)
    feats = [torch.randn(B, T, 256, generator=g).to(dev) for _ in range(K)]
    target = torch.randint(0, 2, (B,), generator=g).to(dev)
    criterion = CrossEntropyLoss(torch.FloatTensor([0.266, 0.734])).to(dev)   # video_task_2loader.py:21-22
    from egot2_amd.functional import unit_grad
    one = unit_grad(dev)

    def fwd_bwd():
        for p in params:
            p.grad = None
        loss = criterion(model.forward_features(*feats), target)
        loss.backward(gradient=one)     # persistent 1.0: no ones-fill, and the fused CE skips the multiply by it
        return loss

    def sync():
        torch.cuda.synchronize()
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    def capture(fn):
        """One hipGraph for `fn`: the library only enqueues kernels on the current stream and the dropout seed / Adam
        step count live in device memory (advanced inside the graph), so a replay is exact training work."""
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                fn()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        # thread_local: with world > 1 the RCCL watchdog thread may touch the HIP runtime while this thread records
        with torch.cuda.graph(gr, capture_error_mode="thread_local"):
            fn()
        torch.cuda.synchronize()
        return gr

    def timed(step_fn, warmup, steps):
        for _ in range(warmup):
            step_fn()
        sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            step_fn()
        sync()
        dt = time.perf_counter() - t0
        if multi:
            t = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = t.item()
        return dt

    use_graph = not args.no_graph
    if use_graph:
        model.enable_device_seed()
    opt = FusedAdam(params, lr=5e-4) if args.optimizer else None     # Adam(lr=5e-4): video_task_2loader.py:62-64

    from egot2_amd import functional as F_egx

    def staged_backward_ok():
        """The overlapped exchange needs the staged backward (fused path): check it against the one-shot backward."""
        if not use_graph:
            return False
        try:
            grads = []
            for defer in (False, True):
                model._egx_seed_dev.fill_(12345)
                model.egx_defer_small = defer
                fwd_bwd()
                F_egx.run_deferred()
                grads.append([p.grad.detach().clone() for p in params])
            torch.cuda.synchronize()
            ok = F_egx.last_grad_layout.get("late_floats", 0) > 0 and all(
                torch.allclose(a, b, rtol=1e-3, atol=1e-5) for a, b in zip(*grads))
        except Exception as e:          # noqa: BLE001  (any failure -> the plain path)
            print(f"[bench] staged backward unavailable ({e}); using the plain all-reduce", file=sys.stderr)
            ok = False
        model.egx_defer_small = False
        return ok

    overlap = multi and args.overlap and staged_backward_ok()

    def make_step(with_opt):
        """fwd + weighted CE + bwd (+ gradient all-reduce over RCCL when world > 1) (+ Adam)."""
        if overlap:
            # graph 1: forward + loss + backward up to the grouped small weight gradients; graph 2: those. The all-reduce
            # of everything else runs on RCCL's stream while graph 2 executes, then the late region follows.
            model.egx_defer_small = True
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(3):
                    fwd_bwd()
                    F_egx.run_deferred()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            g1, g2 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
            with torch.cuda.graph(g1, capture_error_mode="thread_local"):
                fwd_bwd()
            with torch.cuda.graph(g2, pool=g1.pool(), capture_error_mode="thread_local"):
                F_egx.run_deferred()
            torch.cuda.synchronize()

            def step_overlapped():
                g1.replay()
                ddp.allreduce_gradients_overlapped(g2.replay, force=args.force_dist)
                if with_opt:
                    with_opt.step()
            return step_overlapped
        if with_opt and not multi:
            def body():
                fwd_bwd()
                with_opt.step()
            gr = capture(body) if use_graph else None
            return gr.replay if gr is not None else body
        gr = capture(fwd_bwd) if use_graph else None

        def step():
            if gr is not None:
                gr.replay()
            else:
                fwd_bwd()
            ddp.allreduce_gradients(params, force=args.force_dist)
            if with_opt:
                with_opt.step()
        return step

    step = make_step(opt)
    dt = timed(step, args.warmup, args.steps)
    graph = use_graph
    ms_per_step = dt / args.steps * 1e3
    value = B * world * args.steps / dt
    fwd_f, bwd_f = algorithmic_flops(B, K, T, 256, d, h, L, dff)

    out = {
        "metric": "clips/sec fwd+bwd, 3-task TTM translator (B=256,T=15,d=128)",
        "value": value, "unit": "clips/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": f"configs[1]: TTM 3-task translator (LAM+TTM+ASD), {L} layer d=128 h=4 d_ff=2048, "
                               f"B={B}/GPU T={T} S={K * T}, synthetic N(0,1) features, random-init weights, "
                               f"train mode dropout={args.dropout} (+0.1 on PE), weighted CE, fwd+bwd"
                               + (" + FusedAdam" if opt else "") + ((" + RCCL grad all-reduce" + (" overlapped with the backward tail" if overlap else "")) if multi else ""),
                   "global_batch": B * world, "parallelism": f"dp{world}", "impl": args.impl,
                   "launch": "one hipGraph replay per step" if graph else "eager"},
        "step_tflops": (fwd_f + bwd_f) / (ms_per_step * 1e-3) / 1e12,
        "step_frac_of_mfma_peak": (fwd_f + bwd_f) / (ms_per_step * 1e-3) / 1e12 / PEAK_TFLOPS[args.dtype],
    }

    if not args.optimizer and not args.no_optimizer_line:
        # "+ optimizer step reported separately" (SURVEY.md 8d): the same step with the Adam update inside
        opt2 = FusedAdam(params, lr=5e-4)
        step2 = make_step(opt2)
        dt2 = timed(step2, max(3, args.warmup // 2), args.steps)
        out["with_optimizer"] = {"optimizer": "FusedAdam(lr=5e-4), one launch over the flat parameter buffer",
                                 "ms_per_step": dt2 / args.steps * 1e3, "value": B * world * args.steps / dt2, "unit": "clips/s"}
    if rank == 0 and not args.no_roofline:
        out["roofline"] = measure_roofline(torch, lib, fwd_bwd, B, K, T, d, h, L, dff, args.dtype)
    if rank == 0 and not args.no_cpu_baseline and world == 1:
        from oracle.stock_module import time_cpu_baseline
        out["cpu_baseline"] = time_cpu_baseline(B=B, T=T, n_tasks=K, dim=d, n_heads=h, num_layers=L, dropout=args.dropout)
    if rank == 0:
        # RCCL prints its version banner through C stdio (NCCL_DEBUG=VERSION is exported on the GPU boxes); push it out
        # first so that the JSON line is the LAST line of stdout
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass
        print(json.dumps(out), flush=True)
    if multi:
        dist.barrier()
        dist.destroy_process_group()


def pmc_traffic(dtype, B, T, L, kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (tools/pmc_traffic.sh: FETCH_SIZE and
    WRITE_SIZE in separate runs, KiB units, FETCH_SIZE doubled for gfx950 as MI355X_MICROARCH.md prescribes). PMC
    collection needs the profiler, so the number is read from profiles/ and only when it was taken on this workload."""
    import glob
    here = os.path.dirname(os.path.abspath(__file__))
    for f in sorted(glob.glob(os.path.join(here, "profiles", f"r*_pmc_{dtype}.json")), reverse=True):
        try:
            d = json.load(open(f))
            w = d.get("workload", {})
            if (w.get("batch"), w.get("frames"), w.get("layers")) == (B, T, L) and kernel in d["kernels"]:
                return d["kernels"][kernel]["traffic_bytes"], os.path.relpath(f, here)
        except (OSError, ValueError, KeyError):
            continue
    return None, None


def measure_roofline(torch, lib, step, B, K, T, d, h, L, dff, dtype):
    """Dominant kernel = fused_bwd_kernel (the per-clip backward: head, LayerNorm, attention and projection input
    gradients and the FFN input gradient dH = (W2^T g) .* alive, dX1 = W1^T dH). Its ALGORITHMIC FLOPs per launch are the
    dX-type GEMMs of the backward (BASELINE.md §2 accounting; the QKV / attention-probability recompute is NOT counted):
    L * (2N d 3d + 2N d^2 + 4N d d_ff + 8 B S^2 d). Average launch duration is measured live with hipEvents recorded on
    the launch stream (egx_timing_*), over eager launches of the same step."""
    import ctypes as C
    S = K * T
    N = B * S
    ffn = L * 4.0 * N * d * dff
    attn_fwd = K * 2.0 * B * T * 256 * d + L * (2.0 * N * d * 3 * d + 4.0 * B * S * S * d + 2.0 * N * d * d)
    flops = {
        "fused_bwd_kernel": L * (2.0 * N * d * 3 * d + 2.0 * N * d * d + 8.0 * B * S * S * d) + ffn,
        "fused_fwd_kernel": attn_fwd + ffn,
        "ffn_dw_kernel": ffn,
        "ffn_fwd_kernel": ffn,
        "ffn_bwd_kernel": ffn,
    }
    lib.egx_timing_enable(1)
    reps = 16
    for _ in range(reps):
        step()
    torch.cuda.synchronize()
    res = {}
    for which, name in enumerate(("fused_fwd_kernel", "fused_bwd_kernel", "ffn_dw_kernel", "ffn_fwd_kernel", "ffn_bwd_kernel")):
        tot, cnt = C.c_double(0), C.c_int(0)
        if lib.egx_timing_read(which, C.byref(tot), C.byref(cnt)) == 0 and cnt.value:
            res[name] = tot.value / cnt.value * 1e-3
    # split mode: the FFN halves run as their own launches; the per-clip kernels then hold only the attention halves
    if "ffn_fwd_kernel" in res:
        flops["fused_fwd_kernel"] -= ffn
    if "ffn_bwd_kernel" in res:
        flops["fused_bwd_kernel"] -= ffn
    lib.egx_timing_enable(0)
    if "fused_bwd_kernel" not in res:   # generic path (shape outside the fused kernels)
        return {"bound": "mfma", "kernel": None, "achieved": None, "peak": PEAK_TFLOPS[dtype], "unit": "TFLOP/s",
                "frac": None, "traffic": None}
    t = res["fused_bwd_kernel"]
    ach = flops["fused_bwd_kernel"] / t / 1e12
    traffic, traffic_src = pmc_traffic(dtype, B, T, L, "egx::fused_bwd_kernel")
    return {"bound": "mfma", "kernel": "egx::fused_bwd_kernel", "achieved": ach, "peak": PEAK_TFLOPS[dtype],
            "unit": "TFLOP/s", "frac": ach / PEAK_TFLOPS[dtype], "traffic": traffic, "traffic_unit": "bytes/launch",
            "traffic_source": traffic_src,
            "flops_per_launch": flops["fused_bwd_kernel"], "avg_launch_us": t * 1e6,
            "other_kernels": {k: {"avg_launch_us": v * 1e6, "achieved_tflops": flops[k] / v / 1e12,
                                  "frac": flops[k] / v / 1e12 / PEAK_TFLOPS[dtype]} for k, v in res.items() if k != "fused_bwd_kernel"}}


if __name__ == "__main__":
    main()
