#!/usr/bin/env python3
"""Headline benchmark: clips/s, forward+backward, 3-task TTM translator (B=256 per GPU, T=15, d=128, h=4, L=1,
d_ff=2048) on synthetic backbone features — BASELINE.json configs[1]. `--config c1|c3|c4|c5hhi|c5hoi` times the other
BASELINE configurations the same way (each with its own roofline block). The default line is configs[1] in "f32s"
arithmetic (fp32 storage / accumulation / results, every product formed from an exact three-way bf16 split of its fp32
operands: the top-level "arithmetic" field says so) with the exact-fp32-MFMA step timed beside it as `native_f32`.

  python bench.py --gpus N --steps K --warmup W

With N > 1 and no WORLD_SIZE in the environment the script launches its own ranks — `python -m torch.distributed.run
--nnodes=1 --nproc-per-node N --master-addr 127.0.0.1` on this same file, as a CHILD process started before anything
touches the GPU (the reference does the same through Lightning's `strategy="ddp"`, HOI/scripts/multitask/run.py:41-50)
— relays their output and exits with their code; launched under torchrun it reads RANK / LOCAL_RANK / WORLD_SIZE.

A step = token preparation + encoder + task head + loss + full backward (+ one RCCL all-reduce of the flat gradient
buffer when N > 1) over one batch of 256 clips per GPU (weak scaling). Inputs are resident in HBM before the timed
region. Each trial times EXACTLY --steps steps between barrier + synchronize pairs (max over ranks); `ms_per_step` is
the median trial, p10 / p90 are reported next to it. Rank 0 prints ONE JSON line (the last line of stdout).
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# dense MFMA peaks, /opt/skills/guides/MI355X_MICROARCH.md. "f32s" (fp32 operands split into three bf16 parts, six bf16 MFMAs
# per K-block) is bounded by the bf16 pipe at six instructions per algorithmic K-block: 2500 / 6.
PEAK_TFLOPS = {"f32": 157.3, "bf16": 2500.0, "f32s": 2500.0 / 6.0}
DTYPE_LABEL = {"f32s": "f32 (operands split exactly into 3 bf16 parts; 6 bf16 MFMA products per K-block, f32 accumulate)"}
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


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--trials", type=int, default=0, help="timed repetitions of --steps (0 = auto: >= 5 and >= --min-seconds of timed region)")
    ap.add_argument("--min-seconds", type=float, default=3.6, help="auto trials: total timed GPU work to aim for")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: --batch clips per GPU; strong: --batch clips in total, sharded over the ranks (SURVEY.md 8e)")
    ap.add_argument("--config", default="c2", choices=["c1", "c2", "c3", "c4", "c5hhi", "c5hoi", "pnr"],
                    help="BASELINE.json configuration (default c2 = configs[1], the metric)")
    ap.add_argument("--batch", type=int, default=0,
                    help="clips per GPU (0 = 256; with --frames > 16 on c1 - c3: 3840 // frames, i.e. the headline's 3840 frames per task - "
                         "the reference's sampler packs B * T ~ 400, HHI/dataset/ttm/sampler.py:41, this is 9.6 such batches)")
    ap.add_argument("--frames", type=int, default=15)
    ap.add_argument("--layers", type=int, default=0, help="0 = the configuration's own depth")
    ap.add_argument("--dtype", default=None, choices=["f32", "bf16", "f32s"],
                    help="default: f32s for c1/c2 (fp32-grade split-bf16 arithmetic; f32 = exact fp32 MFMA), bf16 for c3..c5 (BASELINE.json)")
    ap.add_argument("--graph-collectives", action="store_true", help="N > 1 (or --force-dist): capture the gradient all-reduce(s) inside the step's hipGraph")
    ap.add_argument("--no-native-line", action="store_true", help="f32s runs: skip the reference timing of the exact fp32 MFMA path")
    ap.add_argument("--impl", default="auto", choices=["auto", "generic", "fused", "wide", "tiled"])
    ap.add_argument("--dropout", type=float, default=None, help="encoder dropout (default: the reference recipe of the configuration)")
    ap.add_argument("--optimizer", action="store_true", help="run the Adam update inside the timed step (headline excludes it by default)")
    ap.add_argument("--graph", action="store_true", help="(default since round 5: every configuration replays as one hipGraph; kept for old command lines)")
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel eagerly instead of replaying one captured hipGraph per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-optimizer-line", action="store_true", help="skip the separate fwd+bwd+Adam measurement")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-overlap", action="store_true", help="world > 1: one all-reduce after the whole backward instead of the exchange overlapped with the backward's tail")
    ap.add_argument("--overlap", action="store_true", help="(default for world > 1; kept for compatibility)")
    ap.add_argument("--force-dist", action="store_true", help="initialise RCCL and run the gradient all-reduce path even with one rank (self-test)")
    ap.add_argument("--launch-selftest", action="store_true",
                    help="exercise only the launcher plumbing (rank spawn, process group on --backend, barrier, max-over-ranks timing, JSON relay) without the model; CPU-runnable")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="process-group backend (gloo only for --launch-selftest)")
    ap.add_argument("--encoder-only", action="store_true", help="c5*: time the task-translation encoder alone (loss = sum of the memory), without the sequence decoder + vocabulary CE")
    ap.add_argument("--deterministic", action="store_true", help="fixed-order reductions in the backward (bit-identical gradients run to run)")
    ap.add_argument("--feat-dtype", default="f32", choices=["f32", "bf16"], help="c4: dtype of the backbone features handed to the translator (row F4)")
    ap.add_argument("--feat-source", default="tensor", choices=["tensor", "sink"],
                    help="c4: 'sink' = the PNR / OSCC token rows are produced by egx_pool_pack into a FeatureSink (row F4 producer side) "
                         "and read by the translator as packed bf16 in place; the producer kernel is timed separately (`producer`)")
    ap.add_argument("--feat-frames", type=int, default=1, help="c4: per-frame PNR / OSCC features, this many frames per clip, temporal mean fused into the hand-off")
    ap.add_argument("--master-port", type=int, default=0)
    ap.add_argument("--no-weight-cache", action="store_true",
                    help="per-clip kernels: pack the weights into MFMA-fragment order inside every step (rounds 1-5) instead of once, outside the timed "
                         "loop (forward + backward only: the weights do not change between steps; the with_optimizer line always re-packs)")
    ap.add_argument("--no-fused-ce", action="store_true", help="c1 / c2: the weighted cross entropy as a launch of its own (rounds 1-5) instead of in the forward's head epilogue")
    return ap.parse_args(argv)


# ---- N > 1 without torchrun: launch the ranks ourselves ---------------------------------------------------------------
def self_launch(args, argv) -> int:
    """Start `args.gpus` ranks of this script under torch.distributed.run as a child process. This process has made no
    HIP call (torch is not even imported), so nothing is re-exec'd after GPU initialisation. The child's stdout is relayed;
    the JSON line of rank 0 is re-printed LAST. Returns the child's exit code."""
    port = args.master_port or (29600 + os.getpid() % 2000)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    print(f"[bench] launching {args.gpus} ranks: {' '.join(cmd)}", file=sys.stderr, flush=True)
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True)
    last_json = None
    for line in proc.stdout:
        s = line.strip()
        if s.startswith("{") and s.endswith("}") and '"metric"' in s:
            last_json = s
        else:
            sys.stdout.write(line)
    rc = proc.wait()
    sys.stdout.flush()
    if rc != 0:
        print(f"[bench] rank launch failed with exit code {rc}", file=sys.stderr)
        return rc
    if last_json is None:
        print("[bench] the ranks produced no JSON line", file=sys.stderr)
        return 1
    print(last_json, flush=True)
    return 0


def percentile(xs, q):
    xs = sorted(xs)
    if len(xs) == 1:
        return xs[0]
    pos = q * (len(xs) - 1)
    lo = int(pos)
    hi = min(lo + 1, len(xs) - 1)
    return xs[lo] + (xs[hi] - xs[lo]) * (pos - lo)


def launch_selftest(args):
    """The distributed plumbing of bench.py without the model: process group, barrier-bracketed timing with max over
    ranks, one JSON line from rank 0. Runs on CPU with --backend gloo (tests/test_cpu_host.py)."""
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world > 1:
        dist.init_process_group(args.backend)
    t = torch.tensor([float(rank + 1)])
    if world > 1:
        dist.all_reduce(t)
        dist.barrier()
    t0 = time.perf_counter()
    time.sleep(0.01 * (rank + 1))
    dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    if world > 1:
        dist.barrier()
        dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    if rank == 0:
        print(json.dumps({"metric": "launch selftest", "value": None, "n_gpus": world, "rank_sum": t.item(),
                          "max_rank_seconds": dt.item(), "config": {"parallelism": f"dp{world}"}, "selftest": True}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


def resolve_batch(args) -> int:
    """--batch 0 = the configuration's default: 256 clips, or 3840 frames per task for the long-sequence workloads."""
    if args.batch > 0:
        return args.batch
    return 3840 // args.frames if (args.config in ("c1", "c2", "c3") and args.frames > 16) else 256


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse_args(argv)
    env_world = os.environ.get("WORLD_SIZE")
    if args.gpus > 1 and env_world is None:
        return self_launch(args, argv)                  # before ANY torch / HIP call in this process
    if env_world is not None and int(env_world) != args.gpus:
        # launched under torchrun: the environment's world size is the truth (a harness may not repeat --gpus)
        print(f"[bench] --gpus {args.gpus} but WORLD_SIZE={env_world}: using the environment's world size", file=sys.stderr)
        args.gpus = int(env_world)
    if args.launch_selftest:
        return launch_selftest(args)
    return run(args)


def run(args) -> int:
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
            os.environ.setdefault("MASTER_PORT", str(args.master_port or 29533))
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=dev)

    from egot2_amd import ddp, synth, _lib
    from egot2_amd import functional as F_egx
    from egot2_amd.train import FusedAdam
    lib = _lib.load()

    # headline arithmetic of c1 / c2: fp32 operands split exactly into three bf16 parts, six bf16 MFMA products per K-block,
    # fp32 accumulation ("f32s": fp32-grade results, tests/test_gpu_translator.py::test_split_bf16_mode_is_fp32_grade);
    # `--dtype f32` is the exact v_mfma_f32_16x16x4_f32 path, reported beside it as `native_f32`
    run_dtype = args.dtype or ("f32s" if args.config in ("c1", "c2") else None)
    args.batch = resolve_batch(args)
    # weak scaling: --batch clips on every GPU. strong scaling ("global B = 256 -> 32 clips / GPU at n = 8"): the
    # --batch clips are sharded over the ranks, the per-GPU work shrinks with N
    if args.scaling == "strong":
        if args.batch % world:
            raise SystemExit(f"--scaling strong: --batch {args.batch} is not divisible by {world} ranks")
        local_batch = args.batch // world
    else:
        local_batch = args.batch
    wl = synth.make_workload(args.config, dev, batch=local_batch, frames=args.frames, layers=args.layers or None,
                             dtype=run_dtype, impl=args.impl, dropout=args.dropout, seed=1234 + rank, encoder_only=args.encoder_only,
                             feat_dtype=args.feat_dtype, feat_frames=args.feat_frames, feat_source=args.feat_source,
                             fused_ce=not args.no_fused_ce)
    model, params, B = wl["model"], wl["params"], wl["B"]
    dtype = wl["compute"]
    # forward + backward only: the weights are the same in every step, so their MFMA-fragment-packed copies are made ONCE (a persistent
    # cache, model.enable_weight_cache(frozen=True)) instead of by a 6 us launch in front of every forward; a step with the optimizer inside
    # re-packs after every update (frozen = False below)
    wcache_on = hasattr(model, "enable_weight_cache") and not args.no_weight_cache
    if wcache_on:
        model.enable_weight_cache(frozen=not args.optimizer)
    if args.deterministic:
        model.set_deterministic(True)
    ddp.broadcast_parameters(model)
    egx_comm = None
    if args.graph_collectives and multi:
        # Captured collectives run on the C ABI's own RCCL communicator (egx_allreduce on streams this process owns): no ProcessGroupNCCL work
        # objects and no watchdog thread inside or beside the capture (round 5: hipErrorCapturedEvent once in six runs). The 128-byte id travels
        # over the torch process group that launched the ranks.
        def _bcast(idb):
            box = [idb]
            dist.broadcast_object_list(box, src=0)
            return box[0]
        egx_comm = ddp.EgxComm(rank, world, bcast=_bcast if world > 1 else None)
        ddp.use_egx_comm(egx_comm)
    one = F_egx.unit_grad(dev)
    loss_fn = wl["loss_fn"]

    def fwd_bwd():
        for p in params:
            p.grad = None
        loss = loss_fn()
        loss.backward(gradient=one)     # persistent 1.0: no ones-fill, and the fused CE skips the multiply by it
        return loss

    def sync():
        torch.cuda.synchronize()
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    def capture(fn, warm=3):
        """One hipGraph for `fn`: the library only enqueues kernels on the current stream and the dropout seed / Adam
        step count live in device memory (advanced inside the graph), so a replay is exact training work."""
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warm):
                fn()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        # thread_local: with world > 1 the RCCL watchdog thread may touch the HIP runtime while this thread records
        with torch.cuda.graph(gr, capture_error_mode="thread_local"):
            fn()
        torch.cuda.synchronize()
        return gr

    def time_trials(step_fn, warmup, steps, trials):
        """`trials` timed regions of exactly `steps` steps each, barrier + synchronize on both sides, max over ranks."""
        for _ in range(warmup):
            step_fn()
        out = []
        for _ in range(trials):
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
            out.append(dt)
        return out

    def auto_trials(step_fn, steps):
        if args.trials > 0:
            return args.trials
        sync()
        t0 = time.perf_counter()
        for _ in range(20):     # (three steps put the sync latency into the estimate: 2.95 s of timed work for a 3.6 s target)
            step_fn()
        sync()
        est = (time.perf_counter() - t0) / 20 * steps
        n = max(5, int(args.min_seconds / max(est, 1e-6)) + 1)
        n = min(n, 2000)
        if multi:       # every rank must run the same number of trials
            t = torch.tensor([n], device=dev, dtype=torch.int64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            n = int(t.item())
        return n

    # device-resident dropout seed: the per-clip / tiled kernels derive their keys in-kernel, the wide bf16 path and the fused
    # decoder from a key table computed on the stream (round 4) - every configuration replays as one hipGraph
    # (round 4: c5hoi measured 7-10 % slower as one graph than with eager launches - the decoder backward's side-stream forks became
    # graph branches; round 5: under capture its weight gradients leave as one grouped launch instead (wide_gemm.hip wide_tn_queue_*)
    # and the captured step is the faster one, 3.68 vs 3.83 ms: profiles/r05_dec_group.txt. --no-graph keeps eager launches.)
    use_graph = not args.no_graph
    if use_graph:
        model.enable_device_seed()
    opt = FusedAdam(params, lr=5e-4) if args.optimizer else None     # Adam(lr=5e-4): video_task_2loader.py:62-64

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

    overlap = multi and not args.no_overlap and staged_backward_ok()
    # wide / generic configurations: per-layer buckets exchanged under the backward (eager launches: the bucket announcements are
    # host callbacks) - or, with --graph-collectives, the same exchange CAPTURED with the step (announcements happen once, at
    # capture time; the collectives replay from RCCL's stream as a parallel branch of the graph)
    bucketed = multi and not args.no_overlap and not overlap and (wl["name"] in ("c4", "c5", "c5hhi", "c5hoi") or not use_graph)
    exch = {"collectives": None}

    def place_optimizer(with_opt):
        """FusedAdam re-points the parameters into its flat buffer on its FIRST step; a graph captured before that would
        keep reading the old (freed) parameter storage. Run one eager step first so that the capture sees the final
        addresses, and assert they no longer move."""
        if with_opt is None:
            return
        fwd_bwd()
        F_egx.run_deferred()
        ddp.allreduce_gradients(params, force=args.force_dist)
        with_opt.step()
        before = [p.data_ptr() for p in params]
        fwd_bwd()
        F_egx.run_deferred()
        with_opt.step()
        torch.cuda.synchronize()
        assert before == [p.data_ptr() for p in params], "FusedAdam moved a parameter after its first step"

    def make_step(with_opt):
        """fwd + loss + bwd (+ gradient all-reduce over RCCL when world > 1) (+ Adam)."""
        place_optimizer(with_opt)
        if args.graph_collectives and use_graph and (multi or args.force_dist):
            # opt-in: the WHOLE step incl. the RCCL collectives as one hipGraph (RCCL kernels are capturable); no host work
            # between the backward, the exchange and the update. Not the default: a failed capture of a collective cannot
            # be recovered from reliably inside a multi-rank job.
            model.egx_defer_small = bool(overlap)

            def body():
                if bucketed:
                    for p in params:
                        p.grad = None
                    with ddp.BucketedExchange(params, force=args.force_dist) as ex:
                        fwd_bwd()
                    exch["collectives"] = ex.collectives
                else:
                    fwd_bwd()
                    if overlap:
                        ddp.allreduce_gradients_overlapped(F_egx.run_deferred, params, force=args.force_dist)
                    else:
                        ddp.allreduce_gradients(params, force=args.force_dist)
                if with_opt:
                    with_opt.step()
            gr = capture(body)
            return gr.replay
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
                ddp.allreduce_gradients_overlapped(g2.replay, params, force=args.force_dist)
                if with_opt:
                    with_opt.step()
            return step_overlapped
        model.egx_defer_small = False
        if bucketed:
            # wide / generic paths (c4, c5*): the backward announces its gradient buffer slice by slice (per encoder layer, last
            # layer first; the decoder's buffer as a whole) and every slice is all-reduced while the rest of the backward runs
            def step_bucketed():
                for p in params:            # the exchange requires a backward that CREATES every .grad
                    p.grad = None
                with ddp.BucketedExchange(params, force=args.force_dist) as ex:
                    fwd_bwd()
                exch["collectives"] = ex.collectives
                if with_opt:
                    with_opt.step()
            return step_bucketed
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

    try:
        step = make_step(opt)
    except Exception as e:      # noqa: BLE001  (overlapped capture failed: the plain exchange still measures the step)
        if not overlap:
            raise
        print(f"[bench] overlapped exchange unavailable ({e}); using one all-reduce after the backward", file=sys.stderr)
        overlap = False
        model.egx_defer_small = False
        F_egx.run_deferred()
        step = make_step(opt)
    trials = auto_trials(step, args.steps)
    dts = time_trials(step, args.warmup, args.steps, trials)
    dt_med = percentile(dts, 0.5)
    ms = [d / args.steps * 1e3 for d in dts]
    ms_per_step = dt_med / args.steps * 1e3
    value = B * world / (ms_per_step * 1e-3)
    fwd_f, bwd_f = wl["flops"]

    # the exchange on its own (same buffers, same stream order), for the `allreduce_us` field
    allreduce_us = None
    if multi:
        if not any(p.grad is not None for p in params):     # (a captured step leaves .grad as the capture left it; make sure there is something to exchange)
            fwd_bwd()
            F_egx.run_deferred()
        sync()
        t0 = time.perf_counter()
        for _ in range(20):
            ddp.allreduce_gradients(params, force=args.force_dist)
        torch.cuda.synchronize()
        allreduce_us = (time.perf_counter() - t0) / 20 * 1e6
        t = torch.tensor([allreduce_us], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        allreduce_us = t.item()
        for p in params:        # 20 averaged copies later the gradients are still finite; nothing reads them again
            p.grad = None

    # launches per step: kernels the library enqueues for one eager fwd + bwd (a hipGraph replay issues the same nodes)
    launches_per_step = None
    if hasattr(lib, "egx_launch_count"):
        lib.egx_launch_count(1)
        fwd_bwd()
        F_egx.run_deferred()
        torch.cuda.synchronize()
        launches_per_step = int(lib.egx_launch_count(1))
        for p in params:
            p.grad = None

    metric = ("clips/sec fwd+bwd, 3-task TTM translator (B=256,T=15,d=128)" if wl["name"] == "c2"
              else f"clips/sec fwd+bwd, {wl['name']}")
    out = {
        "metric": metric,
        "value": value, "unit": "clips/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
        "dtype": DTYPE_LABEL.get(dtype, dtype), "arithmetic": dtype, "data": "synthetic",
        "trials": trials, "ms_per_step_p10": percentile(ms, 0.1), "ms_per_step_p90": percentile(ms, 0.9),
        "ms_per_step_min": min(ms), "timed_seconds": sum(dts),
        "config": {"workload": wl["describe"] + ", fwd+bwd" + (" + FusedAdam" if opt else "")
                               + ((" + RCCL grad all-reduce" + (" overlapped with the backward tail" if overlap else
                                                                 " in per-layer buckets overlapped with the backward" if bucketed else "")) if multi else ""),
                   "global_batch": B * world, "parallelism": f"dp{world}", "impl": args.impl, "workgroups_per_clip": F_egx.last_encoder_slices(),
                   "launch": "one hipGraph replay per step" if (use_graph and not (bucketed and not args.graph_collectives)) else "eager",
                   "deterministic": bool(args.deterministic),
                   "compute_is_module_default": bool(dtype == getattr(type(model), "egx_compute", None)),
                   "weights_packed": ("once, outside the timed loop (forward + backward only: the weights do not change between steps; "
                                      "with_optimizer re-packs every step)" if (wcache_on and not args.optimizer) else "every step"),
                   "loss": ("weighted CE in the forward's head epilogue (egx_ce)" if (wl["name"] in ("c1", "c2") and not args.no_fused_ce)
                            else "lossAV in the encoder's launches where the per-clip kernels can (egx_token_ce)" if (wl["name"] == "c3" and not args.no_fused_ce)
                            else "separate launch(es)")},
        "library_launches_per_step": launches_per_step,
        "step_tflops": (fwd_f + bwd_f) / (ms_per_step * 1e-3) / 1e12,
        "step_frac_of_mfma_peak": (fwd_f + bwd_f) / (ms_per_step * 1e-3) / 1e12 / PEAK_TFLOPS[dtype],
    }
    if multi:
        # what took part: the rank count as a REAL collective sees it (sum of ones over the group), its backend, and how much of the
        # exchange the step actually pays for: the same step timed without any collective (same launches, same graph policy)
        t = torch.ones(1, device=dev, dtype=torch.float32)
        if egx_comm is not None:
            egx_comm.allreduce_(t, average=False)
            out["collective_backend"] = "egx_allreduce (RCCL through the C ABI: " + (lib.egx_comm_library() or b"?").decode() + ")"
        else:
            dist.all_reduce(t)
            out["collective_backend"] = dist.get_backend()
        out["rccl_ranks"] = int(round(t.item()))
        model.egx_defer_small = False
        gr0 = capture(fwd_bwd) if use_graph else None
        step_noex = gr0.replay if gr0 is not None else fwd_bwd
        d_ne = time_trials(step_noex, max(3, args.warmup // 2), args.steps, max(3, trials // 2))
        ms_ne = percentile(d_ne, 0.5) / args.steps * 1e3
        out["ms_per_step_without_exchange"] = ms_ne
        out["exposed_collective_us"] = (ms_per_step - ms_ne) * 1e3
    if allreduce_us is not None:
        out["allreduce_us"] = allreduce_us
        out["overlap"] = "staged" if overlap else ("bucketed" if bucketed else "none")
        if exch["collectives"] is not None:
            out["collectives_per_step"] = exch["collectives"]

    if not args.optimizer and not args.no_optimizer_line:
        # "+ optimizer step reported separately" (SURVEY.md 8d): the same step with the Adam update inside
        opt2 = FusedAdam(params, lr=5e-4)
        if wcache_on:
            model.enable_weight_cache(frozen=False)      # the update changes the weights: the packing launch is part of this step
        step2 = make_step(opt2)
        d2 = time_trials(step2, max(3, args.warmup // 2), args.steps, max(3, trials // 2))
        m2 = percentile(d2, 0.5) / args.steps * 1e3
        out["with_optimizer"] = {"optimizer": "FusedAdam(lr=5e-4), one launch over the flat parameter buffer; the weights are re-packed every step",
                                 "ms_per_step": m2, "value": B * world / (m2 * 1e-3), "unit": "clips/s"}
        if wcache_on:
            F_egx.note_weights_changed()                 # (the replayed updates wrote the parameters behind torch's back)
            model.enable_weight_cache(frozen=True)
    if dtype == "f32s" and not multi and not args.no_native_line:
        # the same step on the exact fp32 MFMA path (v_mfma_f32_16x16x4_f32), for reference
        model.set_compute("f32", args.impl)
        step3 = make_step(None)
        d3 = time_trials(step3, max(3, args.warmup // 2), args.steps, max(3, trials // 2))
        m3 = percentile(d3, 0.5) / args.steps * 1e3
        out["native_f32"] = {"arithmetic": "v_mfma_f32_16x16x4_f32 (exact fp32 products)", "ms_per_step": m3,
                             "value": B * world / (m3 * 1e-3), "unit": "clips/s",
                             "step_frac_of_mfma_peak": (fwd_f + bwd_f) / (m3 * 1e-3) / 1e12 / PEAK_TFLOPS["f32"]}
        model.set_compute("f32s", args.impl)
    if wl["name"] in ("c1", "c2") and not multi and not args.no_native_line and hasattr(model, "dp_rate"):
        # two series. The line above is the reference recipe (dropout 0.5 + 0.1 on the positional encoding); this is
        # the deterministic p = 0 configuration the parity tests check, the GPU partner of cpu_baseline.train_p0
        p_keep, pe_keep = model.dp_rate, model.pos_embed.dropout.p
        model.dp_rate, model.pos_embed.dropout.p = 0.0, 0.0
        step0 = make_step(None)
        d0 = time_trials(step0, max(3, args.warmup // 2), args.steps, max(3, trials // 2))
        m0 = percentile(d0, 0.5) / args.steps * 1e3
        out["dropout0"] = {"dropout": 0.0, "ms_per_step": m0, "value": B * world / (m0 * 1e-3), "unit": "clips/s",
                           "note": "same step, train mode, every dropout probability 0 (no mask work in the kernels)"}
        model.dp_rate, model.pos_embed.dropout.p = p_keep, pe_keep
    if rank == 0 and not args.no_roofline:
        model.egx_defer_small = False
        out["roofline"] = measure_roofline(torch, lib, fwd_bwd, wl, dtype)
        # `frac` is the dominant LAUNCH's fraction; `step_frac` the whole step's (all algorithmic FLOPs of forward + backward over the step time):
        # the number that moves only when the step does (cutting a kernel in two raises `frac` without a faster step)
        out["roofline"]["step_frac"] = out["step_frac_of_mfma_peak"]
        out["roofline"]["step_tflops"] = out["step_tflops"]
        if wl.get("producer"):
            out["producer"] = wl["producer"]
    if rank == 0 and not args.no_cpu_baseline and world == 1 and wl["name"] in ("c1", "c2"):
        from oracle.stock_module import time_cpu_baseline
        out["cpu_baseline"] = time_cpu_baseline(B=B, T=args.frames, n_tasks=len(wl["feats"]), dim=128, n_heads=4,
                                                num_layers=wl["L"], dropout=0.5 if args.dropout is None else args.dropout)
    if rank == 0:
        # RCCL prints its version banner through C stdio (NCCL_DEBUG=VERSION is exported on the GPU boxes); push it out
        # first so that the JSON line is the LAST line of stdout
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass
        print(json.dumps(out), flush=True)
    if egx_comm is not None:
        ddp.use_egx_comm(None)
        egx_comm.close()
    if multi:
        dist.barrier()
        dist.destroy_process_group()
    return 0


def csrc_sha():
    """sha256 over the kernel sources (egot2_amd/csrc/*.hip, *.h and include/egot2x.h): committed counter files carry it, and
    the bench line drops their numbers when the kernels have changed since they were taken."""
    import glob
    import hashlib
    here = os.path.dirname(os.path.abspath(__file__))
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(here, "egot2_amd", "csrc", "*.hip")) + glob.glob(os.path.join(here, "egot2_amd", "csrc", "*.h")))
    for f in files + [os.path.join(here, "include", "egot2x.h")]:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def committed_counters(config, dtype, B, T, L, encoder_only=False):
    """Per-kernel hardware counters of this workload from the newest matching profiles/r*_pmc_*.json (tools/pmc_collect.py:
    rocprofv3 --pmc passes, FETCH_SIZE / WRITE_SIZE separately, KiB units, FETCH_SIZE doubled for gfx950 as
    MI355X_MICROARCH.md prescribes). PMC collection needs the profiler, so the numbers are a builder-side measurement read
    from profiles/ — used only when the file was taken on THIS workload and on THESE kernel sources (csrc_sha); otherwise
    (None, reason)."""
    import glob
    here = os.path.dirname(os.path.abspath(__file__))
    sha = csrc_sha()
    stale = None
    for f in sorted(glob.glob(os.path.join(here, "profiles", "r*_pmc_*.json")), reverse=True):
        try:
            d = json.load(open(f))
            w = d.get("workload", {})
            if (w.get("config"), w.get("batch"), w.get("frames"), w.get("layers") or 0, bool(w.get("encoder_only"))) != (config, B, T, L or 0, bool(encoder_only)):
                continue
            wd = w.get("dtype") or ("f32s" if config in ("c1", "c2") else "bf16")      # pmc_collect records the --dtype ARGUMENT (None = the default)
            if wd != dtype or "kernels" not in d:
                continue
            if d.get("csrc_sha") != sha:
                stale = stale or f"{os.path.relpath(f, here)} was taken on other kernel sources (csrc_sha {d.get('csrc_sha')} != {sha}): dropped"
                continue
            return d, os.path.relpath(f, here)
        except (OSError, ValueError, KeyError):
            continue
    return None, stale


def counter_fields(cnt, kernel, avg_launch_us):
    """traffic / hbm_gbps / mfma_busy of one kernel from a committed counter file (None where not collected)."""
    ks = (cnt or {}).get("kernels", {})
    k = ks.get(kernel) or {}
    if not k and kernel.endswith("::ffn_dw_kernel"):      # the library's timer name covers the variant that actually ran
        k = (ks.get("egx::ffn_dw_split2w8_kernel") or ks.get("egx::ffn_dw_split2_kernel") or ks.get("egx::ffn_dw_stored_kernel")
             or ks.get("egx::ffn_dw_bf16_ring_kernel") or {})
    traffic = k.get("traffic_bytes")
    out = {"traffic": traffic,
           "hbm_gbps": (traffic / (avg_launch_us * 1e-6) / 1e9) if traffic and avg_launch_us else None}
    if k.get("SQ_BUSY_CYCLES"):
        # SQ_BUSY_CYCLES is summed over the 32 shader engines, SQ_VALU_MFMA_BUSY_CYCLES over the 1024 SIMDs
        out["mfma_busy"] = (k.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / 1024.0) / (k["SQ_BUSY_CYCLES"] / 32.0)
        out["valu_per_mfma"] = k.get("SQ_INSTS_VALU", 0.0) / max(k.get("SQ_INSTS_MFMA", 0.0), 1.0)
        out["lds_bank_conflict_cycles"] = k.get("SQ_LDS_BANK_CONFLICT")
    else:
        out["mfma_busy"] = None
    return out


TIMED_KERNELS = ("fused_fwd_kernel", "fused_bwd_kernel", "ffn_dw_kernel", "ffn_fwd_kernel", "ffn_bwd_kernel",
                 "wide_gemm_kernel", "wide_attn_fwd_kernel", "wide_attn_bwd_kernel")


def measure_roofline(torch, lib, step, wl, dtype):
    """Roofline block of the dominant kernel of the workload. Average launch durations are measured live with hipEvents
    recorded by the library on the launch stream (egx_timing_*), over eager launches of the same step.

    Fused per-clip path (c1..c3): dominant kernel = fused_bwd_kernel (head, LayerNorm, attention and projection input
    gradients and the FFN input gradient dH = (W2^T g) .* alive, dX1 = W1^T dH). Its ALGORITHMIC FLOPs per launch are the
    dX-type GEMMs of the backward (BASELINE.md §2 accounting; the QKV / probability recompute is NOT counted):
    L * (2N d 3d + 2N d^2 + 4N d d_ff + 8 B S^2 d).
    Wide path (c4, c5): dominant kernel = wide_gemm_kernel; its algorithmic FLOPs are ALL dense-layer GEMMs of the step
    (forward, dX and dW: 3 x forward GEMM FLOPs minus the feature-projection dX) summed over its launches."""
    import ctypes as C
    B, S, d, L, segs = wl["B"], wl["S"], wl["d"], wl["L"], wl["segs"]
    dff = wl.get("dff", 2048)
    N = B * S
    ffn = L * 4.0 * N * d * dff
    proj = sum(2.0 * B * t * k * d for t, k, pj in segs if pj)
    attn_fwd = proj + L * (2.0 * N * d * 3 * d + 4.0 * B * S * S * d + 2.0 * N * d * d)
    flops = {
        "fused_bwd_kernel": L * (2.0 * N * d * 3 * d + 2.0 * N * d * d + 8.0 * B * S * S * d) + ffn,
        "fused_fwd_kernel": attn_fwd + ffn,
        # one launch PER LAYER (until round 5 these carried the whole step's FFN FLOPs against the average LAUNCH: L x too high on C3 / PNR; C2 has L = 1)
        "ffn_dw_kernel": ffn / L,
        "ffn_fwd_kernel": ffn / L,
        "ffn_bwd_kernel": ffn / L,
        "wide_gemm_kernel": 3.0 * (proj + L * (2.0 * N * d * 3 * d + 2.0 * N * d * d) + ffn) - 2.0 * proj + proj,
        "wide_attn_fwd_kernel": L * 4.0 * B * S * S * d,
        "wide_attn_bwd_kernel": L * 8.0 * B * S * S * d,
    }
    lib.egx_timing_enable(1)
    reps = 16 if wl["name"] in ("c1", "c2", "c3", "pnr") else 4
    for _ in range(reps):
        step()
    torch.cuda.synchronize()
    res, tot_ms, counts = {}, {}, {}
    for which, name in enumerate(TIMED_KERNELS):
        tot, cnt = C.c_double(0), C.c_int(0)
        if lib.egx_timing_read(which, C.byref(tot), C.byref(cnt)) == 0 and cnt.value:
            res[name] = tot.value / cnt.value * 1e-3
            tot_ms[name], counts[name] = tot.value, cnt.value
    # tiled mode (S > 48): the tile kernels run L + 1 launches per step and the attention is its own kernels: times become
    # per-STEP sums and the attention FLOPs move to the attention entries
    from egot2_amd import functional as F_egx
    tiled = F_egx.last_encoder_impl() == "tiled"
    if tiled:
        for k in list(res):
            res[k] = tot_ms[k] * 1e-3 / reps
        flops["fused_bwd_kernel"] -= L * 8.0 * B * S * S * d
        flops["fused_fwd_kernel"] -= L * 4.0 * B * S * S * d
        for k in ("ffn_dw_kernel", "ffn_fwd_kernel", "ffn_bwd_kernel"):
            flops[k] = ffn          # per-step sums here: every layer's launch
    # split mode: the FFN halves run as their own launches; the per-clip kernels then hold only the attention halves
    # (cut mode launches the attention-side kernels once per layer as well)
    if "ffn_fwd_kernel" in res:
        flops["fused_fwd_kernel"] = (flops["fused_fwd_kernel"] - ffn) / L
    if "ffn_bwd_kernel" in res:
        flops["fused_bwd_kernel"] = (flops["fused_bwd_kernel"] - ffn) / L
    lib.egx_timing_enable(0)
    peak = PEAK_TFLOPS[dtype]
    cnt, cnt_src = committed_counters(wl["name"], dtype, wl.get("batch_arg", B), wl.get("frames"), wl.get("layers_arg"), wl.get("encoder_only", False))
    if "fused_bwd_kernel" in res:
        # dominant kernel = the longest average launch among the per-clip path's kernels (one-launch kernels: fused_bwd; cut mode: one
        # of the FFN launches or the FFN weight-gradient kernel); the others follow in other_kernels
        dom = max((k for k in res if k in ("fused_bwd_kernel", "fused_fwd_kernel", "ffn_dw_kernel", "ffn_fwd_kernel", "ffn_bwd_kernel")),
                  key=lambda k: res[k])
        if tiled or "ffn_fwd_kernel" not in res:
            dom = "fused_bwd_kernel"
        t = res[dom]
        ach = flops[dom] / t / 1e12
        extra = {}
        if dtype == "f32s":
            extra = {"peak_note": "bf16 dense MFMA peak (2500 TFLOP/s) / 6 instructions per algorithmic K-block; the exact fp32 MFMA peak "
                                  "is 157.3 TFLOP/s", "frac_of_native_f32_mfma_peak": ach / PEAK_TFLOPS["f32"]}
        cf = counter_fields(cnt, "egx::" + dom, t * 1e6)
        step_traffic = None
        if cnt:
            step_traffic = sum(v.get("traffic_bytes", 0.0) * v.get("launches", 0) for k, v in cnt["kernels"].items()
                               if k.startswith("egx::")) / max(cnt.get("steps_profiled", 1), 1)
        if tiled:
            extra["tiled"] = (f"S = {S} > 48: the per-clip kernels run over 48-token tiles, {L + 1} launches per step (times and FLOPs are "
                              "per-step sums over them), attention in tiled_attn kernels (other_kernels: wide_attn_*)")
        if "ffn_fwd_kernel" in res:
            extra["cut_mode"] = ("the per-clip kernels are cut at the FFN (egot2_amd/csrc/ffn_cut.hip): ffn_fwd / ffn_bwd_kernel = the hidden loops with "
                                 "eight waves per clip, fused_fwd / fused_bwd_kernel = the attention-side launches; `kernel` is the longest launch")
        return {"bound": "mfma", "kernel": "egx::" + dom, "achieved": ach, "peak": peak,
                "unit": "TFLOP/s", "frac": ach / peak, **extra, **cf, "traffic_unit": "bytes/launch",
                "counters_source": cnt_src, "hbm_bytes_per_step": step_traffic,
                "flops_per_launch": flops[dom], "avg_launch_us": t * 1e6,
                "other_kernels": {k: {"avg_launch_us": v * 1e6, "achieved_tflops": flops[k] / v / 1e12,
                                      "frac": flops[k] / v / 1e12 / peak, **counter_fields(cnt, "egx::" + k, v * 1e6)}
                                  for k, v in res.items() if k != dom}}
    if "wide_gemm_kernel" in res:
        # many launches of different shapes per step: achieved = all GEMM FLOPs of one step / all GEMM time of one step
        t_step = tot_ms["wide_gemm_kernel"] * 1e-3 / reps
        ach = flops["wide_gemm_kernel"] / t_step / 1e12
        others = {}
        for k in ("wide_attn_fwd_kernel", "wide_attn_bwd_kernel"):
            if k in res:
                ts = tot_ms[k] * 1e-3 / reps
                others[k] = {"us_per_step": ts * 1e6, "launches_per_step": counts[k] / reps,
                             "achieved_tflops": flops[k] / ts / 1e12, "frac": flops[k] / ts / 1e12 / peak}
        traffic = mfma_busy = None
        if cnt:     # the GEMM launches of one step together (NT + TN kernels), per step
            gk = [v for k, v in cnt["kernels"].items() if k.startswith("egx::wide_gemm")]
            steps_p = max(cnt.get("steps_profiled", 1), 1)
            if gk and all("traffic_bytes" in v for v in gk):
                traffic = sum(v["traffic_bytes"] * v["launches"] for v in gk) / steps_p
            if gk and all(v.get("SQ_BUSY_CYCLES") for v in gk):
                mfma_busy = sum(v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) * v["launches"] for v in gk) / 1024.0 / (
                    sum(v["SQ_BUSY_CYCLES"] * v["launches"] for v in gk) / 32.0)
        return {"bound": "mfma", "kernel": "egx::wide_gemm_kernel", "achieved": ach, "peak": peak, "unit": "TFLOP/s",
                "frac": ach / peak, "traffic": traffic, "traffic_unit": "bytes/step (all GEMM launches of one step)",
                "hbm_gbps": (traffic / t_step / 1e9) if traffic else None, "mfma_busy": mfma_busy, "counters_source": cnt_src,
                "flops_per_step": flops["wide_gemm_kernel"],
                "us_per_step": t_step * 1e6, "launches_per_step": counts["wide_gemm_kernel"] / reps,
                "avg_launch_us": res["wide_gemm_kernel"] * 1e6, "other_kernels": others}
    return {"bound": "mfma", "kernel": None, "achieved": None, "peak": peak, "unit": "TFLOP/s", "frac": None, "traffic": None}


if __name__ == "__main__":
    sys.exit(main())
