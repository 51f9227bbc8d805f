#!/usr/bin/env python3
"""Headline benchmark: x-vector embeddings/s on 300-frame x 24-MFCC utterances.

    python bench.py --gpus N --steps K --warmup W

N>1 runs one rank per GPU over RCCL.  Either the caller starts the ranks (`python -m torch.distributed.run
--nproc-per-node N ... bench.py --gpus N ...`: WORLD_SIZE is set) or bench.py does it itself: a plain
`python bench.py --gpus N` starts that same launcher as a CHILD process before anything here touches the GPU,
passes rank 0's JSON line through and exits with the child's code.

One "step" = one pass of the extraction path (XVectorModel.extract_x_vec, layer 6) over one
batch of 256 synthetic utterances already resident in HBM (BASELINE.json configs[1]).
Every rank runs the same per-GPU workload (weak scaling); with N>1 the job ends with the one
real exchange of the path, an RCCL all-gather of the [K*256, 512] fp32 embeddings, inside
the timed region.  Rank 0 prints ONE JSON line.

The other BASELINE configs are parity-test cases, not the bench line; they can still be timed:
    --workload ragged     configs[2]: 256 utterances of 200-1000 frames, padded + lengths mask
    --workload job        configs[3]: 100 000 utterances, sharded over the ranks, one all-gather (strong scaling)
    --dtype bf16          configs[4]
"""
import argparse
import datetime
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# dmabuf IPC: on this driver RCCL (and any sharing of device memory between processes) fails with
# `hipIpcGetMemHandle: invalid argument` without it.  Set HERE -- before torch is imported, before any HIP call -- so that
# BOTH launch forms have it: the ranks `torch.distributed.run ... bench.py` starts directly, and the ones self_launch() starts.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np
import torch

HBM_PEAK = 8.0e12         # B/s, MI355X_MICROARCH.md (spec)
FP32_MFMA_PEAK = 157.3e12  # FLOP/s dense fp32 MFMA (= fp32 vector peak), MI355X_MICROARCH.md
BF16_MFMA_PEAK = 2.5e15    # FLOP/s dense bf16 MFMA, MI355X_MICROARCH.md
BYTES_PER_UTT = 8_238_208  # every layer reads its input once, writes its output once (T=300, fp32);
                           # bf16 activations halve it (4.12 MB/utt)


def layer_flops(T):
    """Algorithmic FLOPs per utterance of the five frame-level layers."""
    return [2 * 120 * 512 * (T - 4), 2 * 1536 * 512 * (T - 8), 2 * 1536 * 512 * (T - 14),
            2 * 512 * 512 * (T - 14), 2 * 512 * 1500 * (T - 14) + 4 * 1500 * (T - 14)]


def usable_cpus():
    """CPU share of this process: the smaller of the affinity mask and the cgroup quota
    (the GPU box shows 256 logical CPUs but grants a 16-CPU quota per GPU)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def traffic_key(dtype, pp16):
    """Key of the dominant kernel (layers 2-4) in profiles/traffic.json: a substring of its demangled symbol, as rocprofv3
    names it.  tests/test_host_logic.py checks every such key against the symbols of the built library."""
    return {"fp32": "tdnn_kernel<0, false, true, false, false, false>",
            "bf16": "pp16::tdnn_pp_kernel<false, false>" if pp16 else "tdnn_kernel<0, false, true, true, true, false>",
            "bf16x3": "pp16::tdnn_pp_kernel<false, true>" if pp16 else "tdnn_kernel<0, false, true, true, true, true>"}[dtype]


def total_flops(T):
    return sum(layer_flops(T)) + 2 * 3000 * 512


def self_launch(n, argv, port=0):
    """`python bench.py --gpus N` without a launcher: start the N ranks with torch.distributed.run as a CHILD of
    this process (which has made no HIP call: a process that has initialised the GPU must never exec or be
    replaced), pass the child's stdout -- rank 0's single JSON line -- through and return its exit code."""
    import socket
    import subprocess
    if not port:
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)                                # (HSA_ENABLE_IPC_MODE_LEGACY=0 is in it: module top)
    # stdout carries exactly the ranks' JSON line(s); anything else a rank or a backend prints there (gloo's
    # connection notes in a dry run) goes to stderr
    with subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, bufsize=1) as child:
        for line in child.stdout:
            (sys.stdout if line.lstrip().startswith("{") else sys.stderr).write(line)
            sys.stdout.flush()
        return child.wait()


def attribution(marks, collective, world):
    """config.per_rank_compute_s / config.all_gather_ms of a --gpus N line: every rank's own extraction time and the time of the
    one collective (which includes waiting for the slowest rank), gathered over the group AFTER the timed region, so that
    a scaling loss can be put on imbalance, on the exchange or on neither.  marks: extract.Marks with three marks."""
    import torch.distributed as dist
    spans = marks.spans()
    mine = {"compute_s": spans[0], "all_gather_s": spans[1] if len(spans) > 1 else 0.0}
    rows = [mine] * world
    if collective:
        rows = [None] * world
        dist.all_gather_object(rows, mine)
    comp = [r["compute_s"] for r in rows]
    gath = [r["all_gather_s"] for r in rows]
    return {"per_rank_compute_s": {"min": round(min(comp), 6), "max": round(max(comp), 6),
                                   "ranks": [round(c, 6) for c in comp]},
            "all_gather_ms": {"min": round(min(gath) * 1e3, 4), "max": round(max(gath) * 1e3, 4)}}


def dry_run(args, world, rank):
    """The N>1 control flow on CPU (gloo): barriers, K steps, the all-gather of [K*B,512] per rank (or the sharded
    job), max-over-ranks timing, one JSON line from rank 0.  No GPU, no library, no measurement."""
    import torch.distributed as dist
    from xvector_amd import extract                     # host-side sharding logic only (no library, no device)
    B, K = args.batch, args.steps
    collective = world > 1 or args.force_collective
    if collective:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=args.dist_timeout))
    if args.fail_rank == rank:        # (tests) a rank that dies before the first barrier must end the job, not hang its peers
        raise RuntimeError(f"bench.py --fail-rank {rank}: simulated rank failure before the first barrier")
    if args.report_env:
        # (tests) what every rank sees of the variable RCCL needs, gathered over the group and printed by rank 0 alone
        vals = [os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")] * world
        if collective:
            dist.all_gather_object(vals, os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY"))
        if rank == 0:
            print("ENV " + json.dumps({"HSA_ENABLE_IPC_MODE_LEGACY": vals}), file=sys.stderr, flush=True)
    fake = lambda x: torch.full((x.shape[0], 512), float(rank))      # noqa: E731
    if collective:
        dist.barrier()
    marks = extract.Marks()
    t0 = time.perf_counter()
    if args.workload == "job":
        full = extract.extract_sharded(fake, lambda lo, hi: torch.zeros((hi - lo, 1, 1)), args.utterances,
                                       batch_size=B, force_collective=args.force_collective, marks=marks)
        assert full.shape == (args.utterances, 512)
        n_done = args.utterances
    else:
        marks.mark()
        emb = torch.cat([fake(torch.zeros((B, 1, 1))) for _ in range(K)])
        marks.mark()
        if collective:
            gathered = torch.empty((world * K * B, 512))
            dist.all_gather_into_tensor(gathered, emb)
            assert all(float(gathered[r * K * B, 0]) == r for r in range(world))
        marks.mark()
        n_done = world * K * B
    if collective:
        dist.barrier()
    dt = time.perf_counter() - t0
    attr = attribution(marks, collective, world)
    if collective:
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if rank == 0:
        print(json.dumps({
            "metric": "x-vector embeddings/sec (300-frame utt)", "value": round(n_done / dt, 1), "unit": "embeddings/s",
            "n_gpus": world, "steps": K, "warmup": args.warmup, "ms_per_step": round(dt / max(K, 1) * 1e3, 4),
            "higher_is_better": True, "scaling": "strong" if args.workload == "job" else "weak", "vs_baseline": None,
            "dtype": "none", "data": "dry-run",
            "config": {"workload": f"DRY RUN of the launch / collective path on CPU (gloo), {args.workload}; no GPU work",
                       "batch_per_gpu": B, "sharding": f"utterance-sharded x{world}", **attr}}), flush=True)
    if collective:
        dist.destroy_process_group()
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--frames", type=int, default=300)
    ap.add_argument("--cpu-budget", type=float, default=40.0,
                    help="seconds of CPU-baseline work over its four legs (B=1/64 x all cores/1 thread; 0 = skip)")
    ap.add_argument("--preroll", type=float, default=0.5,
                    help="seconds the path runs UNTIMED before the W warm-up steps, so that the GPU has left its idle "
                         "power state when the clock starts: the driver's 5+20 steps are 10-70 ms of work, shorter than "
                         "the ramp (same kernels, bf16: 540 k embeddings/s after 5 warm-up steps, 619 k after 1000; "
                         "fp32 +0.9 %).  0 = off.  The timed region is unchanged: exactly K steps between barriers")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the secondary legs of the default line (configs[4] bf16 and configs[2] ragged figures)")
    ap.add_argument("--force-collective", action="store_true",
                    help="initialise the RCCL process group and run the all-gather leg even with one rank "
                         "(one-GPU rehearsal of the N>1 code path)")
    ap.add_argument("--dtype", choices=("fp32", "bf16", "bf16x3"), default="fp32",
                    help="frame-level arithmetic; the headline (BASELINE configs[1]) is fp32 (exact fp32 MFMA).  "
                         "bf16x3: fp32 values as two bf16 planes, three bf16 products per k-step (parity bar 1e-4)")
    ap.add_argument("--workload", choices=("fixed", "ragged", "job", "wave"), default="fixed",
                    help="fixed: configs[1] (the bench line).  ragged: configs[2], one batch of utterances of "
                         "200-1000 frames, zero-padded with a lengths mask.  job: configs[3], --utterances "
                         "fixed-length utterances generated on the device batch by batch, sharded over the ranks, "
                         "one all-gather at the end (--steps is derived).  wave: 3 s waveforms at 16 kHz in HBM -> "
                         "MFCC front end (next row N3) -> the path (299 frames), one step = one batch")
    ap.add_argument("--utterances", type=int, default=100_000, help="job size of --workload job")
    ap.add_argument("--dry-run", action="store_true",
                    help="rehearse the launch / collective / reporting path WITHOUT a GPU (CPU tests of the N>1 path): "
                         "gloo backend, the extraction step replaced by a constant [B,512] tensor, no kernel figures; "
                         "the line says data: \"dry-run\" and its value means nothing")
    ap.add_argument("--master-port", type=int, default=0, help="rendezvous port of the self-launched ranks (0: pick a free one)")
    ap.add_argument("--dist-timeout", type=float, default=180.0,
                    help="seconds a rank waits in the rendezvous or in a collective for its peers before it gives up and exits "
                         "non-zero (torch's default is 10 minutes: a rank that died before the first barrier would hold the "
                         "others, and the driver, that long)")
    ap.add_argument("--fail-rank", type=int, default=-1, help=argparse.SUPPRESS)      # tests: this rank raises before the first barrier
    ap.add_argument("--report-env", action="store_true", help=argparse.SUPPRESS)     # tests (dry run): print the ranks' IPC variable
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus, sys.argv[1:], args.master_port))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        args.gpus = world
    if args.dry_run:
        return dry_run(args, world, rank)
    # stdout carries exactly ONE line, the JSON: whatever a library prints there while the ranks run (RCCL's version
    # banner at communicator creation, for one) goes to stderr; the line itself is written to the saved descriptor
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import torch.distributed as dist
    import xvector_amd as xa
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    collective = world > 1 or args.force_collective
    if collective:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=dev, timeout=datetime.timedelta(seconds=args.dist_timeout))
    if args.fail_rank == rank:
        raise RuntimeError(f"bench.py --fail-rank {rank}: simulated rank failure before the first barrier")

    B, T, K, W = args.batch, args.frames, args.steps, args.warmup
    lengths = None
    n_local = None
    if args.workload == "ragged":
        lens_np = xa.synth.make_lengths(B)                  # default_rng(1234).integers(200, 1001, B)  
        lengths = lens_np.tolist()
        T = int(lens_np.max())
    elif args.workload == "wave":
        T = 299                                             # frames of a 3 s crop (reference dataset.py:124-135)
    elif args.workload == "job":
        lo, hi = xa.extract.shard_bounds(args.utterances, rank, world)
        n_local = hi - lo
        K = -(-(args.utterances // world + (1 if args.utterances % world else 0)) // B)   # batches of the largest shard
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in xa.synth.make_state_dict(seed=42).items()}
    model = xa.XVectorModel(precision=args.dtype)
    model.load_state_dict(sd)
    model = model.to(dev).eval()
    gen = torch.Generator(device=dev).manual_seed(1000 + rank)
    x = torch.randn((B, T, 24), generator=gen, device=dev, dtype=torch.float32)
    if lengths is not None:
        x *= (torch.arange(T, device=dev)[None, :] < torch.tensor(lengths, device=dev)[:, None])[:, :, None]
    # the all-gather leg needs the K batches of embeddings in one contiguous buffer; a single rank keeps the
    # tensor each step returns (the product's deliverable) and makes no extra copy of it
    emb = torch.empty((K * B, 512), device=dev, dtype=torch.float32) if n_local is None and collective else None
    kept = [None] * 4
    gathered = torch.empty((world * K * B, 512), device=dev, dtype=torch.float32) if collective and n_local is None else None

    waves = fe = None
    if args.workload == "wave":
        waves = 0.1 * torch.randn((B, 48000), generator=gen, device=dev, dtype=torch.float32)
        fe = xa.MfccFrontEnd(device=dev)
        assert fe(waves).shape == (B, T, 24)

    def step(k):
        out = model.extract_x_vec(fe(waves)) if waves is not None else model.extract_x_vec(x, lengths=lengths)
        if emb is not None:
            emb[k * B:(k + 1) * B] = out
        else:
            kept[k & 3] = out

    def job():
        # the product's sharded job (extract.extract_sharded): contiguous utterance block per rank,
        # batches generated on the device, one all_gather_into_tensor of the [N/W, 512] shards
        return xa.extract.extract_sharded(
            model.extract_x_vec,
            lambda lo, hi: torch.randn((hi - lo, T, 24), generator=gen, device=dev, dtype=torch.float32),
            args.utterances, batch_size=B, force_collective=args.force_collective, marks=marks)

    def barrier():
        if collective:
            dist.barrier()

    def preroll(fn, seconds=None):
        # untimed: keep the device busy for `seconds` (see --preroll)
        seconds = args.preroll if seconds is None else seconds
        t_end = time.perf_counter() + seconds
        while time.perf_counter() < t_end:
            for _ in range(8):
                fn()
            torch.cuda.synchronize(dev)

    if n_local is None:
        preroll(lambda: step(0))
    for _ in range(W):
        model.extract_x_vec(x, lengths=lengths)
    if collective:   # warm the collective too (communicator setup is not part of a step)
        if n_local is None:
            dist.all_gather_into_tensor(gathered, emb)
        else:
            xa.extract.gather_embeddings(torch.zeros((n_local, 512), device=dev), args.utterances,
                                         force=args.force_collective)
    torch.cuda.synchronize(dev)
    barrier()
    torch.cuda.synchronize(dev)
    marks = xa.extract.Marks(dev)          # events on the launch stream: a rank's own work | the collective (no host sync between)
    t0 = time.perf_counter()
    if n_local is not None:
        full = job()
        assert full.shape == (args.utterances, 512)
    else:
        marks.mark()
        for k in range(K):
            step(k)
        marks.mark()
        if collective:
            dist.all_gather_into_tensor(gathered, emb)
        marks.mark()
    torch.cuda.synchronize(dev)
    barrier()
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    if collective:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    attr = attribution(marks, collective, world)

    # ---- single-GPU extras, reported beside the headline and never as `value`.  Skipped for N>1 (they
    # are per-GPU figures, and a graph capture beside a live RCCL communicator is not worth risking
    # the scaling line for); any failure here leaves the fields null instead of losing the line.
    dt_pcie = dt_pcie_ovl = dt_graph = None
    K_pcie = min(K, 50)

    def extras():
        nonlocal dt_pcie, dt_pcie_ovl, dt_graph
        # ---- boundary hands over host buffers: same steps with the batch copied from pinned host memory
        # and the embeddings delivered to ordinary host memory each step, nothing overlapped (reported
        # beside the headline, never as `value`)
        x_host = x.cpu().pin_memory()
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        for k in range(K_pcie):
            out_host = model.extract_x_vec(x_host.to(dev, non_blocking=True), lengths=lengths).cpu()
        torch.cuda.synchronize(dev)
        dt_pcie = time.perf_counter() - t1
        # the product's pipelined form (extract.stream_x_vectors: next batch's H2D on a side stream)
        if lengths is None:
            # warm-up as long as the timed call: the first ~50 batches a process sends through the three-stream
            # pipeline run at about half speed whatever was done before (0.78 vs 0.43 ms per bf16 batch; an 8-batch
            # warm-up that touched every ring slot did not change that, a 50-batch one does)
            for _ in xa.extract.stream_x_vectors(model, (x_host for _ in range(max(K_pcie, 50)))):
                pass
            torch.cuda.synchronize(dev)
            t2 = time.perf_counter()
            for _ in xa.extract.stream_x_vectors(model, (x_host for _ in range(K_pcie))):
                pass
            torch.cuda.synchronize(dev)
            dt_pcie_ovl = time.perf_counter() - t2

        # ---- the same steps as ONE hipGraph replay each (model.graphed): what the launch gaps cost
        if lengths is None and n_local is None and waves is None:
            gp = model.graphed(x)
            for _ in range(3):
                gp(x)
            torch.cuda.synchronize(dev)
            t3 = time.perf_counter()
            for _ in range(K):
                gp(x)
            torch.cuda.synchronize(dev)
            dt_graph = time.perf_counter() - t3

    if world == 1 and not args.force_collective:
        try:
            extras()
        except Exception as e:      # noqa: BLE001
            print(f"bench.py: extras skipped ({type(e).__name__}: {e})", file=sys.stderr, flush=True)

    # ---- per-kernel durations: hipEvents recorded by the library on the launch stream -----
    names = ("tdnn1", "tdnn2", "tdnn3", "tdnn4", "tdnn5_pool", "pool_finalize", "segment6")

    def per_kernel_ms(mdl, xin, lens, n_iter):
        mdl.set_profiling(True, dev)
        acc = {n: 0.0 for n in names}
        for _ in range(n_iter):
            mdl.extract_x_vec(xin, lengths=lens)
            tm = mdl.timings_ms(dev)       # synchronises on the step's last event
            for n in names:
                acc[n] += tm[n]
        mdl.set_profiling(False, dev)
        return {n: acc[n] / n_iter for n in names}

    avg_ms = per_kernel_ms(model, x, lengths, min(K, 50))

    # ---- secondary legs of the default line (never `value`): the other single-GPU BASELINE configs measured
    # in the same driver run -- configs[4] (the same batch in bf16) and configs[2] (ragged 200-1000 frames)
    secondary = {}

    def secondary_legs():
        K2 = min(K, 50)
        m16 = xa.XVectorModel(precision="bf16")
        m16.load_state_dict(sd)
        m16 = m16.to(dev).eval()
        preroll(lambda: m16.extract_x_vec(x))
        for _ in range(5):
            m16.extract_x_vec(x)
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        for _ in range(K2):
            m16.extract_x_vec(x)
        torch.cuda.synchronize(dev)
        d16 = time.perf_counter() - t1
        ms16 = per_kernel_ms(m16, x, None, min(K2, 20))
        lf16 = [f * B for f in layer_flops(T)]
        dom16 = (lf16[1] + lf16[2] + lf16[3]) / ((ms16["tdnn2"] + ms16["tdnn3"] + ms16["tdnn4"]) * 1e-3)
        secondary.update({
            "bf16_embeddings_per_s": round(K2 * B / d16, 1), "bf16_ms_per_step": round(d16 / K2 * 1e3, 4),
            "bf16_dominant_tflops": round(dom16 / 1e12, 1), "bf16_roofline_frac": round(dom16 / BF16_MFMA_PEAK, 4),
            "bf16_path_flop_frac_of_peak": round(K2 * B / d16 * total_flops(T) / BF16_MFMA_PEAK, 4),
            "bf16_path_hbm_frac_algorithmic": round(K2 * B / d16 * BYTES_PER_UTT * T / 300.0 * 0.5 / HBM_PEAK, 4),
            "bf16_per_kernel_ms": {n: round(v, 4) for n, v in ms16.items()}})
        # bf16x3: fp32 values as two bf16 planes, three bf16 products -- the same 1e-4 bar as the headline
        m3 = xa.XVectorModel(precision="bf16x3")
        m3.load_state_dict(sd)
        m3 = m3.to(dev).eval()
        preroll(lambda: m3.extract_x_vec(x), min(args.preroll, 0.25))
        for _ in range(5):
            m3.extract_x_vec(x)
        torch.cuda.synchronize(dev)
        t3 = time.perf_counter()
        for _ in range(K2):
            m3.extract_x_vec(x)
        torch.cuda.synchronize(dev)
        d3 = time.perf_counter() - t3
        secondary.update({"bf16x3_embeddings_per_s": round(K2 * B / d3, 1), "bf16x3_ms_per_step": round(d3 / K2 * 1e3, 4)})
        del m3
        lens_np = xa.synth.make_lengths(B)
        Tr = int(lens_np.max())
        xr = torch.randn((B, Tr, 24), generator=gen, device=dev, dtype=torch.float32)
        xr *= (torch.arange(Tr, device=dev)[None, :] < torch.tensor(lens_np, device=dev)[:, None])[:, :, None]
        ll = lens_np.tolist()
        K3 = min(K, 20)
        preroll(lambda: model.extract_x_vec(xr, lengths=ll), min(args.preroll, 0.25))
        for _ in range(3):
            model.extract_x_vec(xr, lengths=ll)
        torch.cuda.synchronize(dev)
        t2 = time.perf_counter()
        for _ in range(K3):
            model.extract_x_vec(xr, lengths=ll)
        torch.cuda.synchronize(dev)
        dr = time.perf_counter() - t2
        secondary.update({
            "ragged_utt_per_s": round(K3 * B / dr, 1), "ragged_valid_frames_per_s": round(K3 * int(lens_np.sum()) / dr, 1),
            "ragged_ms_per_step": round(dr / K3 * 1e3, 4), "ragged_valid_frames_per_batch": int(lens_np.sum()),
            "ragged_workload": f"configs[2]: batch={B} utterances of 200-1000 frames (numpy default_rng(1234)), zero-padded "
                               f"to {Tr} with a lengths mask, fp32"})
        # configs[2] x configs[4]: the same ragged batch through the bf16 path
        preroll(lambda: m16.extract_x_vec(xr, lengths=ll), min(args.preroll, 0.25))
        for _ in range(3):
            m16.extract_x_vec(xr, lengths=ll)
        torch.cuda.synchronize(dev)
        t4 = time.perf_counter()
        for _ in range(K3):
            m16.extract_x_vec(xr, lengths=ll)
        torch.cuda.synchronize(dev)
        dr16 = time.perf_counter() - t4
        secondary.update({"ragged_bf16_utt_per_s": round(K3 * B / dr16, 1),
                          "ragged_bf16_valid_frames_per_s": round(K3 * int(lens_np.sum()) / dr16, 1),
                          "ragged_bf16_ms_per_step": round(dr16 / K3 * 1e3, 4)})
        del m16

    def next_row_legs():
        """The rows either side of the path, short timed loops, never `value`: waveform -> MFCC ->
        path in fp32 and bf16, the MFCC kernel alone, and the PLDA score matrix of a VoxCeleb1-test-sized set."""
        K4 = min(K, 20)
        wv = 0.1 * torch.randn((B, 48000), generator=gen, device=dev, dtype=torch.float32)     # 3 s at 16 kHz (dataset.py:124-135)
        fe2 = xa.MfccFrontEnd(device=dev)

        def timed(fn, n, pre=0.15):
            preroll(fn, min(args.preroll, pre))
            for _ in range(3):
                fn()
            torch.cuda.synchronize(dev)
            t = time.perf_counter()
            for _ in range(n):
                fn()
            torch.cuda.synchronize(dev)
            return (time.perf_counter() - t) / n

        secondary["wave_utt_per_s"] = round(B / timed(lambda: model.extract_x_vec(fe2(wv)), K4), 1)
        m16w = xa.XVectorModel(precision="bf16")
        m16w.load_state_dict(sd)
        m16w = m16w.to(dev).eval()
        secondary["wave_bf16_utt_per_s"] = round(B / timed(lambda: m16w.extract_x_vec(fe2(wv)), K4), 1)
        # ... and with the boundary handing over HOST waveforms (the reference's producer is host-side, dataset.py:124-128):
        # pinned fp32 [B, 48000] batches through the product's overlapped pipeline (extract.stream_x_vectors: H2D on a side
        # stream, MFCC + path on the compute stream, results to pinned host buffers); 49 MB per batch over PCIe
        wv_host = wv.cpu().pin_memory()
        n_w = 30
        for _ in xa.extract.stream_x_vectors(model, (wv_host for _ in range(n_w)), prepare=fe2):      # warm the rings (see extras())
            pass
        torch.cuda.synchronize(dev)
        t_w = time.perf_counter()
        for _ in xa.extract.stream_x_vectors(model, (wv_host for _ in range(n_w)), prepare=fe2):
            pass
        torch.cuda.synchronize(dev)
        secondary["wave_pcie_inclusive_utt_per_s"] = round(n_w * B / (time.perf_counter() - t_w), 1)
        # the bf16 path behind the same boundary: fp32 samples (49 MB per batch: the link, not the GPU, sets the rate) and
        # 16-bit PCM as scipy.io.wavfile.read yields it (dataset.py:125; xvec_mfcc_i16 converts in the kernel: half the bytes)
        pcm_host = (wv * 3.0e4).clamp(-32768, 32767).to(torch.int16).cpu().pin_memory()
        for key, host in (("wave_bf16_pcie_inclusive_f32_samples_utt_per_s", wv_host), ("wave_bf16_pcie_inclusive_utt_per_s", pcm_host)):
            n_b = 60
            for _ in xa.extract.stream_x_vectors(m16w, (host for _ in range(n_b)), prepare=fe2):
                pass
            torch.cuda.synchronize(dev)
            t_b = time.perf_counter()
            for _ in xa.extract.stream_x_vectors(m16w, (host for _ in range(n_b)), prepare=fe2):
                pass
            torch.cuda.synchronize(dev)
            secondary[key] = round(n_b * B / (time.perf_counter() - t_b), 1)
        secondary["wave_bf16_pcie_inclusive_input"] = "int16 PCM [B, 48000] in pinned host memory (24.6 MB per batch), scale 1"
        pcm_dev = pcm_host.to(dev)
        secondary["mfcc_i16_us_per_batch"] = round(timed(lambda: fe2(pcm_dev), 50, 0.05) * 1e6, 2)
        del m16w
        secondary["mfcc_us_per_batch"] = round(timed(lambda: fe2(wv), 50, 0.05) * 1e6, 2)
        secondary["mfcc_algorithmic_gb_per_s"] = round(B * (48000 * 4 + 299 * 24 * 4) / (secondary["mfcc_us_per_batch"] * 1e-6) / 1e9, 1)
        # which kernel that was (xvec_mfcc_kernel_form): 2 = nfft-512 kernel with the banded filterbank, 1 = its dense form, 0 = general
        secondary["mfcc_kernel_form"] = fe2.kernel_form()
        n_sc = 4874                                                    # VoxCeleb1 test set (plda_score_stat.py:19-20: every x-vector against every other)
        mean, F, Sigma = xa.synth.make_plda(512, 200, seed=21)
        scorer = xa.scoring.PldaScorer(mean, F, Sigma, device=dev)
        xs = torch.randn((n_sc, 512), generator=gen, device=dev, dtype=torch.float32).double() + torch.from_numpy(mean).to(dev)
        secondary["plda_score_ms_n4874"] = round(timed(lambda: scorer.score(xs), 10, 0.05) * 1e3, 4)
        secondary["plda_score_form"] = (f"self-score of {n_sc} x-vectors, low-rank form (rank {scorer.rank} of 512): y = (x - mean) L, "
                                        "[y W | row dots], upper triangle of y W y' mirrored")
        dense = xa.scoring.PldaScorer(mean, F, Sigma, device=dev, lowrank=False)
        secondary["plda_dense_score_ms_n4874"] = round(timed(lambda: dense.score(xs), 10, 0.05) * 1e3, 4)
        # fp64 rate of the dense form's [n, n] launch work actually done: upper triangle of tiles + the [n, 1024] prelude product
        secondary["plda_dense_tflops_f64"] = round((n_sc * n_sc * 512.0 + 2.0 * n_sc * 1024 * 512) / (secondary["plda_dense_score_ms_n4874"] * 1e-3) / 1e12, 1)

    def job_leg():
        """configs[3] at its own size on this one GPU (the N = 1 anchor of the scaling curve): 100 000 utterances generated
        on the device batch by batch through the product's sharded job (extract.extract_sharded, one rank: no exchange)."""
        n_job = 100_000
        t_j = time.perf_counter()
        full = xa.extract.extract_sharded(
            model.extract_x_vec, lambda lo, hi: torch.randn((hi - lo, T, 24), generator=gen, device=dev, dtype=torch.float32),
            n_job, batch_size=B)
        torch.cuda.synchronize(dev)
        assert full.shape == (n_job, 512)
        secondary["job100k_embeddings_per_s"] = round(n_job / (time.perf_counter() - t_j), 1)
        # configs[3] x configs[4]: the same job through the bf16 path
        m16j = xa.XVectorModel(precision="bf16")
        m16j.load_state_dict(sd)
        m16j = m16j.to(dev).eval()
        preroll(lambda: m16j.extract_x_vec(x), min(args.preroll, 0.25))
        torch.cuda.synchronize(dev)
        t_j = time.perf_counter()
        full = xa.extract.extract_sharded(
            m16j.extract_x_vec, lambda lo, hi: torch.randn((hi - lo, T, 24), generator=gen, device=dev, dtype=torch.float32),
            n_job, batch_size=B)
        torch.cuda.synchronize(dev)
        assert full.shape == (n_job, 512)
        secondary["job100k_bf16_embeddings_per_s"] = round(n_job / (time.perf_counter() - t_j), 1)

    if (world == 1 and not args.force_collective and not args.no_secondary and args.workload == "fixed"
            and args.dtype == "fp32"):
        for leg in (secondary_legs, next_row_legs, job_leg):
            try:
                leg()
            except Exception as e:      # noqa: BLE001
                print(f"bench.py: {leg.__name__} skipped ({type(e).__name__}: {e})", file=sys.stderr, flush=True)

    if rank == 0:
        act_bytes_scale = 0.5 if args.dtype == "bf16" else 1.0     # bf16x3 moves two bf16 planes = fp32 bytes
        if lengths is None:
            lf = [f * B for f in layer_flops(T)]
            n_done = world * K * B if n_local is None else args.utterances
            frames_done = n_done * T
            path_flops, path_bytes = total_flops(T), BYTES_PER_UTT * T / 300.0 * act_bytes_scale
        else:                                       # per-utterance lengths: sum the per-layer FLOPs
            per = [layer_flops(t) for t in lengths]
            lf = [sum(p[i] for p in per) for i in range(5)]
            n_done = world * K * B
            frames_done = world * K * sum(lengths)
            path_flops = (sum(lf) + B * 2 * 3000 * 512) / B
            path_bytes = BYTES_PER_UTT * (sum(lengths) / B) / 300.0 * act_bytes_scale
        tdnn_names = names[:5]
        tdnn_ms = sum(avg_ms[n] for n in tdnn_names)
        # dominant kernel: the plain tdnn_f32_kernel instance (layers 2-4 launch the same code object)
        dom_ms = (avg_ms["tdnn2"] + avg_ms["tdnn3"] + avg_ms["tdnn4"]) / 3
        dom_flops = (lf[1] + lf[2] + lf[3]) / 3
        achieved = dom_flops / (dom_ms * 1e-3) / 1e12
        bf = args.dtype in ("bf16", "bf16x3")
        # bf16x3 spends three bf16 MFMAs per algorithmic product: its roof is a third of the bf16 peak
        peak = {"fp32": FP32_MFMA_PEAK, "bf16": BF16_MFMA_PEAK, "bf16x3": BF16_MFMA_PEAK / 3}[args.dtype]
        # which kernel layers 2-4 actually went to, as the library reports it (xvec_get_dispatch): bf16 at this batch
        # and bf16x3 at this batch size run the 256-channel ping-pong mapping (csrc/tdnn_pp16.hip); smaller batches, other
        # CU counts and XVEC_PP=0 the 128x128 kernel (csrc/tdnn_layer.hip)
        disp = model.last_dispatch(dev)
        pp16 = disp[1:4] == ["pp", "pp", "pp"]
        dom_kernel = (f"xvec::pp16::tdnn_pp_kernel<false, {'true' if args.dtype == 'bf16x3' else 'false'}> (layers 2-4, "
                      "v_mfma_f32_16x16x32_bf16, LDS-DMA operands"
                      + (", three K-tiles per 64-channel slab)" if args.dtype == "bf16x3" else ")") if pp16 else
                      "xvec::tdnn_kernel<0,false,true,true,true,X3> (layers 2-4, bf16 MFMA"
                      + (", three products per k-step)" if args.dtype == "bf16x3" else ")") if bf else
                      "xvec::tdnn_kernel<0,false,true,false,false,false> (layers 2-4, fp32 MFMA)")
        value = n_done / dt
        # HBM bytes per launch of the dominant kernel come from the committed rocprofv3 --pmc pass of
        # this same command (profiles/traffic.json, written by profiles/summarize_pmc.py); counters
        # cannot be collected from inside the benchmark process.
        traffic, traffic_src, traffic_err = None, None, None
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
            tj = tj[args.dtype]                             # one section per arithmetic
            key = traffic_key(args.dtype, pp16)
            traffic, traffic_src = tj[key]["hbm_bytes_per_launch"], tj["source"]
        except (OSError, KeyError, ValueError) as e:        # said out loud: a renamed or re-tiled kernel must not report stale bytes silently
            traffic_err = f"profiles/traffic.json has no entry for this run's dominant kernel ({type(e).__name__}: {e}); re-run profiles/run_round.sh"
        out = {
            "metric": "x-vector embeddings/sec (300-frame utt)", "value": round(value, 1), "unit": "embeddings/s",
            "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": round(dt / K * 1e3, 4),
            "higher_is_better": True, "scaling": "strong" if n_local is not None else "weak", "vs_baseline": None, "dtype": {"fp32": "f32", "bf16": "bf16", "bf16x3": "bf16x3"}[args.dtype], "data": "synthetic",
            "config": {"workload": {
                           "fixed": f"configs[{4 if args.dtype == 'bf16' else 1}]: 1xMI355X batch={B} fixed {T}-frame x 24-MFCC "
                                    f"utterances, {args.dtype} frame-level stack, ",
                           "ragged": f"configs[2]: batch={B} utterances of 200-1000 frames (numpy default_rng(1234)), zero-padded "
                                     f"to {T} with a lengths mask, {args.dtype}, ",
                           "wave": f"next row N3 + path: batch={B} waveforms of 48000 samples (3 s, 16 kHz) in HBM -> MFCC "
                                   f"kernel -> {T} frames x 24 -> {args.dtype} frame-level stack, ",
                           "job": f"configs[3]: {args.utterances} utterances of {T} frames generated on the device batch by "
                                  f"batch ({B}), sharded over {world} rank(s), {args.dtype}, "}[args.workload]
                                   + "extract_x_vec layer 6, random-init weights seed 42",
                       "build": xa.hip.version(),
                       "batch_per_gpu": B, "frames": T, "valid_frames_per_s": round(frames_done / dt, 1),
                       "pcie_inclusive_embeddings_per_s_per_gpu": round(K_pcie * B / dt_pcie, 1) if dt_pcie else None,
                       "graph_replay_embeddings_per_s_per_gpu": round(K * B / dt_graph, 1) if dt_graph else None,
                       "pcie_inclusive_overlapped_embeddings_per_s_per_gpu":
                           round(K_pcie * B / dt_pcie_ovl, 1) if dt_pcie_ovl else None,
                       "preroll_s": args.preroll if n_local is None else 0.0,
                       "sharding": f"utterance-sharded x{world}"
                       + (", one all-gather of [K*B,512] fp32 in the timed region" if collective else ""),
                       **attr, **secondary},
            "roofline": {
                "bound": "mfma", "kernel": dom_kernel,
                "achieved": round(achieved, 2), "peak": peak / 1e12, "unit": "TFLOP/s",
                "frac": round(achieved * 1e12 / peak, 4), "traffic": traffic, "traffic_source": traffic_src,
                **({"traffic_error": traffic_err} if traffic_err else {}),
                "avg_launch_ms": round(dom_ms, 4), "flops_per_launch": dom_flops,
                "per_kernel_ms": {n: round(v, 4) for n, v in avg_ms.items()},
                "per_kernel_tflops": {n: round(lf[i] / (avg_ms[n] * 1e-3) / 1e12, 2) for i, n in enumerate(tdnn_names)},
                "tdnn_stack_tflops": round(sum(lf) / (tdnn_ms * 1e-3) / 1e12, 2),
                "path_flop_frac_of_peak": round(value / world * path_flops / peak, 4),
                "path_hbm_frac_algorithmic": round(value / world * path_bytes / HBM_PEAK, 4),
            },
        }
        if world == 1 and args.cpu_budget > 0:
            sys.path.insert(0, os.path.join(ROOT, "oracle"))
            import xvector_oracle as oracle
            p = {k: v for k, v in sd.items() if v.is_floating_point()}
            legs = oracle.time_cpu_baseline(p, T=T, budget_s=args.cpu_budget, threads=usable_cpus())
            best = max(legs, key=lambda k: legs[k]["embeddings_per_s"])      # the fastest honest CPU leg is the headline
            head = legs[best]
            secs = sum(l["seconds"] for l in legs.values())
            out["cpu_baseline"] = {
                "value": head["embeddings_per_s"], "unit": "embeddings/s", "cores": head["threads"], "kind": "port",
                "leg": best,
                "sample": f"{T}-frame utterances, fp32, oracle/xvector_oracle.py (PyTorch CPU restatement of the "
                          f"reference's op sequence), SURVEY 8(d) protocol: B=64 and B=1, {head['threads']} threads and 1 "
                          f"thread, 3 warm-ups + median of 10 per leg (a leg that would not fit its share of the "
                          f"{args.cpu_budget:.0f} s budget is cut to 1 warm-up + >=3 timed passes: see legs.*.reps); "
                          f"value = the fastest of the four legs ({best}: B={head.get('batch', '?')}, {head['threads']} "
                          f"thread(s)); {secs:.1f} s of CPU work in all",
                "legs": legs}
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if collective:
        dist.destroy_process_group()


if __name__ == "__main__":
    # a rank that fails exits non-zero (torch.distributed.run then ends its peers and fails the job; self_launch() passes
    # that code on); never a re-exec -- this process may have touched the GPU
    sys.exit(main() or 0)
