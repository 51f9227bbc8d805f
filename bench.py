#!/usr/bin/env python3
"""Headline benchmark: x-vector embeddings/s on 300-frame x 24-MFCC utterances.

    python bench.py --gpus N --steps K --warmup W        (N>1: launched by torch.distributed.run)

One "step" = one pass of the extraction path (XVectorModel.extract_x_vec, layer 6) over one
batch of 256 synthetic utterances already resident in HBM (BASELINE.json configs[1]).
Every rank runs the same per-GPU workload (weak scaling); with N>1 the job ends with the one
real exchange of the path, an RCCL all-gather of the [K*256, 512] fp32 embeddings, inside
the timed region.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch

HBM_PEAK = 8.0e12         # B/s, MI355X_MICROARCH.md (spec)
FP32_MFMA_PEAK = 157.3e12  # FLOP/s dense fp32 MFMA (= fp32 vector peak), MI355X_MICROARCH.md
BF16_MFMA_PEAK = 2.5e15    # FLOP/s dense bf16 MFMA, MI355X_MICROARCH.md
BYTES_PER_UTT = 8_238_208  # every layer reads its input once, writes its output once (T=300, fp32)


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


def total_flops(T):
    return sum(layer_flops(T)) + 2 * 3000 * 512


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--frames", type=int, default=300)
    ap.add_argument("--cpu-budget", type=float, default=15.0, help="seconds of CPU-baseline work (0 = skip)")
    ap.add_argument("--dtype", choices=("fp32", "bf16"), default="fp32",
                    help="frame-level arithmetic; the headline (BASELINE configs[1]) is fp32")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with python -m torch.distributed.run --nproc-per-node N")
        args.gpus = world

    import torch.distributed as dist
    import xvector_amd as xa
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    B, T, K, W = args.batch, args.frames, args.steps, args.warmup
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in xa.synth.make_state_dict(seed=42).items()}
    model = xa.XVectorModel(precision=args.dtype)
    model.load_state_dict(sd)
    model = model.to(dev).eval()
    gen = torch.Generator(device=dev).manual_seed(1000 + rank)
    x = torch.randn((B, T, 24), generator=gen, device=dev, dtype=torch.float32)
    emb = torch.empty((K * B, 512), device=dev, dtype=torch.float32)
    gathered = torch.empty((world * K * B, 512), device=dev, dtype=torch.float32) if world > 1 else None

    def barrier():
        if world > 1:
            dist.barrier()

    for _ in range(W):
        model.extract_x_vec(x)
    if world > 1:   # warm the collective too (communicator setup is not part of a step)
        dist.all_gather_into_tensor(gathered, emb)
    torch.cuda.synchronize(dev)
    barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for k in range(K):
        emb[k * B:(k + 1) * B] = model.extract_x_vec(x)
    if world > 1:
        dist.all_gather_into_tensor(gathered, emb)
    torch.cuda.synchronize(dev)
    barrier()
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # ---- boundary hands over host buffers: same steps with a pinned H2D of x and a D2H of the
    # embeddings per step, not overlapped (reported beside the headline, never as `value`)
    x_host = x.cpu().pin_memory()
    out_host = torch.empty((B, 512), dtype=torch.float32).pin_memory()
    torch.cuda.synchronize(dev)
    t1 = time.perf_counter()
    for k in range(K):
        out_host.copy_(model.extract_x_vec(x_host.to(dev, non_blocking=True)), non_blocking=True)
    torch.cuda.synchronize(dev)
    dt_pcie = time.perf_counter() - t1

    # ---- per-kernel durations: hipEvents recorded by the library on the launch stream -----
    model.set_profiling(True, dev)
    names = ("tdnn1", "tdnn2", "tdnn3", "tdnn4", "tdnn5_pool", "pool_finalize", "segment6")
    acc = {n: 0.0 for n in names}
    for _ in range(K):
        model.extract_x_vec(x)
        tm = model.timings_ms(dev)       # synchronises on the step's last event
        for n in names:
            acc[n] += tm[n]
    model.set_profiling(False, dev)
    avg_ms = {n: acc[n] / K for n in names}

    if rank == 0:
        lf = [f * B for f in layer_flops(T)]
        tdnn_names = names[:5]
        tdnn_ms = sum(avg_ms[n] for n in tdnn_names)
        # dominant kernel: the plain tdnn_f32_kernel instance (layers 2-4 launch the same code object)
        dom_ms = (avg_ms["tdnn2"] + avg_ms["tdnn3"] + avg_ms["tdnn4"]) / 3
        dom_flops = (lf[1] + lf[2] + lf[3]) / 3
        achieved = dom_flops / (dom_ms * 1e-3) / 1e12
        bf = args.dtype == "bf16"
        peak = BF16_MFMA_PEAK if bf else FP32_MFMA_PEAK
        dom_kernel = ("xvec::tdnn_kernel<false,false,true,true,true> (layers 2-4, bf16 MFMA)" if bf else
                      "xvec::tdnn_kernel<false,false,true,false,false> (layers 2-4, fp32 MFMA)")
        value = world * K * B / dt
        # HBM bytes per launch of the dominant kernel come from the committed rocprofv3 --pmc pass of
        # this same command (profiles/traffic.json, written by profiles/summarize_pmc.py); counters
        # cannot be collected from inside the benchmark process.
        traffic, traffic_src = None, None
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
            key = "tdnn_kernel<false, false, true, true, true>" if args.dtype == "bf16" else \
                "tdnn_kernel<false, false, true, false, false>"
            traffic, traffic_src = tj[key]["hbm_bytes_per_launch"], tj["source"]
        except (OSError, KeyError, ValueError):
            pass
        out = {
            "metric": "x-vector embeddings/sec (300-frame utt)", "value": round(value, 1), "unit": "embeddings/s",
            "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": round(dt / K * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32" if args.dtype == "fp32" else "bf16", "data": "synthetic",
            "config": {"workload": f"configs[{1 if args.dtype == 'fp32' else 4}]: 1xMI355X batch={B} fixed {T}-frame x 24-MFCC "
                                   f"utterances, {args.dtype} frame-level stack, "
                                   "extract_x_vec layer 6, random-init weights seed 42",
                       "batch_per_gpu": B, "frames": T,
                       "pcie_inclusive_embeddings_per_s_per_gpu": round(K * B / dt_pcie, 1),
                       "sharding": f"utterance-sharded x{world}"
                       + (", one all-gather of [K*B,512] fp32 in the timed region" if world > 1 else "")},
            "roofline": {
                "bound": "mfma", "kernel": dom_kernel,
                "achieved": round(achieved, 2), "peak": peak / 1e12, "unit": "TFLOP/s",
                "frac": round(achieved * 1e12 / peak, 4), "traffic": traffic, "traffic_source": traffic_src,
                "avg_launch_ms": round(dom_ms, 4), "flops_per_launch": dom_flops,
                "per_kernel_ms": {n: round(v, 4) for n, v in avg_ms.items()},
                "per_kernel_tflops": {n: round(lf[i] / (avg_ms[n] * 1e-3) / 1e12, 2) for i, n in enumerate(tdnn_names)},
                "tdnn_stack_tflops": round(sum(lf) / (tdnn_ms * 1e-3) / 1e12, 2),
                "path_flop_frac_of_peak": round(value / world * total_flops(T) / peak, 4),
                "path_hbm_frac_algorithmic": round(value / world * BYTES_PER_UTT / HBM_PEAK, 4),
            },
        }
        if world == 1 and args.cpu_budget > 0:
            sys.path.insert(0, os.path.join(ROOT, "oracle"))
            import xvector_oracle as oracle
            p = {k: v for k, v in sd.items() if v.is_floating_point()}
            eps, threads, n_utts, secs = oracle.time_cpu_baseline(p, T=T, batch=64, budget_s=args.cpu_budget,
                                                                  threads=usable_cpus())
            out["cpu_baseline"] = {"value": round(eps, 1), "unit": "embeddings/s", "cores": threads, "kind": "port",
                                   "sample": f"{n_utts} utterances of {T} frames in batches of 64, fp32, "
                                             f"{secs:.1f} s of oracle/xvector_oracle.py (PyTorch CPU restatement of "
                                             "the reference's op sequence)"}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
