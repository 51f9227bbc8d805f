"""Callers on either side of the hot path: utterance-sharded extraction over the GPUs of one
node, and the reference's x-vector record / CSV format.

Reference behaviour mirrored here (not its code):
  * main.py:135-146  test_step/test_epoch_end: each row of the [B,512] fp32 result becomes
    (id: str, label: int, vec: float64[512]) appended in input order;
  * main.py:246-247  pd.DataFrame(x_vector).to_csv(path)  -> columns ",0,1,2";
  * main.py:276-279  reader: np.array(s[1:-1].split(), dtype=float64) on column 3.

Multi-GPU: the path shards by utterance with no cross-GPU dependency, so
rank r of W takes the contiguous block [r*ceil(N/W), ...) and the only collective is one
all-gather of the [N/W, 512] fp32 embeddings (RCCL over xGMI when the backend is "nccl").
"""
from __future__ import annotations

from typing import Callable, Iterable, List, Sequence, Tuple

import numpy as np
import torch


# --------------------------------------------------------------------------- sharding
def shard_bounds(n_total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous utterance block of `rank`: ceil(n/world) utterances each, the tail ranks
    get the remainder (possibly none)."""
    per = -(-n_total // world)
    lo = min(rank * per, n_total)
    return lo, min(lo + per, n_total)


def balanced_order(lengths: Sequence[int], world: int) -> List[List[int]]:
    """Variable-length batches: assign utterances to ranks greedily by total frames
    (longest first), so every GPU gets about the same number of frames.  Returns the
    utterance indices per rank; gather_embeddings(order=...) undoes the permutation."""
    loads = [0] * world
    buckets: List[List[int]] = [[] for _ in range(world)]
    for i in sorted(range(len(lengths)), key=lambda j: -int(lengths[j])):
        r = loads.index(min(loads))
        buckets[r].append(i)
        loads[r] += int(lengths[i])
    return [sorted(b) for b in buckets]


class Marks:
    """Time marks of one rank for attributing a multi-GPU line (bench.py `config.per_rank_compute_s`, `config.all_gather_ms`):
    events on the device's current stream (no host synchronisation between the marks, so measuring does not change what is
    measured) or, on the CPU (gloo rehearsals), the host clock.  spans() -> seconds between consecutive marks, to be called
    after the stream has been synchronised."""

    def __init__(self, device=None):
        self.cuda = device is not None and torch.device(device).type == "cuda"
        self.device = device
        self.marks = []

    def mark(self):
        if self.cuda:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record(torch.cuda.current_stream(self.device))
            self.marks.append(ev)
        else:
            import time
            self.marks.append(time.perf_counter())

    def spans(self):
        if self.cuda:
            return [a.elapsed_time(b) * 1e-3 for a, b in zip(self.marks, self.marks[1:])]
        return [b - a for a, b in zip(self.marks, self.marks[1:])]


def gather_embeddings(local: torch.Tensor, n_total: int, group=None,
                      order: Sequence[Sequence[int]] = None, force: bool = False) -> torch.Tensor:
    """All-gather the per-rank [n_local, D] embeddings into [n_total, D] on every rank.

    One all_gather_into_tensor of equal-size (padded) shards; rows are then trimmed back
    (contiguous-block sharding) or scattered to their original positions (`order`).
    A single rank needs no exchange and returns `local` (re-ordered when `order` is given);
    `force=True` runs the collective and the assembly even then (one-GPU rehearsal of the
    multi-GPU code path; needs an initialised process group)."""
    import torch.distributed as dist
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1 and not force:
        if order is None:
            return local
        out = torch.empty((n_total, local.shape[1]), dtype=local.dtype, device=local.device)
        out[torch.as_tensor(list(order[0]), dtype=torch.long, device=local.device)] = local
        return out
    D = local.shape[1]
    if order is None:
        counts = [shard_bounds(n_total, r, world)[1] - shard_bounds(n_total, r, world)[0] for r in range(world)]
    else:
        counts = [len(o) for o in order]
    per = max(counts)
    send = local
    if local.shape[0] != per:
        send = torch.zeros((per, D), dtype=local.dtype, device=local.device)
        send[: local.shape[0]] = local
    recv = torch.empty((world * per, D), dtype=local.dtype, device=local.device)
    if dist.is_initialized():
        dist.all_gather_into_tensor(recv, send.contiguous(), group=group)
    else:                                   # force=True without a process group: world is 1
        recv.copy_(send)
    if order is None:
        if all(c == per for c in counts):
            return recv
        return torch.cat([recv[r * per: r * per + counts[r]] for r in range(world)], 0)
    out = torch.empty((n_total, D), dtype=local.dtype, device=local.device)
    for r in range(world):
        if counts[r]:
            idx = torch.as_tensor(list(order[r]), dtype=torch.long, device=local.device)
            out[idx] = recv[r * per: r * per + counts[r]]
    return out


def extract_sharded(extract_fn: Callable[[torch.Tensor], torch.Tensor], make_batch: Callable[[int, int], torch.Tensor],
                    n_total: int, batch_size: int = 256, group=None, force_collective: bool = False,
                    embed_dim: int = None, device=None, marks: "Marks" = None) -> torch.Tensor:
    """Utterance-sharded extraction job (BASELINE configs[3]).

    make_batch(lo, hi) returns the device tensor [hi-lo, T, C] for global utterances
    lo..hi-1 (a loader, or an on-device generator in the benchmark); extract_fn is
    model.extract_x_vec.  Every rank returns the full [n_total, D] matrix.

    A rank whose block is empty (n_total < world * ceil(n_total / world)) extracts nothing: it joins the
    all-gather with a [0, D] shard, D = `embed_dim` or, when that is not given, learnt from its peers by one
    scalar all-reduce (every rank takes part in it, and only when the last rank's block is empty).  Such a rank has no
    tensor to take its device from: pass `device`, or -- with the "nccl" backend -- have torch.cuda.set_device(local_rank)
    called before (bench.py does), since the fallback is the process's CURRENT device.

    `marks` (a Marks): three marks -- start, this rank's own extraction done, exchange done -- so that a scaling line can say
    which part of a rank's time was its own work and which the collective (including the wait for slower ranks)."""
    import torch.distributed as dist
    if n_total < 1:                        # on EVERY rank, before any collective: nobody is left waiting in one
        raise ValueError("extract_sharded: nothing to extract (n_total < 1)")
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    lo, hi = shard_bounds(n_total, rank, world)
    parts = []
    if marks is not None:
        marks.mark()
    for b0 in range(lo, hi, batch_size):
        b1 = min(b0 + batch_size, hi)
        parts.append(extract_fn(make_batch(b0, b1)))
    local = torch.cat(parts, 0) if parts else None
    if marks is not None:
        marks.mark()
    last_lo, last_hi = shard_bounds(n_total, world - 1, world)
    if world > 1 and last_hi <= last_lo:                       # some rank has no utterances: same answer on every rank
        if device is None:
            device = local.device if local is not None else (
                torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu"))
        D = embed_dim
        if D is None:
            t = torch.tensor([local.shape[1] if local is not None else 0], dtype=torch.int64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
            D = int(t.item())
        if local is None:
            local = torch.zeros((0, D), dtype=torch.float32, device=device)
    full = gather_embeddings(local, n_total, group, force=force_collective)
    if marks is not None:
        marks.mark()
    return full


def extract_balanced(extract_ragged: Callable[[List[int]], torch.Tensor], lengths: Sequence[int], group=None,
                     force_collective: bool = False) -> torch.Tensor:
    """Variable-length job (BASELINE config 3 sharded): utterances are dealt to the ranks by total
    frames (`balanced_order`), every rank extracts its own list with
    `extract_ragged(indices) -> [len(indices), D]` (e.g. a padded batch + `lengths=` call of
    model.extract_x_vec), and one all-gather + un-permute returns [len(lengths), D] in input
    order on every rank."""
    import torch.distributed as dist
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    order = balanced_order(lengths, world)
    local = extract_ragged(order[rank])
    return gather_embeddings(local, len(lengths), group, order=order, force=force_collective)


# --------------------------------------------------------------------------- records / CSV
def to_records(x_vecs: torch.Tensor, labels, ids) -> List[tuple]:
    """[B,D] fp32 device tensor -> [(id, int(label), float64[D])] in input order with ONE
    device-to-host copy (the reference does B of them, main.py:144-145).  The fp32->float64
    widening is exact, so records equal the reference's bit for bit given equal x_vecs."""
    host = x_vecs.detach().to("cpu", torch.float32).numpy()
    if torch.is_tensor(labels):
        labels = labels.detach().cpu().numpy()
    return [(i, int(l), np.array(v, dtype=np.float64)) for v, l, i in zip(host, labels, ids)]


_SIDE = {}


def _side_streams(dev):
    """The copy streams of a device, created once: every torch stream has its own block pool, so a fresh pair
    per call would send the input ring to hipMalloc (which synchronises the device) call after call."""
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    if key not in _SIDE:
        _SIDE[key] = (torch.cuda.Stream(dev), torch.cuda.Stream(dev))
    return _SIDE[key]


def stream_x_vectors(model, host_batches: Iterable[torch.Tensor], device=None, depth: int = 3,
                     prepare: Callable[[torch.Tensor], torch.Tensor] = None):
    """Overlapped extraction from HOST batches: yields the fp32 [B, D] x-vectors of every batch as
    host tensors, in input order.

    Three streams, nothing in the loop blocks the host unless `depth` batches are already in flight:
      * batch k+1 is copied host->device on a side stream while batch k runs on the caller's current
        stream (device input slots: a ring of depth+1 buffers per shape, allocated once -- an allocation
        inside the loop could reach hipMalloc, which synchronises the device);
      * the result of batch k is copied device->host on a third stream, enqueued right behind the
        batch's `done` event into a ring of page-locked result buffers (non-blocking), and handed out
        `depth` batches later -- or earlier, as soon as its copy is known to have landed
        (event.query(), no wait) -- as an ordinary (pageable) clone, so the slot can be reused.
        The ring lives in torch's caching pinned allocator, so only the first call pays for
        hipHostMalloc.  (Locking ordinary memory with hipHostRegister per call instead cost ~90 ms
        per buffer -- 13 ms per batch over a 30-batch job; CPU reads of either kind of page-locked
        memory take 0.03 ms per 512 KiB on the boxes measured, profiles/diag/stream_probe3.py.)
    Round 1 fetched every result with a host-blocking `.cpu()` behind an event wait; on the driver's
    box that pipeline was slower than the plain loop (BENCH_r01: 78.0 k vs 80.2 k embeddings/s against
    90.1 k resident), here it is not (profiles/diag/stream_probe3.py) -- host wake-up latencies differ
    between boxes, so the loop must not depend on them.
    Host batches may be pinned or pageable, and float64 (what the reference's DataLoader yields,
    main.py:137): the cast to fp32 happens on the device, as in test_step.
    `prepare`: a device-side step between the copy and the path, e.g. an MfccFrontEnd when the host batches are raw
    waveforms [B, n_samples] (the reference computes its MFCCs on the host, dataset.py:124-128; here 192 KB of samples per
    3 s utterance cross PCIe instead of 28.7 KB of features, and the features never leave the device)."""
    dev = torch.device(device) if device is not None else next(model.parameters()).device
    if dev.type != "cuda":
        raise RuntimeError("stream_x_vectors: the model must live on a HIP device")
    if depth < 1:
        raise ValueError("stream_x_vectors: depth must be >= 1")
    compute = torch.cuda.current_stream(dev)
    h2d, d2h = _side_streams(dev)
    n_ring = depth + 1
    slots = [None] * n_ring      # device inputs
    consumed = [None] * n_ring   # event: the batch that read the slot has been computed
    # one event per slot and hop, re-recorded every time round the ring (no hipEventCreate/Destroy per batch)
    arrived_ev = [torch.cuda.Event() for _ in range(n_ring)]
    done_ev = [torch.cuda.Event() for _ in range(n_ring)]
    landed_ev = [torch.cuda.Event() for _ in range(n_ring)]
    results = [None] * n_ring    # page-locked host result buffers [rows, D]
    inflight = []                # (landed event, host buffer, rows, device result)

    def locked(rows, D, dtype):
        return torch.empty((max(rows, 1), D), dtype=dtype, pin_memory=True)

    def harvest():
        ev, buf, n, _dev_out = inflight.pop(0)     # the device result is released here, after its copy landed
        ev.synchronize()                           # returns at once when query() was already true
        # a plain single-threaded memcpy: tensor.clone() goes through torch's intra-op thread pool, and waking
        # that pool (one thread per visible CPU -- 256 on the GPU box, under a 16-CPU quota) next to the
        # runtime's polling threads took 9-12 ms per 512 KiB clone, 100 ms at worst (stream_probe3.py)
        return torch.from_numpy(buf[:n].numpy().copy())

    def pipeline():
        try:
            yield from batches()
        finally:
            # the slots go back to the copy stream's pool when this frame dies: order that stream behind the
            # batches that read them (also when the consumer abandons the generator half way)
            h2d.wait_stream(compute)

    def batches():
        k = 0
        for xb in host_batches:
            if xb.device.type != "cpu":
                raise ValueError("stream_x_vectors: expected host tensors (device batches: call extract_x_vec directly)")
            i = k % n_ring
            fresh = slots[i] is None or slots[i].shape != xb.shape or slots[i].dtype != xb.dtype
            with torch.cuda.stream(h2d):
                if consumed[i] is not None:
                    h2d.wait_event(consumed[i])        # the slot's previous batch has been read
                if fresh:
                    # from the COPY stream's pool: whatever held that memory before was released in this stream's
                    # order, so the first batches need no wait on the compute stream either
                    slots[i] = None
                    slots[i] = torch.empty(xb.shape, dtype=xb.dtype, device=dev)
                slots[i].copy_(xb, non_blocking=True)
                arrived = arrived_ev[i]
                arrived.record(h2d)
            compute.wait_event(arrived)
            out = model.extract_x_vec(prepare(slots[i]) if prepare is not None else slots[i])   # enqueued on the current stream
            done = done_ev[i]
            done.record(compute)
            consumed[i] = done
            n, D = out.shape
            if results[i] is None or results[i].shape[0] < n or results[i].shape[1] != D:
                results[i] = locked(n, D, out.dtype)
            with torch.cuda.stream(d2h):               # not the compute stream: there the copy would queue
                d2h.wait_event(done)                   # behind the younger batches already enqueued
                results[i][:n].copy_(out, non_blocking=True)
                landed = landed_ev[i]                  # its previous batch was handed out: at most `depth` in flight
                landed.record(d2h)
            # `out` stays referenced until its copy has landed.  (out.record_stream(d2h) instead parks the block
            # in the allocator until an event query says the side stream is done with it; meanwhile the next
            # batches' results come from hipMalloc, which synchronises the device: 0.63 -> 0.48 ms per bf16
            # batch over the first calls, profiles/diag/stream_probe5.py)
            inflight.append((landed, results[i], n, out))
            k += 1
            while inflight and (len(inflight) > depth or inflight[0][0].query()):
                yield harvest()
        while inflight:
            yield harvest()

    yield from pipeline()


def extract_x_vectors(model, loader: Iterable) -> List[tuple]:
    """Lightning-free stand-in for `trainer.test(model)` in extraction mode
    (main.py:237-267): every (samples, labels, ids) batch of the loader goes through the overlapped
    pipeline above (same arithmetic as model.test_step: cast to fp32, extract_x_vec) and the
    records the reference accumulates in its module-global `x_vector` list come back in order."""
    records: List[tuple] = []
    meta = []

    def samples():
        for batch_samples, labels, ids in loader:
            meta.append((labels, ids))
            yield batch_samples if torch.is_tensor(batch_samples) else torch.as_tensor(batch_samples)

    for k, host in enumerate(stream_x_vectors(model, samples())):
        labels, ids = meta[k]
        records.extend(to_records(host, labels, ids))
    return records


def write_x_vector_csv(records: List[tuple], path: str, npy_sidecar: bool = True):
    """Same file the reference writes (pd.DataFrame(x_vector).to_csv): header ',0,1,2',
    vector column = numpy's str() of the float64 array (8 significant digits -- lossy, kept
    for drop-in compatibility); the exact vectors go to `<path>.npy` as a side-car."""
    import pandas as pd
    pd.DataFrame(records).to_csv(path)
    if npy_sidecar and records:
        np.save(path + ".npy", np.stack([r[2] for r in records]))


def read_x_vector_csv(path: str):
    """The reference's reader (main.py:276-279, plda_score_stat.py:16-17): returns
    (ids, labels, vectors[float64])."""
    import pandas as pd
    df = pd.read_csv(path)
    ids = np.array(df.iloc[:, 1])
    labels = np.array(df.iloc[:, 2], dtype=int)
    vecs = np.array([np.array(s[1:-1].split(), dtype=np.float64) for s in df.iloc[:, 3]])
    return ids, labels, vecs
