"""CPU oracle for the x-vector embedding-extraction path.  TEST INFRASTRUCTURE ONLY.

This file is a CPU restatement (pure PyTorch tensor ops, no Lightning, no nn.Module
state) of the algorithm the reference implements in

    /root/reference/tdnn_layer.py:26-60   (TdnnLayer.forward, get_time_context)
    /root/reference/main.py:59-94         (stat_pool, forward, extract_x_vec)

It is the checker for the HIP path, never the product: only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import it.  The product
package never imports anything under oracle/ and fails loudly without its HIP library.

Parity pin: the reference ships no tests or golden vectors for this path
(SURVEY.md §4), so the oracle is pinned against outputs of the reference itself,
generated in the build container by tests/golden/make_golden.py (which imports the
reference's own XVectorModel) and committed under tests/golden/*.npz.  See
tests/test_oracle_golden.py.

Every function takes a plain dict of tensors keyed by the reference's state_dict
names (main.py:38-47):  time_context_layers.{i}.linear.{weight,bias},
time_context_layers.{i}.norm.{weight,bias,running_mean,running_var},
segment_layer6/7.{weight,bias}, output.{weight,bias}.
"""
from __future__ import annotations

import time
from typing import Dict, List, Optional, Sequence

import torch

# Layer stack of the reference (main.py:38-44): (context, out_channels or None=hidden)
CONTEXTS: List[List[int]] = [[-2, -1, 0, 1, 2], [-2, 0, 2], [-3, 0, 3], [0], [0]]
POOL_CHANNELS = 1500          # main.py:43 (hard-coded)
BN_EPS = 1e-5                 # nn.BatchNorm1d default, tdnn_layer.py:22
TOTAL_CONTEXT = 14            # frames lost by the valid convolutions: 4 + 4 + 6


def get_time_context(x: torch.Tensor, c: Sequence[int]) -> List[torch.Tensor]:
    """tdnn_layer.py:43-60.  One time-shifted view of x[B,T,C] per context offset.

    Slice j covers input frames [c[-1]+c[j], T + c[0]+c[j]) -- the last one is
    open-ended -- so every slice has T - (c[-1]-c[0]) frames for a symmetric context.
    """
    last = len(c) - 1
    out = []
    for cc in c:
        if cc != c[last]:
            out.append(x[:, c[last] + cc: c[0] + cc, :])
        else:
            out.append(x[:, c[last] + cc:, :])
    return out


def tdnn_layer(x: torch.Tensor, p: Dict[str, torch.Tensor], prefix: str,
               context: Sequence[int], batch_norm: bool = True) -> torch.Tensor:
    """tdnn_layer.py:26-41 in eval mode: cat(context) -> Linear -> ReLU -> BatchNorm1d.

    BatchNorm is applied AFTER the ReLU with running statistics (eval):
    y = (v - mean) / sqrt(var + eps) * gamma + beta, per output channel.
    Dropout (p=0 by default, tdnn_layer.py:23-24,33-34) is the identity in eval.
    """
    xc = torch.cat(get_time_context(x, context), 2)                      # :28-29
    v = xc @ p[prefix + "linear.weight"].t() + p[prefix + "linear.bias"]  # :30
    v = torch.relu(v)                                                    # :31
    if batch_norm:                                                       # :36-39
        mean = p[prefix + "norm.running_mean"]
        var = p[prefix + "norm.running_var"]
        v = (v - mean) / torch.sqrt(var + BN_EPS) * p[prefix + "norm.weight"] \
            + p[prefix + "norm.bias"]
    return v


def time_context_layers(x: torch.Tensor, p: Dict[str, torch.Tensor],
                        batch_norm: bool = True, upto: int = 5) -> torch.Tensor:
    """main.py:38-44: the nn.Sequential of five TdnnLayers."""
    for i in range(upto):
        x = tdnn_layer(x, p, f"time_context_layers.{i}.", CONTEXTS[i], batch_norm)
    return x


def stat_pool(x: torch.Tensor) -> torch.Tensor:
    """main.py:59-63: mean over time ‖ UNBIASED std over time (torch.std default)."""
    mean = torch.mean(x, 1)
    std = torch.std(x, 1)
    return torch.cat((mean, std), 1)


def _linear(x, p, name):
    return x @ p[name + ".weight"].t() + p[name + ".bias"]


def forward(x: torch.Tensor, p: Dict[str, torch.Tensor], batch_norm: bool = True) -> torch.Tensor:
    """main.py:66-75: logits [B, num_classes] (no softmax)."""
    out = stat_pool(time_context_layers(x, p, batch_norm))
    out = torch.relu(_linear(out, p, "segment_layer6"))
    out = torch.relu(_linear(out, p, "segment_layer7"))
    return _linear(out, p, "output")


def extract_x_vec(x: torch.Tensor, p: Dict[str, torch.Tensor], layer: int = 6,
                  batch_norm: bool = True) -> torch.Tensor:
    """main.py:81-94: pre-ReLU output of segment_layer6 (layer 6 or anything else)
    or of segment_layer7 fed by relu(segment_layer6) (layer 7)."""
    out = stat_pool(time_context_layers(x, p, batch_norm))
    if layer == 7:
        return _linear(torch.relu(_linear(out, p, "segment_layer6")), p, "segment_layer7")
    return _linear(out, p, "segment_layer6")


def extract_x_vec_ragged(x: torch.Tensor, lengths: Sequence[int], p, layer: int = 6,
                         batch_norm: bool = True) -> torch.Tensor:
    """Semantics of a padded batch with a length mask (BASELINE config 3): the
    reference has no mask, so the definition is the reference run per utterance at
    batch=1 on the un-padded slice x[i, :lengths[i]]."""
    rows = [extract_x_vec(x[i:i + 1, :int(n)], p, layer, batch_norm) for i, n in enumerate(lengths)]
    return torch.cat(rows, 0)


def cast_params(p: Dict[str, torch.Tensor], dtype) -> Dict[str, torch.Tensor]:
    return {k: v.to(dtype) for k, v in p.items() if v.is_floating_point()}


def flops_per_utt(T: int, layer: int = 6) -> int:
    """Algorithmic FLOPs of extract_x_vec for one T-frame utterance."""
    f = 2 * (120 * 512 * (T - 4) + 1536 * 512 * (T - 8) + 1536 * 512 * (T - 14)
             + 512 * 512 * (T - 14) + 512 * 1500 * (T - 14)) + 4 * 1500 * (T - 14) + 2 * 3000 * 512
    if layer == 7:
        f += 2 * 512 * 512
    return f


def _time_leg(p, x, threads: int, budget_s: float, warmups: int = 3, reps: int = 10):
    """One leg of the protocol: `warmups` untimed passes, then the MEDIAN of `reps` timed passes over
    the batch x with torch.set_num_threads(threads).  When warmups + reps passes would not fit
    budget_s (estimated from the first pass), the leg is cut to 1 warm-up and >= 3 timed passes and
    says so in its record."""
    torch.set_num_threads(threads)
    with torch.no_grad():
        t0 = time.perf_counter()
        extract_x_vec(x, p)                           # first pass: warm-up #1 and the cost estimate
        est = time.perf_counter() - t0
        if est * (warmups + reps) > budget_s:
            warmups, reps = 1, max(3, min(reps, int(budget_s / max(est, 1e-9)) - 1))
        for _ in range(warmups - 1):
            extract_x_vec(x, p)
        times = []
        for _ in range(reps):
            t0 = time.perf_counter()
            extract_x_vec(x, p)
            times.append(time.perf_counter() - t0)
    times.sort()
    med = times[len(times) // 2] if len(times) % 2 else 0.5 * (times[len(times) // 2 - 1] + times[len(times) // 2])
    B, T = x.shape[0], x.shape[1]
    return {"batch": B, "threads": torch.get_num_threads(), "warmups": warmups, "reps": reps,
            "median_ms": round(med * 1e3, 3), "embeddings_per_s": round(B / med, 2),
            "gflops": round(B * flops_per_utt(T) / med / 1e9, 1), "seconds": round(sum(times) + est * warmups, 2)}


def time_cpu_baseline(p: Dict[str, torch.Tensor], T: int = 300, budget_s: float = 40.0,
                      threads: Optional[int] = None, seed: int = 0):
    """SURVEY.md 8(d) / BASELINE.md 4 protocol: this oracle (the reference's op sequence on the host
    CPU, fp32) on identical synthetic inputs at B=1 and B=64, with n = `threads` (all usable cores)
    and n = 1; 3 warm-ups + median of 10 per leg, each leg bounded to its share of budget_s (the
    one-thread B=64 leg, by far the longest, gets 55 %).
    Returns the four leg records, keyed "b64_all", "b1_all", "b64_1t", "b1_1t"."""
    n_all = threads if threads is not None else torch.get_num_threads()
    g = torch.Generator().manual_seed(seed)
    x64 = torch.randn(64, T, 24, generator=g, dtype=torch.float32)
    x1 = x64[:1].clone()
    keep = torch.get_num_threads()
    legs = {}
    try:
        for key, x, n, share in (("b64_all", x64, n_all, 0.35), ("b1_all", x1, n_all, 0.05), ("b1_1t", x1, 1, 0.05),
                                 ("b64_1t", x64, 1, 0.55)):
            legs[key] = _time_leg(p, x, n, budget_s * share)
    finally:
        torch.set_num_threads(keep)
    return legs
