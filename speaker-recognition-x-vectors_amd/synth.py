"""Deterministic synthetic weights and MFCC inputs (numpy PCG64: stable across
platforms and library versions, unlike torch's generators).

There is no network for VoxCeleb or checkpoints, so benchmarks, golden fixtures and
parity tests all draw from here.  Distributions follow SURVEY.md §8(d): Linear
U(+-1/sqrt(fan_in)) (PyTorch's default init scale), BatchNorm gamma~U(0.5,1.5),
beta~N(0,0.1), running_mean~N(0,0.5), running_var~U(0.5,2): non-trivial BN statistics
so an epilogue bug cannot hide behind an identity-like BatchNorm.

Keys and shapes are those of the reference's state_dict (main.py:38-47).
"""
from __future__ import annotations

from typing import Dict

import numpy as np

CONTEXTS = [[-2, -1, 0, 1, 2], [-2, 0, 2], [-3, 0, 3], [0], [0]]
POOL_CHANNELS = 1500


def layer_dims(input_size=24, hidden_size=512):
    """(in_channels, out_channels, context) of the five TDNN layers (main.py:38-44)."""
    ins = [input_size, hidden_size, hidden_size, hidden_size, hidden_size]
    outs = [hidden_size, hidden_size, hidden_size, hidden_size, POOL_CHANNELS]
    return list(zip(ins, outs, CONTEXTS))


def make_state_dict(seed: int = 42, input_size: int = 24, hidden_size: int = 512,
                    num_classes: int = 1211, x_vector_size: int = 512,
                    batch_norm: bool = True) -> Dict[str, np.ndarray]:
    rng = np.random.default_rng(seed)
    sd: Dict[str, np.ndarray] = {}

    def linear(name, fan_in, fan_out):
        k = 1.0 / np.sqrt(fan_in)
        sd[name + ".weight"] = rng.uniform(-k, k, (fan_out, fan_in)).astype(np.float32)
        sd[name + ".bias"] = rng.uniform(-k, k, (fan_out,)).astype(np.float32)

    for i, (cin, cout, ctx) in enumerate(layer_dims(input_size, hidden_size)):
        pre = f"time_context_layers.{i}."
        linear(pre + "linear", cin * len(ctx), cout)
        if batch_norm:
            sd[pre + "norm.weight"] = rng.uniform(0.5, 1.5, cout).astype(np.float32)
            sd[pre + "norm.bias"] = (0.1 * rng.standard_normal(cout)).astype(np.float32)
            sd[pre + "norm.running_mean"] = (0.5 * rng.standard_normal(cout)).astype(np.float32)
            sd[pre + "norm.running_var"] = rng.uniform(0.5, 2.0, cout).astype(np.float32)
            sd[pre + "norm.num_batches_tracked"] = np.array(1, dtype=np.int64)
    linear("segment_layer6", 2 * POOL_CHANNELS, x_vector_size)
    linear("segment_layer7", x_vector_size, x_vector_size)
    linear("output", x_vector_size, num_classes)
    return sd


def make_mfcc(batch: int, frames: int, input_size: int = 24, seed: int = 0) -> np.ndarray:
    """N(0,1) fp32 stand-in for a [batch, frames, 24] MFCC tensor."""
    rng = np.random.default_rng(seed)
    return rng.standard_normal((batch, frames, input_size), dtype=np.float32)


def make_lengths(batch: int, lo: int = 200, hi: int = 1000, seed: int = 1234) -> np.ndarray:
    """BASELINE config 3: lengths ~ U{lo..hi}."""
    return np.random.default_rng(seed).integers(lo, hi + 1, batch).astype(np.int32)


def make_plda(dim: int = 512, rank: int = 200, seed: int = 21):
    """A random, well-conditioned PLDA model (mean[D], F[D,R], Sigma[D,D], float64) for the scoring back end's
    benchmark figure (there is no trained model without VoxCeleb): the shapes the reference's
    `plda_classifier.setup_plda(rank_f=...)` trains (plda_classifier.py:31-45)."""
    rng = np.random.default_rng(seed)
    mean = rng.normal(0, 1, dim)
    F = rng.normal(0, 1 / np.sqrt(dim), (dim, rank))
    A = rng.normal(0, 1 / np.sqrt(dim), (dim, dim))
    return mean, F, A @ A.T + 0.5 * np.eye(dim)
