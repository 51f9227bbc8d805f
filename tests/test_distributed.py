"""The N>1 path on CPU: world_size-2 gloo processes exercise the sharding + all-gather
plumbing of extract.py (the collective is the same call RCCL serves on the GPUs).  The
per-utterance 'extractor' here is a deterministic stand-in so the test checks routing, row
order and trimming -- numerics of the real extractor are the GPU parity tests' job."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _fake_extract(x):
    # [B,T,C] -> [B,8]: depends on every frame of the utterance and on nothing else
    return torch.stack([x.mean(dim=(1, 2)) * (k + 1) + x[:, 0, 0] for k in range(8)], 1)


def _make_batch(lo, hi):
    g = torch.Generator().manual_seed(0)
    full = torch.randn(23, 20, 4, generator=g)
    return full[lo:hi]


def _worker(rank, world, port, n_total, ragged, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from xvector_amd import extract
    if not ragged:
        got = extract.extract_sharded(_fake_extract, _make_batch, n_total, batch_size=4)
    else:
        lengths = [5 + (i * 7) % 16 for i in range(n_total)]
        x = _make_batch(0, n_total)
        got = extract.extract_balanced(lambda mine: _fake_extract(x[mine]) if mine else torch.zeros((0, 8)), lengths)
    torch.save(got, os.path.join(out_dir, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_total,ragged", [(23, False), (8, False), (1, False), (23, True)])
def test_sharded_extraction_world2(tmp_path, n_total, ragged):
    port = 29500 + os.getpid() % 2000 + n_total + (50 if ragged else 0)
    mp.spawn(_worker, args=(2, port, n_total, ragged, str(tmp_path)), nprocs=2, join=True)
    want = _fake_extract(_make_batch(0, n_total))
    for r in range(2):
        got = torch.load(os.path.join(str(tmp_path), f"r{r}.pt"))
        assert got.shape == want.shape
        assert torch.allclose(got, want, rtol=0, atol=0), f"rank {r} differs"


@pytest.mark.parametrize("n_total,ragged", [(23, False), (3, False), (23, True)])
def test_sharded_extraction_world4(tmp_path, n_total, ragged):
    """Four ranks (half the 8-GPU node of BASELINE configs[3]): uneven shards (6, 6, 6, 5), a rank with no
    utterance at all (3 over 4 ranks), and the frame-balanced ragged split."""
    port = 33500 + os.getpid() % 2000 + n_total + (50 if ragged else 0)
    mp.spawn(_worker, args=(4, port, n_total, ragged, str(tmp_path)), nprocs=4, join=True)
    want = _fake_extract(_make_batch(0, n_total))
    for r in range(4):
        got = torch.load(os.path.join(str(tmp_path), f"r{r}.pt"))
        assert got.shape == want.shape
        assert torch.allclose(got, want, rtol=0, atol=0), f"rank {r} differs"


def test_single_process_is_identity():
    sys.path.insert(0, ROOT)
    from xvector_amd import extract
    got = extract.extract_sharded(_fake_extract, _make_batch, 10, batch_size=3)
    assert torch.equal(got, _fake_extract(_make_batch(0, 10)))


def _worker_world1(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=0, world_size=1)
    from xvector_amd import extract
    a = extract.extract_sharded(_fake_extract, _make_batch, 10, batch_size=3, force_collective=True)
    lengths = [5 + (i * 7) % 16 for i in range(10)]
    x = _make_batch(0, 10)
    b = extract.extract_balanced(lambda mine: _fake_extract(x[mine]), lengths, force_collective=True)
    torch.save((a, b), os.path.join(out_dir, "w1.pt"))
    dist.destroy_process_group()


def test_forced_collective_at_world1(tmp_path):
    """The one-GPU rehearsal path of the RCCL leg (tests/test_multigpu_gpu.py) on gloo: with
    force_collective the all-gather and the assembly run even for a single rank."""
    mp.spawn(_worker_world1, args=(1, 31500 + os.getpid() % 2000, str(tmp_path)), nprocs=1, join=True)
    a, b = torch.load(os.path.join(str(tmp_path), "w1.pt"))
    want = _fake_extract(_make_batch(0, 10))
    assert torch.equal(a, want) and torch.equal(b, want)


def test_single_rank_order_is_undone():
    """gather_embeddings(order=...) without a process group: rows return to input order."""
    sys.path.insert(0, ROOT)
    from xvector_amd import extract
    x = _make_batch(0, 9)
    perm = [4, 0, 8, 2, 6, 1, 7, 3, 5]
    got = extract.gather_embeddings(_fake_extract(x[perm]), 9, order=[perm])
    assert torch.equal(got, _fake_extract(x))
    got = extract.gather_embeddings(_fake_extract(x[perm]), 9, order=[perm], force=True)
    assert torch.equal(got, _fake_extract(x))
