"""XVectorModel.pipelined(): every batch's tail (pooling merge + segment layers) on the library's own stream beside the next
batch's first layer (include/xvec_hip.h, xvec_set_tail_overlap; model.py, PipelinedPath) gives the results of the plain loop
bit for bit, in input order, for every arithmetic, for ragged batches, with a front end in the pipeline, with plain calls in
between, and for more batches in flight than the library keeps completion events for."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _model(sd, precision):
    import xvector_amd as xa
    m = xa.XVectorModel(precision=precision)
    m.load_state_dict(sd)
    return m.to(DEV).eval()


@pytest.mark.parametrize("precision,B", [("fp32", 24), ("bf16", 96), ("bf16x3", 64)])
def test_pipelined_equals_plain_loop(sd42, synth, precision, B):
    m = _model(sd42, precision)
    batches = [torch.from_numpy(synth.make_mfcc(B if k % 3 else B // 2, 300, seed=40 + k)).to(DEV) for k in range(7)]
    want = [m.extract_x_vec(x).clone() for x in batches]
    pipe = m.pipelined()
    for _ in range(2):
        got = pipe.map(batches)
        torch.cuda.synchronize()
        assert len(got) == len(want)
        for k, (g, w) in enumerate(zip(got, want)):
            assert torch.equal(g, w), f"batch {k}: pipelined result differs from the plain loop's"


def test_pipelined_ragged_and_logits(sd42, synth):
    m = _model(sd42, "fp32")
    x = torch.from_numpy(synth.make_mfcc(12, 400, seed=5)).to(DEV)
    lens = [400, 333, 200, 15 + 14, 64, 399, 250, 300, 123, 77, 400, 16]
    want_r = m.extract_x_vec(x, lengths=lens).clone()
    want_l = m(x).clone()
    pipe = m.pipelined()
    a = pipe.submit(x, lengths=lens)
    b = pipe.submit(x, logits=True)
    c = pipe.submit(x, lengths=torch.tensor(lens))
    assert torch.equal(a.wait(), want_r) and torch.equal(b.wait(), want_l) and torch.equal(c.wait(), want_r)


def test_pipelined_with_front_end(sd42):
    import xvector_amd as xa
    m = _model(sd42, "bf16")
    fe = xa.MfccFrontEnd(device=DEV)
    g = torch.Generator(device=DEV).manual_seed(3)
    waves = [0.1 * torch.randn((64, 48000), device=DEV, generator=g) for _ in range(4)]
    want = [m.extract_x_vec(fe(w)).clone() for w in waves]
    got = m.pipelined(prepare=fe).map(waves)
    torch.cuda.synchronize()
    for g_, w_ in zip(got, want):
        assert torch.equal(g_, w_)


def test_pipelined_rejects_cpu_tensors(sd42, synth):
    m = _model(sd42, "fp32")
    with pytest.raises(RuntimeError):
        m.pipelined().submit(torch.zeros((2, 300, 24)))


def test_many_in_flight_and_plain_calls_between(sd42, synth):
    """40 batches submitted before the first result is asked for (the handle keeps 16 completion events: older tickets wait
    for a newer tail), plain extract_x_vec calls between submits (they must come back complete: the overlap is per call),
    a batch that makes the engine's workspace grow while tails are in flight."""
    m = _model(sd42, "bf16")
    xs = [torch.from_numpy(synth.make_mfcc(8 + (k % 5) * 4, 300, seed=70 + k)).to(DEV) for k in range(8)]
    big = torch.from_numpy(synth.make_mfcc(80, 300, seed=99)).to(DEV)
    want = [m.extract_x_vec(x).clone() for x in xs]
    want_big = m.extract_x_vec(big).clone()
    m._engines.clear()                                     # a fresh engine: its workspace grows on the way
    pipe = m.pipelined()
    pend, plain = [], []
    for k in range(40):
        pend.append(pipe.submit(xs[k % 8]))
        if k % 9 == 4:
            plain.append((k % 8, m.extract_x_vec(xs[k % 8])))
        if k == 20:
            pend_big = pipe.submit(big)
    torch.cuda.synchronize()          # (not needed for correctness of result(); keeps the check below simple)
    for k, p in enumerate(pend):
        assert torch.equal(p.result(), want[k % 8]), f"submit {k}"
    assert torch.equal(pend_big.result(), want_big)
    for i, got in plain:
        assert torch.equal(got, want[i])


def test_overlapped_call_cannot_be_captured(sd42, synth):
    from xvector_amd import hip
    m = _model(sd42, "fp32")
    x = torch.from_numpy(synth.make_mfcc(4, 300, seed=1)).to(DEV)
    m.extract_x_vec(x)
    torch.cuda.synchronize()
    pipe = m.pipelined()
    g = torch.cuda.CUDAGraph()
    with pytest.raises(hip.XvecError) as e:
        with torch.cuda.graph(g):
            pipe.submit(x)
    assert e.value.code == hip.ERR_STATE
    torch.cuda.synchronize()
    assert torch.equal(m.extract_x_vec(x), pipe.submit(x).wait())        # and the handle is usable afterwards
