"""The threading contract of include/xvec_hip.h: "distinct handles may be used from distinct threads", and
xvec_last_error() is the calling thread's own.  One process, two Python threads (ctypes releases
the GIL around every library call, so the calls really overlap), two handles on cuda:0, two streams."""
import ctypes as C
import threading

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


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_two_threads_two_handles(sd42, synth, precision):
    from xvector_amd import hip
    n_calls = 50
    models = [_model(sd42, precision), _model(sd42, precision)]
    xs = [torch.from_numpy(synth.make_mfcc(64, 300, seed=11 + i)).to(DEV) for i in range(2)]
    # single-thread results first (this also packs the weights, on the main thread)
    want = [m.extract_x_vec(x).clone() for m, x in zip(models, xs)]
    torch.cuda.synchronize()
    assert models[0]._engine(torch.device(DEV)).h.value != models[1]._engine(torch.device(DEV)).h.value
    streams = [torch.cuda.Stream(DEV), torch.cuda.Stream(DEV)]
    gate = threading.Barrier(2)
    report = [None, None]

    def worker(i):
        try:
            bad = 0
            with torch.cuda.stream(streams[i]):
                for k in range(n_calls):
                    if k % 10 == 0:
                        gate.wait(timeout=60)                       # keep the two threads' calls interleaved
                    got = models[i].extract_x_vec(xs[i])
                    if k % 7 == 0 or k == n_calls - 1:
                        streams[i].synchronize()
                        bad += int(not torch.equal(got, want[i]))
                    if i == 0 and k == 25:
                        # a deliberate argument error on THIS thread only (T below the receptive field)
                        with pytest.raises(ValueError):
                            models[0].extract_x_vec(xs[0][:, :10])
                        eng = models[0]._engine(torch.device(DEV))
                        out = torch.empty((1, 512), device=DEV)
                        rc = hip.lib.xvec_forward(eng.h, xs[0].data_ptr(), None, 1, 10, hip.MODE_XVEC6, hip.F32, out.data_ptr(),
                                                  eng.workspace.data_ptr(), eng.workspace.numel(), streams[0].cuda_stream)
                        assert rc == hip.ERR_ARG
                        assert b"need at least" in hip.lib.xvec_last_error()
                streams[i].synchronize()
            report[i] = (bad, hip.lib.xvec_last_error())
        except BaseException as e:      # noqa: BLE001  (reported by the main thread)
            gate.abort()
            report[i] = e

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
        assert not t.is_alive(), "worker thread hung"
    for i, r in enumerate(report):
        assert not isinstance(r, BaseException), f"thread {i}: {r!r}"
    assert report[0][0] == 0 and report[1][0] == 0, f"results differ from the single-thread ones: {report}"
    assert b"need at least" in report[0][1]                 # thread 0 still sees its own error ...
    assert report[1][1] == b"", report[1][1]                # ... thread 1, which never failed, sees none


def test_layers_loaded_on_different_streams_in_any_order(sd42, synth):
    """xvec_load_tdnn(l) re-packs layer l + 1's bf16 copies from what an EARLIER load of l + 1 left -- possibly on another stream
    (ADVICE r04).  Here every layer is loaded on a stream of its own, last layer first, with nothing between the calls; the
    library orders the re-folds behind the loads they read (per-layer events).  Plain bf16 (the mode that folds) must then give
    exactly what a model loaded the usual way gives."""
    import ctypes as C
    import xvector_amd as xa
    from xvector_amd import hip
    from xvector_amd.model import _Engine
    dev = torch.device(DEV)
    ref = _model(sd42, "bf16")
    x = torch.from_numpy(synth.make_mfcc(64, 300, seed=3)).to(DEV)
    want = ref.extract_x_vec(x).clone()
    m = xa.XVectorModel(precision="bf16")
    m.load_state_dict(sd42)
    m = m.to(DEV).eval()
    eng = _Engine(m.hparams, dev)
    streams = [torch.cuda.Stream(DEV) for _ in range(6)]
    keep = []
    torch.cuda.synchronize()
    for i in reversed(range(5)):
        layer = m.time_context_layers[i]
        ts = [layer.linear.weight, layer.linear.bias, layer.norm.weight, layer.norm.bias, layer.norm.running_mean, layer.norm.running_var]
        cs = [t.detach().contiguous() for t in ts]
        keep += cs
        hip.check(hip.lib.xvec_load_tdnn(eng.h, i, *[t.data_ptr() for t in cs], layer.norm.eps, streams[i].cuda_stream))
    for which, lin in ((hip.SEG6, m.segment_layer6), (hip.SEG7, m.segment_layer7), (hip.OUTPUT, m.output)):
        hip.check(hip.lib.xvec_load_affine(eng.h, which, lin.weight.data_ptr(), lin.bias.data_ptr(), streams[5].cuda_stream))
    torch.cuda.synchronize()
    eng.signature = tuple((t.data_ptr(), t._version) for t in m._hot_tensors())
    m._engines[(dev.type, dev.index)] = eng
    assert torch.equal(m.extract_x_vec(x), want)
    assert m._engines[(dev.type, dev.index)] is eng          # the hand-loaded engine is the one that ran
