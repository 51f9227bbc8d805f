"""Parity of the HIP path (through the C ABI) with the oracle and the golden vectors.
Run on the MI355X box: python -m pytest tests -m gpu."""
import numpy as np
import pytest
import torch

import xvector_oracle as oracle
from conftest import assert_parity, float_params, load_golden

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _gpu(a):
    return torch.as_tensor(np.asarray(a)).to(DEV)


def test_library_is_the_native_one():
    from xvector_amd import hip
    assert "gfx950" in hip.version()
    assert torch.cuda.get_device_properties(0).gcnArchName.startswith("gfx950")


# ------------------------------------------------------------------------------- per layer
def test_g2_tdnn_layers_vs_golden(gpu_model, synth):
    """TdnnLayer.forward for each of the five layer shapes (tdnn_layer.py:26-41)."""
    g = load_golden("g2_layers.npz")
    h = _gpu(synth.make_mfcc(int(g["B"]), int(g["T"]), seed=int(g["seed_x"])))
    for i, layer in enumerate(gpu_model.time_context_layers):
        h = layer(h)
        assert list(h.shape) == g[f"l{i}_shape"].tolist()
        assert_parity(h[:, g[f"l{i}_frames"].tolist(), :], g[f"l{i}_rows"], 1e-4, f"layer {i} rows")
        assert_parity(h.double().sum(dim=(0, 1))[None], g[f"l{i}_sum"][None], 1e-4, f"layer {i} sums")


@pytest.mark.parametrize("layer,B,T", [(0, 2, 15), (0, 5, 131), (1, 3, 129), (2, 1, 300), (3, 7, 37), (4, 2, 150)])
def test_tdnn_layer_vs_oracle_every_element(gpu_model, sd42, layer, B, T):
    """Whole-tensor check against the fp64 oracle at sizes that straddle tile edges
    (T not a multiple of anything, B*T below/above one 128-row tile)."""
    p64 = oracle.cast_params(float_params(sd42), torch.float64)
    cin = 24 if layer == 0 else 512
    x = torch.from_numpy(np.random.default_rng(100 + layer).standard_normal((B, T, cin), dtype=np.float32))
    ref = oracle.tdnn_layer(x.double(), p64, f"time_context_layers.{layer}.", oracle.CONTEXTS[layer])
    got = gpu_model.time_context_layers[layer](x.to(DEV))
    assert_parity(got, ref, 1e-4, f"layer {layer} B={B} T={T}")


# ------------------------------------------------------------------------------- pooling
def test_g3_stat_pool_vs_golden(gpu_model):
    g = load_golden("g3_stat_pool.npz")
    rng = np.random.default_rng(int(g["seed"]))
    x = (rng.standard_normal((4, 286, 1500)) * float(g["scale"]) + float(g["shift"])).astype(np.float32)
    assert_parity(gpu_model.stat_pool(_gpu(x)), g["pool_286"], 1e-4, "pool 286")
    for n in (2, 3):
        assert_parity(gpu_model.stat_pool(_gpu(np.ascontiguousarray(x[:, :n, :64]))), g[f"pool_{n}"], 1e-4, f"pool {n}")


def test_stat_pool_edge_cases(gpu_model):
    rng = np.random.default_rng(5)
    # n == 1 -> NaN std, finite mean (torch.std semantics)
    one = gpu_model.stat_pool(_gpu(rng.standard_normal((3, 1, 10), dtype=np.float32))).cpu()
    assert torch.isnan(one[:, 10:]).all() and torch.isfinite(one[:, :10]).all()
    # |mean| >> std: the shifted/Chan formulation must not cancel catastrophically
    x = (1000.0 + 0.01 * rng.standard_normal((2, 286, 1500))).astype(np.float32)
    ref = oracle.stat_pool(torch.from_numpy(x).double())
    assert_parity(gpu_model.stat_pool(_gpu(x)), ref, 2e-3, "ill-conditioned pool")   # fp32 input rounding dominates
    # constant column -> std exactly 0, not NaN
    c = gpu_model.stat_pool(torch.full((1, 50, 8), 3.25, device=DEV)).cpu()
    assert torch.equal(c[:, 8:], torch.zeros(1, 8)) and torch.equal(c[:, :8], torch.full((1, 8), 3.25))
    # odd channel count (scalar-load kernel) and a length mask
    x = rng.standard_normal((3, 40, 7), dtype=np.float32)
    lens = [40, 2, 17]
    ref = torch.cat([oracle.stat_pool(torch.from_numpy(x[i:i + 1, :n]).double()) for i, n in enumerate(lens)])
    assert_parity(gpu_model.stat_pool(_gpu(x), lengths=lens), ref, 1e-4, "masked pool")


# ------------------------------------------------------------------------------- affines
@pytest.mark.parametrize("M", [1, 5, 32, 33, 256])
def test_affine_layers(gpu_model, sd42, M):
    p64 = oracle.cast_params(float_params(sd42), torch.float64)
    rng = np.random.default_rng(M)
    for name, K in (("segment_layer6", 3000), ("segment_layer7", 512), ("output", 512)):
        x = torch.from_numpy(rng.standard_normal((M, K), dtype=np.float32))
        ref = x.double() @ p64[name + ".weight"].t() + p64[name + ".bias"]
        assert_parity(gpu_model.affine(name, x.to(DEV)), ref, 1e-4, f"{name} M={M}")
        assert_parity(gpu_model.affine(name, x.to(DEV), relu=True), torch.relu(ref), 1e-4, f"{name} relu M={M}")


@pytest.mark.parametrize("precision", ["fp32", "bf16", "bf16x3"])
@pytest.mark.parametrize("B", [3, 37, 256])
def test_segment_layers_inside_the_path(sd42, synth, precision, B):
    """The segment-level layers as the forward path runs them (split over K into the dead activation buffers, the
    ranges summed in order by a second kernel; bf16x3 products in plain bf16): from the SAME run's pooled
    statistics, segment_layer6 / 7 / output in fp64 must agree at the fp32 bar in every precision -- the frame-level
    arithmetic is out of the comparison.  The fixed summation order makes repeats bit-identical."""
    import xvector_amd as xa
    p64 = oracle.cast_params(float_params(sd42), torch.float64)
    m = xa.XVectorModel(precision=precision)
    m.load_state_dict(sd42)
    m = m.to(DEV).eval()
    x = _gpu(synth.make_mfcc(B, 150, seed=B))
    pooled = m.pooled(x).double().cpu()
    z6 = pooled @ p64["segment_layer6.weight"].t() + p64["segment_layer6.bias"]
    z7 = torch.relu(z6) @ p64["segment_layer7.weight"].t() + p64["segment_layer7.bias"]
    lg = torch.relu(z7) @ p64["output.weight"].t() + p64["output.bias"]
    x6 = m.extract_x_vec(x)
    assert_parity(x6, z6, 3e-5, f"{precision} segment_layer6 B={B}")
    m.x_vec_extract_layer = 7
    assert_parity(m.extract_x_vec(x), z7, 1e-4, f"{precision} segment_layer7 B={B}")
    m.x_vec_extract_layer = 6
    assert_parity(m.forward(x), lg, 1e-4, f"{precision} logits B={B}")
    for _ in range(20):
        assert torch.equal(m.extract_x_vec(x), x6), "repeat runs of the split-K reduction differ"


# ------------------------------------------------------------------------------- whole path
@pytest.mark.parametrize("B,T", [(1, 299), (8, 299), (1, 300), (8, 300)])
def test_g4_full_path_vs_golden(sd42, synth, B, T):
    import xvector_amd as xa
    g = load_golden("g4_full.npz")
    key = f"B{B}_T{T}"
    x = _gpu(synth.make_mfcc(B, T, seed=int(g[key + "_seed_x"])))
    m6 = xa.XVectorModel()
    m6.load_state_dict(sd42)
    m6 = m6.to(DEV)
    assert_parity(m6.extract_x_vec(x), g[key + "_xvec6"], 1e-4, "xvec6")
    assert_parity(m6(x), g[key + "_logits"], 1e-4, "logits")
    m6.x_vec_extract_layer = 7
    assert_parity(m6.extract_x_vec(x), g[key + "_xvec7"], 1e-4, "xvec7")
    m6.x_vec_extract_layer = 3          # "any other value behaves as 6" (main.py:91-92)
    assert_parity(m6.extract_x_vec(x), g[key + "_xvec6"], 1e-4, "xvec default branch")


def test_g5_ragged_vs_golden(gpu_model, synth):
    """Padded batch + lengths == the reference run per utterance on the un-padded slice."""
    g = load_golden("g5_ragged.npz")
    lens = g["lengths"].tolist()
    x = synth.make_mfcc(3, 1000, seed=int(g["seed_x"]))
    for i, n in enumerate(lens):           # poison the padding: it must not leak
        x[i, n:] = np.nan
    assert_parity(gpu_model.extract_x_vec(_gpu(x), lengths=lens), g["xvec6"], 1e-4, "ragged xvec6")
    assert_parity(gpu_model(_gpu(x), lengths=lens), g["logits"], 1e-4, "ragged logits")
    # packed entry point, same utterances without any padding
    packed = np.concatenate([x[i, :n] for i, n in enumerate(lens)])
    offs = np.concatenate([[0], np.cumsum(lens)]).tolist()
    assert_parity(gpu_model.extract_packed(_gpu(packed), offs), g["xvec6"], 1e-4, "packed xvec6")


@pytest.mark.parametrize("tag", ["bn", "nobn"])
def test_g6_tiny_model(synth, tag):
    """Reduced-width model (hidden 32, x-vector 16, 10 classes): exercises channel padding
    of every layer and the batch_norm=False constructor path."""
    import xvector_amd as xa
    g = load_golden("g6_tiny.npz")
    bn = tag == "bn"
    if bn:
        sd = {k[len("bn/w/"):]: torch.from_numpy(g[k]) for k in g.files if k.startswith("bn/w/")}
    else:
        sd = {k: torch.from_numpy(np.asarray(v)) for k, v in synth.make_state_dict(
            seed=int(g["nobn/seed_w"]), input_size=24, hidden_size=32, num_classes=10, x_vector_size=16,
            batch_norm=False).items()}
    m = xa.XVectorModel(input_size=24, hidden_size=32, num_classes=10, x_vector_size=16, batch_norm=bn)
    m.load_state_dict(sd)
    m = m.to(DEV)
    x = _gpu(g[f"{tag}/x"])
    assert_parity(m.extract_x_vec(x), g[f"{tag}/xvec6"], 1e-4, "tiny xvec6")
    assert_parity(m(x), g[f"{tag}/logits"], 1e-4, "tiny logits")
    m.x_vec_extract_layer = 7
    assert_parity(m.extract_x_vec(x), g[f"{tag}/xvec7"], 1e-4, "tiny xvec7")
    h = x
    for layer in m.time_context_layers:
        h = layer(h)
    assert_parity(h[:, g[f"{tag}/frames_t"].tolist()], g[f"{tag}/frames"], 1e-4, "tiny frames")


def test_g7_caller_contract(gpu_model, synth):
    """test_step / test_epoch_end I/O (main.py:135-146): float64 loader output is cast to
    fp32, rows keep input order, vectors are widened exactly to float64."""
    from xvector_amd import extract
    g = load_golden("g7_caller.npz")
    xs = torch.from_numpy(synth.make_mfcc(5, 299, seed=int(g["seed_x"]))).double()
    labels = torch.from_numpy(g["labels"])
    ids = g["ids"].tolist()
    recs = extract.extract_x_vectors(gpu_model, [(xs, labels, ids)])
    assert [r[0] for r in recs] == g["out_ids"].tolist()
    assert [r[1] for r in recs] == g["out_labels"].tolist()
    vecs = np.stack([r[2] for r in recs])
    assert vecs.dtype == np.float64
    assert np.array_equal(vecs, vecs.astype(np.float32).astype(np.float64))      # exact widening
    assert_parity(vecs, g["out_vecs"], 1e-4, "caller vectors")


def test_streamed_extraction_from_host_batches(gpu_model, synth):
    """extract.stream_x_vectors (next batch's H2D overlapped with compute): same vectors, same
    order as the direct calls; float64 host batches as the reference's DataLoader yields them."""
    from xvector_amd import extract
    sizes = [7, 64, 1, 33, 64, 64, 5]
    batches = [torch.from_numpy(synth.make_mfcc(b, 299, seed=40 + i)).double() for i, b in enumerate(sizes)]
    got = list(extract.stream_x_vectors(gpu_model, iter(batches), depth=2))
    assert [g.shape[0] for g in got] == sizes and all(g.device.type == "cpu" for g in got)
    for g, xb in zip(got, batches):
        assert torch.equal(g, gpu_model.extract_x_vec(xb.to(DEV).float()).cpu())
    # pinned input, deeper pipeline
    pinned = [b.float().pin_memory() for b in batches]
    got2 = list(extract.stream_x_vectors(gpu_model, iter(pinned), depth=4))
    assert all(torch.equal(a, b) for a, b in zip(got, got2))
    with pytest.raises(ValueError):
        list(extract.stream_x_vectors(gpu_model, iter([batches[0].to(DEV)])))


def test_streamed_extraction_from_host_waveforms(gpu_model):
    """extract.stream_x_vectors(prepare=MfccFrontEnd): host WAVEFORMS in (what the reference's Dataset holds before its
    per-utterance MFCC call, dataset.py:124-128), the front end on the device between the copy and the path -- the vectors of
    the direct calls, in order."""
    import xvector_amd as xa
    from xvector_amd import extract
    fe = xa.MfccFrontEnd(device=DEV)
    g = torch.Generator().manual_seed(8)
    waves = [0.1 * torch.randn((b, 48000), generator=g) for b in (5, 32, 1, 32)]
    got = list(extract.stream_x_vectors(gpu_model, iter(waves), depth=2, prepare=fe))
    assert [v.shape for v in got] == [(b, 512) for b in (5, 32, 1, 32)]
    for v, w in zip(got, waves):
        assert torch.equal(v, gpu_model.extract_x_vec(fe(w.to(DEV))).cpu())


def test_path_is_graph_capturable(gpu_model, synth):
    """xvec_forward neither synchronises nor allocates: torch.cuda.graph captures the whole path and
    a replay on new input contents reproduces the eager result bit for bit."""
    x = _gpu(synth.make_mfcc(4, 300, seed=11))
    eager = gpu_model.extract_x_vec(x)
    torch.cuda.synchronize()
    static_x = x.clone()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        static_out = gpu_model.extract_x_vec(static_x)
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(static_out, eager)
    x2 = _gpu(synth.make_mfcc(4, 300, seed=12))
    static_x.copy_(x2)
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(static_out, gpu_model.extract_x_vec(x2))
    # the packaged form
    gp = gpu_model.graphed(x)
    assert torch.equal(gp(x2), gpu_model.extract_x_vec(x2))
    assert torch.equal(gp(x), eager)
    with pytest.raises(ValueError):
        gp(x[:2])


def test_odd_input_width_and_short_utterances(synth):
    """input_size not a multiple of 4 (row padding kernel) and the shortest legal T=15
    (one pooled frame -> NaN std in the reference too)."""
    import xvector_amd as xa
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in synth.make_state_dict(
        seed=8, input_size=13, hidden_size=40, num_classes=7, x_vector_size=12).items()}
    m = xa.XVectorModel(input_size=13, hidden_size=40, num_classes=7, x_vector_size=12)
    m.load_state_dict(sd)
    m = m.to(DEV)
    p64 = oracle.cast_params(float_params(sd), torch.float64)
    x = torch.from_numpy(np.random.default_rng(3).standard_normal((3, 77, 13), dtype=np.float32))
    assert_parity(m.extract_x_vec(x.to(DEV)), oracle.extract_x_vec(x.double(), p64), 1e-4, "odd widths")
    x16 = x[:, :16]
    assert_parity(m.extract_x_vec(x16.to(DEV)), oracle.extract_x_vec(x16.double(), p64), 1e-4, "T=16")
    out15 = m.extract_x_vec(x[:, :15].to(DEV))
    assert torch.isnan(out15).all()                         # std of one frame is NaN (reference: same)
    with pytest.raises(ValueError):
        m.extract_x_vec(x[:, :14].to(DEV))


# ------------------------------------------------------------------------------- full size
def test_full_size_properties(gpu_model, sd42, synth):
    """BASELINE config 1 size (B=256, T=300): size-independent properties + sampled rows
    against the oracle."""
    x = _gpu(synth.make_mfcc(256, 300, seed=0))
    out = gpu_model.extract_x_vec(x)
    assert out.shape == (256, 512) and torch.isfinite(out).all()
    # determinism: bit-identical on repeat (no atomics / fixed reduction order)
    assert torch.equal(out, gpu_model.extract_x_vec(x))
    # utterance independence: a row does not depend on its batch neighbours
    perm = torch.randperm(256, generator=torch.Generator().manual_seed(1)).to(DEV)
    assert_parity(gpu_model.extract_x_vec(x[perm]), out[perm], 1e-5, "permutation equivariance")
    assert_parity(gpu_model.extract_x_vec(x[37:38]), out[37:38], 1e-5, "batch-of-one")
    # sampled utterances against the fp32 oracle (what the reference computes)
    idx = [0, 1, 127, 128, 255]
    with torch.no_grad():
        ref = oracle.extract_x_vec(x[idx].cpu(), float_params(sd42))
    assert_parity(out[idx], ref, 1e-4, "sampled rows")


def test_full_size_ragged(gpu_model, sd42, synth):
    """BASELINE config 2 (B=256, 200..1000 frames): masked result == per-utterance result."""
    lens = synth.make_lengths(256)
    T = int(lens.max())
    x = synth.make_mfcc(256, T, seed=2)
    out = gpu_model.extract_x_vec(_gpu(x), lengths=lens.tolist())
    assert torch.isfinite(out).all()
    for i in (0, 100, 255, int(lens.argmin()), int(lens.argmax())):
        n = int(lens[i])
        alone = gpu_model.extract_x_vec(_gpu(x[i:i + 1, :n]))
        assert_parity(out[i:i + 1], alone, 1e-5, f"utt {i} len {n}")
    i = int(lens.argmin())
    with torch.no_grad():
        ref = oracle.extract_x_vec(torch.from_numpy(x[i:i + 1, :int(lens[i])]), float_params(sd42))
    assert_parity(out[i:i + 1], ref, 1e-4, "shortest vs oracle")


@pytest.mark.parametrize("precision,tol", [("fp32", 1e-4), ("bf16", 1e-2), ("bf16x3", 1e-4)])
def test_extreme_shapes(sd42, synth, precision, tol):
    """Shapes far from the bench batch: one very long utterance (its pooling partials span ~940
    row groups and every persistent block), thousands of minimal utterances (several utterances
    inside one 32-row group), and a batch at the documented maximum count."""
    import xvector_amd as xa
    m = xa.XVectorModel(precision=precision)
    m.load_state_dict(sd42)
    m = m.to(DEV)
    p32 = float_params(sd42)
    # (1) B=1, T=30 000 (five minutes of speech)
    x = synth.make_mfcc(1, 30000, seed=5)
    with torch.no_grad():
        ref = oracle.extract_x_vec(torch.from_numpy(x), p32)
    assert_parity(m.extract_x_vec(_gpu(x)), ref, tol, "T=30000", elem_tol=2e-2 if precision == "bf16" else None)
    # (2) 3 000 utterances of 16..40 frames, ragged: 2..26 pooled frames each
    rng = np.random.default_rng(9)
    lens = rng.integers(16, 41, 3000)
    xs = synth.make_mfcc(3000, 40, seed=6)
    out = m.extract_x_vec(_gpu(xs), lengths=lens.tolist())
    assert out.shape == (3000, 512) and torch.isfinite(out).all()
    idx = [0, 1, 2, 1499, 2998, 2999, int(lens.argmin()), int(lens.argmax())]
    with torch.no_grad():
        ref = torch.cat([oracle.extract_x_vec(torch.from_numpy(xs[i:i + 1, :int(lens[i])]), p32) for i in idx])
    assert_parity(out[idx], ref, tol, "short ragged", elem_tol=3e-2 if precision == "bf16" else None)
    # (3) the largest batch one call accepts (65 535 utterances), T=16: every row equals the
    # same utterance run alone
    # (bf16x3 addresses its second plane with a 30-bit offset: half as many rows per call)
    reps = 13107 if precision != "bf16x3" else 6000
    big = torch.from_numpy(synth.make_mfcc(5, 16, seed=7)).to(DEV).repeat(reps, 1, 1)
    n_big = 5 * reps
    assert n_big == 65535 or precision == "bf16x3"
    outb = m.extract_x_vec(big)
    alone = m.extract_x_vec(big[:5])
    # bf16 / bf16x3: the large batch runs the 256-channel mapping (tdnn_pp16.hip), five utterances the 128x128 kernel --
    # the same arithmetic type in another summation order; in plain bf16 the large-batch pooling also sums bf16-rounded
    # deviations on the matrix pipe (~1e-4 of a statistic; either result is 9e-4 from fp32).  fp32 has one kernel.
    same = 1e-3 if precision == "bf16" else 1e-5
    assert_parity(outb[:5], alone, same, "max batch head")
    assert_parity(outb[-5:], alone, same, "max batch tail")
    # the same utterance at another place in the batch: equal to fp32 rounding, not bit for bit -- the fused pooling
    # sums deviations from a pivot (the first frame of the 32-row group the rows fall in, csrc/tdnn_common.h), which
    # depends on the position; repeat runs of the same batch ARE bit-identical (test_full_size_properties)
    # (plain bf16 on the large-batch kernel: the pooling pivot is per (block, wave) and the deviations are rounded to bf16 --
    #  tdnn_pp16.hip, SegMx -- so the position shows at ~1e-4, well inside the arithmetic's 1e-2 bar; documented in INTEGRATION.md)
    assert_parity(outb[5:10], outb[n_big - 5:n_big], 1e-5 if precision != "bf16" else same, "same utterances, other batch position")
    # one utterance more than a library call takes: the host module makes two calls of it (round 1 raised here);
    # the C entry point itself still refuses
    over = torch.cat([big, big[:1]], 0) if n_big == 65535 else big
    outo = m.extract_x_vec(over)
    assert outo.shape[0] == over.shape[0] and torch.isfinite(outo).all()
    assert_parity(outo[-1:], alone[:1] if n_big == 65535 else alone[4:5], same, "batch of 65536: last row")
    assert_parity(outo[:5], alone, same, "batch of 65536: head")
    from xvector_amd import hip as _hip
    eng = m._engine(torch.device(DEV))
    import ctypes as C
    rc = _hip.lib.xvec_forward(eng.h, C.c_void_p(over.data_ptr()), None, 65536, 16, 6, 0, C.c_void_p(outo.data_ptr()),
                               C.c_void_p(eng.workspace.data_ptr()), C.c_size_t(eng.workspace.numel()), None)
    assert rc == _hip.ERR_TOO_LARGE and "65535" in _hip.last_error()          # B must be in [1, 65535]: split the batch


@pytest.mark.parametrize("case,precision", [(c, "fp32") for c in range(12)] + [(c, "bf16") for c in (0, 3, 5, 8, 11)]
                         + [(c, "bf16x3") for c in range(12)])
def test_random_model_shapes_vs_oracle(synth, case, precision):
    """Seeded random architectures and batches (widths that are not multiples of the tile sizes, odd
    MFCC counts, with and without BatchNorm, fixed and ragged lengths, all three outputs) against the
    fp64 oracle: the padding and guard paths of every kernel, not only the 24/512/512/1211 model."""
    import xvector_amd as xa
    rng = np.random.default_rng(1000 + case)
    cin = int(rng.integers(5, 41))
    hid = int(rng.choice([24, 32, 72, 96, 128, 136, 200, 256]))
    xv = int(rng.integers(8, 70))
    ncls = int(rng.integers(3, 60))
    bn = bool(rng.integers(0, 2))
    B = int(rng.integers(1, 10))
    T = int(rng.integers(16, 140))
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in synth.make_state_dict(
        seed=2000 + case, input_size=cin, hidden_size=hid, num_classes=ncls, x_vector_size=xv, batch_norm=bn).items()}
    m = xa.XVectorModel(input_size=cin, hidden_size=hid, num_classes=ncls, x_vector_size=xv, batch_norm=bn,
                        precision=precision)
    m.load_state_dict(sd)
    m = m.to(DEV)
    p64 = oracle.cast_params(float_params(sd), torch.float64)
    x = torch.from_numpy(rng.standard_normal((B, T, cin), dtype=np.float32))
    what = f"case {case} {precision}: cin={cin} hid={hid} xv={xv} cls={ncls} bn={bn} B={B} T={T}"
    tol, et = (1e-2, 4e-2) if precision == "bf16" else (1e-4, None)     # bf16x3 answers to the fp32 bar

    def check(got, ref, tag):
        assert_parity(got, ref, tol, what + tag, elem_tol=et)

    check(m.extract_x_vec(x.to(DEV)), oracle.extract_x_vec(x.double(), p64, batch_norm=bn), " xvec6")
    check(m(x.to(DEV)), oracle.forward(x.double(), p64, batch_norm=bn), " logits")
    m.x_vec_extract_layer = 7
    check(m.extract_x_vec(x.to(DEV)), oracle.extract_x_vec(x.double(), p64, layer=7, batch_norm=bn), " xvec7")
    m.x_vec_extract_layer = 6
    if T >= 17 and B > 1:
        lens = rng.integers(16, T + 1, B)
        got = m.extract_x_vec(x.to(DEV), lengths=lens.tolist())
        ref = torch.cat([oracle.extract_x_vec(x[i:i + 1, :int(lens[i])].double(), p64, batch_norm=bn) for i in range(B)])
        check(got, ref, " ragged")


def test_errors_are_loud(gpu_model):
    with pytest.raises(RuntimeError):
        gpu_model.extract_x_vec(torch.zeros(1, 300, 24))           # CPU tensor: no fallback
    with pytest.raises(ValueError):
        gpu_model.extract_x_vec(torch.zeros(1, 300, 23, device=DEV))
    with pytest.raises(ValueError):
        gpu_model.extract_x_vec(torch.zeros(2, 300, 24, device=DEV), lengths=[300, 10])
    gpu_model.train()
    try:
        with pytest.raises(RuntimeError):
            gpu_model(torch.zeros(1, 300, 24, device=DEV))
    finally:
        gpu_model.eval()


# ------------------------------------------------------------------------------- bf16 path
def _bf16_model(sd42):
    import xvector_amd as xa
    m = xa.XVectorModel(precision="bf16")
    m.load_state_dict(sd42)
    return m.to(DEV)


def test_bf16_layers_vs_fp32(gpu_model, sd42, synth):
    """BASELINE config 4: bf16 activations/weights, fp32 accumulation -- each layer within 1e-2
    (norm-wise per frame) of the fp32 layer on the same input.  Measured: 2.4e-3 norm-wise; single
    post-ReLU elements reach 1.05e-2 of (mean|ref|+|ref|), hence 2e-2 element-wise for one layer;
    the end-to-end check below holds the full 1e-2 form (measured 9e-4 / 3e-3)."""
    m16 = _bf16_model(sd42)
    h = _gpu(synth.make_mfcc(3, 150, seed=21))
    for i in range(5):
        ref = gpu_model.time_context_layers[i](h)
        got = m16.time_context_layers[i](h)
        assert_parity(got, ref, 1e-2, f"bf16 layer {i}", elem_tol=2e-2)
        h = ref


@pytest.mark.parametrize("B,T", [(1, 299), (8, 300), (256, 300)])
def test_bf16_full_path_vs_fp32(gpu_model, sd42, synth, B, T):
    m16 = _bf16_model(sd42)
    x = _gpu(synth.make_mfcc(B, T, seed=31))
    assert_parity(m16.extract_x_vec(x), gpu_model.extract_x_vec(x), 1e-2, "bf16 xvec6")
    assert_parity(m16(x), gpu_model(x), 1e-2, "bf16 logits")
    if B == 8:     # and against the reference's own fp32 CPU result
        with torch.no_grad():
            ref = oracle.extract_x_vec(x.cpu(), float_params(sd42))
        assert_parity(m16.extract_x_vec(x), ref, 1e-2, "bf16 xvec6 vs oracle")


def test_bf16_ragged_and_tiny(sd42, synth):
    import xvector_amd as xa
    m16 = _bf16_model(sd42)
    g = load_golden("g5_ragged.npz")
    x = _gpu(synth.make_mfcc(3, 1000, seed=int(g["seed_x"])))
    assert_parity(m16.extract_x_vec(x, lengths=g["lengths"].tolist()), g["xvec6"], 1e-2, "bf16 ragged")
    gt = load_golden("g6_tiny.npz")
    sd = {k[len("bn/w/"):]: torch.from_numpy(gt[k]) for k in gt.files if k.startswith("bn/w/")}
    mt = xa.XVectorModel(input_size=24, hidden_size=32, num_classes=10, x_vector_size=16, precision="bf16")
    mt.load_state_dict(sd)
    mt = mt.to(DEV)
    assert_parity(mt.extract_x_vec(_gpu(gt["bn/x"])), gt["bn/xvec6"], 1e-2, "bf16 tiny")


# ------------------------------------------------------------------------------- bf16x3 path
def _x3_model(sd42):
    import xvector_amd as xa
    m = xa.XVectorModel(precision="bf16x3")
    m.load_state_dict(sd42)
    return m.to(DEV)


@pytest.mark.parametrize("B,T", [(1, 299), (8, 299), (1, 300), (8, 300)])
def test_bf16x3_full_path_vs_golden(sd42, synth, B, T):
    """Two bf16 planes, three bf16 products per k-step: held to the fp32 bar (1e-4) against the
    REFERENCE's own outputs (fixtures g4), all three outputs."""
    g = load_golden("g4_full.npz")
    m = _x3_model(sd42)
    key = f"B{B}_T{T}"
    x = _gpu(synth.make_mfcc(B, T, seed=int(g[key + "_seed_x"])))
    assert_parity(m.extract_x_vec(x), g[key + "_xvec6"], 1e-4, "x3 xvec6")
    assert_parity(m(x), g[key + "_logits"], 1e-4, "x3 logits")
    m.x_vec_extract_layer = 7
    assert_parity(m.extract_x_vec(x), g[key + "_xvec7"], 1e-4, "x3 xvec7")


def test_bf16x3_layers_and_full_size(gpu_model, sd42, synth):
    """Every layer alone against the fp32 kernels (5e-5), the bench batch against them end to end,
    ragged lengths, and the batch-size limit of the mode (plane offsets are 30-bit)."""
    m = _x3_model(sd42)
    x = _gpu(synth.make_mfcc(3, 120, seed=21))
    h32, h3 = x, x
    for l32, l3 in zip(gpu_model.time_context_layers, m.time_context_layers):
        h32, h3 = l32(h32), l3(h32)            # same (fp32) input to both
        assert_parity(h3, h32, 5e-5, "x3 layer", elem_tol=2e-4)
    xb = _gpu(synth.make_mfcc(256, 300, seed=0))
    out = m.extract_x_vec(xb)
    assert_parity(out, gpu_model.extract_x_vec(xb), 2e-5, "x3 vs fp32, B=256")
    assert torch.equal(out, m.extract_x_vec(xb))                      # deterministic
    lens = synth.make_lengths(64)
    xr = _gpu(synth.make_mfcc(64, int(lens.max()), seed=2))
    assert_parity(m.extract_x_vec(xr, lengths=lens.tolist()), gpu_model.extract_x_vec(xr, lengths=lens.tolist()), 2e-5,
                  "x3 ragged")
    # 65 535 x 16 frames exceed the 30-bit plane offsets of ONE library call in this mode: the host module cuts the
    # batch into two calls (round 1 raised here); the C entry point itself refuses, loudly
    big = torch.zeros(65535, 16, 24, device=DEV)
    outb = m.extract_x_vec(big)
    assert outb.shape == (65535, 512) and torch.isfinite(outb).all()
    assert float((outb[0] - outb[-1]).abs().max()) < 1e-5           # same utterance, first and second call
    import ctypes as C
    from xvector_amd import hip as _hip
    eng = m._engine(torch.device(DEV))
    _hip.lib.xvec_workspace_bytes.restype = C.c_size_t
    n = _hip.lib.xvec_workspace_bytes(eng.h, C.c_int64(65535 * 16), C.c_int32(65535))
    ws = torch.empty(n, dtype=torch.uint8, device=DEV)
    rc = _hip.lib.xvec_forward(eng.h, C.c_void_p(big.data_ptr()), None, 65535, 16, 6, 2, C.c_void_p(outb.data_ptr()),
                               C.c_void_p(ws.data_ptr()), C.c_size_t(n), None)
    assert rc == _hip.ERR_TOO_LARGE and "bf16x3" in _hip.last_error()



@pytest.mark.gpu
@pytest.mark.parametrize("B,T", [(63, 300), (100, 300), (128, 300), (160, 300), (70, 517)])
def test_bf16_mid_size_batches(sd42, synth, gpu_model, B, T):
    """Mid-size bf16 batches run the large-batch kernels with few units of 64 frames per CU: blocks of 2, 3 and 5
    units (tiles of 2, 3, 3 + 2 rows of accumulators), partial last units.  Against the fp32 path of the same
    library (bar 1e-2) and, row for row, against the same utterances in a batch of five (the 128x128 kernel)."""
    import xvector_amd as xa
    m = xa.XVectorModel(precision="bf16")
    m.load_state_dict(sd42)
    m = m.to(DEV)
    x = _gpu(synth.make_mfcc(B, T, seed=B + T))
    out = m.extract_x_vec(x)
    assert torch.isfinite(out).all()
    assert_parity(out, gpu_model.extract_x_vec(x), 1e-2, f"bf16 B={B} T={T} vs fp32", elem_tol=2e-2)
    assert torch.equal(out, m.extract_x_vec(x))                                   # deterministic
    idx = [0, 1, B // 2, B - 2, B - 1]
    assert_parity(out[idx], m.extract_x_vec(x[idx]), 2e-4, "rows vs the small-batch kernel")
    lens = np.random.default_rng(B).integers(T // 2, T + 1, B)
    outr = m.extract_x_vec(x, lengths=lens.tolist())
    assert_parity(outr, gpu_model.extract_x_vec(x, lengths=lens.tolist()), 1e-2, "ragged vs fp32", elem_tol=2e-2)


# ------------------------------------------------------------------ plain bf16: BatchNorm deferred into the consumer
@pytest.mark.gpu
def test_bf16_deferred_batchnorm_signs_zeros_and_load_order(sd42, synth):
    """Plain bf16 keeps ReLU outputs between layers and folds each layer's eval BatchNorm (tdnn_layer.py:36-39) into the NEXT
    layer's weights and bias (xvec_api.hip, refold).  The fold must hold for any BatchNorm a checkpoint can carry -- negative
    and exactly-zero gamma, large running means -- and for any order in which the C ABI is given the layers."""
    import ctypes as C
    import xvector_amd as xa
    import xvector_oracle as oracle
    from conftest import float_params
    from xvector_amd import hip as _hip
    sd = {k: v.clone() for k, v in sd42.items()}
    g = torch.Generator().manual_seed(7)
    for i in range(5):
        w = sd[f"time_context_layers.{i}.norm.weight"]
        sign = torch.where(torch.rand(w.shape, generator=g) < 0.3, -1.0, 1.0)
        w.mul_(sign)
        w[torch.rand(w.shape, generator=g) < 0.02] = 0.0                       # channels switched off by gamma = 0
        sd[f"time_context_layers.{i}.norm.running_mean"].add_(2.0 * torch.randn(w.shape, generator=g))
    x = torch.as_tensor(synth.make_mfcc(96, 300, seed=71))
    p64 = oracle.cast_params(float_params(sd), torch.float64)
    idx = [0, 1, 47, 95]
    ref = oracle.extract_x_vec(x[idx].double(), p64)
    ref_pooled = oracle.stat_pool(oracle.time_context_layers(x[idx].double(), p64))
    m = xa.XVectorModel(precision="bf16")
    m.load_state_dict(sd)
    m = m.to(DEV).eval()
    got = m.extract_x_vec(x.to(DEV))
    assert m.last_dispatch() == ["first", "pp", "pp", "pp", "pp"]
    assert_parity(got[idx], ref, 1e-2, "bf16 x-vectors, BatchNorm with negative / zero gamma")
    assert_parity(m.pooled(x.to(DEV))[idx], ref_pooled, 1e-2, "bf16 pooled statistics, BatchNorm with negative / zero gamma")
    # the per-layer entry takes and returns the reference's tensors (BatchNorm applied on both sides)
    h = oracle.tdnn_layer(x[:8].double(), p64, "time_context_layers.0.", oracle.CONTEXTS[0])
    h2 = oracle.tdnn_layer(h, p64, "time_context_layers.1.", oracle.CONTEXTS[1])
    assert_parity(m.time_context_layers[1](h.float().to(DEV)), h2, 1e-2, "bf16 layer 2 alone, zero-gamma input channels", elem_tol=4e-2)
    # the same weights handed to a second handle in REVERSE layer order: every layer's bf16 copies depend on its producer's
    # BatchNorm, whichever of the two arrives first
    eng = m._engine(torch.device(DEV))
    h2nd = C.c_void_p()
    _hip.check(_hip.lib.xvec_create(C.byref(eng.cfg), C.byref(h2nd)))
    try:
        keep = []
        for i in reversed(range(5)):
            layer = m.time_context_layers[i]
            ts = [t.detach().contiguous() for t in (layer.linear.weight, layer.linear.bias, layer.norm.weight, layer.norm.bias,
                                                    layer.norm.running_mean, layer.norm.running_var)]
            keep += ts
            _hip.check(_hip.lib.xvec_load_tdnn(h2nd, i, *[t.data_ptr() for t in ts], layer.norm.eps, None))
        for which, lin in ((_hip.SEG6, m.segment_layer6), (_hip.SEG7, m.segment_layer7), (_hip.OUTPUT, m.output)):
            _hip.check(_hip.lib.xvec_load_affine(h2nd, which, lin.weight.data_ptr(), lin.bias.data_ptr(), None))
        torch.cuda.synchronize()
        xd = x.to(DEV).contiguous()
        n = _hip.lib.xvec_workspace_bytes(h2nd, 96 * 300, 96)
        ws = torch.empty(int(n), dtype=torch.uint8, device=DEV)
        out = torch.empty((96, 512), dtype=torch.float32, device=DEV)
        _hip.check(_hip.lib.xvec_forward(h2nd, xd.data_ptr(), None, 96, 300, _hip.MODE_XVEC6, _hip.BF16, out.data_ptr(),
                                         ws.data_ptr(), ws.numel(), None))
        torch.cuda.synchronize()
        assert torch.equal(out, got), "layers loaded in reverse order give other x-vectors"
    finally:
        _hip.lib.xvec_destroy(h2nd)
