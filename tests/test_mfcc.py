"""MFCC front end (next-row N3).  The package the reference uses (python_speech_features) is not
installed here, so parity is UNPINNED: the oracle restates its published algorithm.  CPU tests
check the restatement's internal identities; the GPU test checks the HIP kernel against it."""
import numpy as np
import pytest
import torch

import mfcc_oracle as mo


def _speechlike(n, seed):
    """Synthetic waveform with a speech-like dynamic range: decaying harmonics + noise floor,
    amplitude-modulated, scaled like 16-bit PCM divided down (the reference min/max-normalises)."""
    rng = np.random.default_rng(seed)
    t = np.arange(n) / 16000.0
    f0 = 110 + 40 * np.sin(2 * np.pi * 1.3 * t)
    x = sum((0.6 ** k) * np.sin(2 * np.pi * np.cumsum(f0 * (k + 1)) / 16000.0) for k in range(12))
    x = x * (0.55 + 0.45 * np.sin(2 * np.pi * 3.1 * t)) + 0.01 * rng.standard_normal(n)
    return (x / np.abs(x).max()).astype(np.float32)


def test_frame_count_matches_reference_shape():
    # 3 s at 16 kHz -> 299 frames x 24 coefficients (reference main.py:113, dataset.py:204)
    assert mo.num_frames(48000) == 299
    f = mo.mfcc(_speechlike(48000, 0), 16000, numcep=24, nfilt=26, nfft=512)
    assert f.shape == (299, 24) and f.dtype == np.float64 and np.isfinite(f).all()
    assert mo.num_frames(400) == 1 and mo.num_frames(401) == 2 and mo.num_frames(100) == 1


def test_oracle_internal_identities():
    rng = np.random.default_rng(1)
    x = rng.standard_normal(2050)
    # pre-emphasis definition
    y = mo.preemphasis(x, 0.97)
    assert y[0] == x[0] and np.allclose(y[1:], x[1:] - 0.97 * x[:-1])
    # framing: 400-sample frames every 160 samples, zero padded tail
    fr = mo.framesig(y, 400, 160)
    assert fr.shape == (12, 400) and np.array_equal(fr[3], y[480:880])
    assert np.array_equal(fr[-1, :290], y[1760:]) and not fr[-1, 290:].any()      # zero-padded tail
    # power spectrum vs an explicit DFT of the zero-padded frame
    k = np.arange(257)[:, None] * np.arange(512)[None, :]
    dft = np.exp(-2j * np.pi * k / 512) @ np.concatenate([fr[2], np.zeros(112)])
    assert np.allclose(mo.powspec(fr[2:3], 512)[0], np.abs(dft) ** 2 / 512, rtol=1e-10)
    # filterbank: 26 triangles, peaks of 1 at the interior bin edges, neighbours sum to 1 between peaks
    fb = mo.get_filterbanks(26, 512, 16000)
    assert fb.shape == (26, 257) and fb.min() >= 0 and np.isclose(fb.max(), 1.0)
    inner = fb[:, int(np.argmax(fb[0])):int(np.argmax(fb[-1])) + 1].sum(0)
    assert np.allclose(inner, 1.0)
    # orthonormal DCT-II equals scipy's
    from scipy.fftpack import dct
    z = rng.standard_normal((5, 26))
    assert np.allclose(mo.dct2_ortho(z, 24), dct(z, type=2, axis=1, norm="ortho")[:, :24])
    assert np.allclose(mo.lifter_coeffs(24)[:3], [1.0, 1 + 11 * np.sin(np.pi / 22), 1 + 11 * np.sin(2 * np.pi / 22)])


def test_c0_is_log_energy():
    x = _speechlike(8000, 3)
    f = mo.mfcc(x, 16000, numcep=24, nfilt=26, nfft=512)
    _, energy = mo.fbank(x.astype(np.float64), 16000, nfilt=26, nfft=512)
    assert np.allclose(f[:, 0], np.log(energy))


@pytest.mark.gpu
@pytest.mark.parametrize("n_samples,B", [(48000, 3), (16000, 1), (401, 2), (250, 1), (47999, 2), (48000, 64), (1000, 300)])
def test_mfcc_kernel_vs_oracle(n_samples, B):
    import xvector_amd as xa
    from conftest import assert_parity
    fe = xa.MfccFrontEnd()
    waves = np.stack([_speechlike(n_samples, 10 + i) * (0.2 + 0.4 * i) for i in range(B)])
    got = fe(torch.from_numpy(waves).to("cuda:0")).cpu().numpy()
    ref = np.stack([mo.mfcc(w, 16000, numcep=24, nfilt=26, nfft=512) for w in waves])
    assert got.shape == ref.shape == (B, mo.num_frames(n_samples), 24)
    # fp32 kernel vs float64 oracle: per-frame norm-wise 1e-4 (the model casts features to fp32
    # anyway, reference main.py:137); element-wise 1e-3 of the mean magnitude (low-energy bands)
    assert_parity(got, ref, 1e-4, "mfcc", elem_tol=1e-3)


def _mfcc_random_cases(n, seed):
    rng = np.random.default_rng(seed)
    return [(int(rng.integers(81, 30000)), int(rng.integers(1, 24))) for _ in range(n)]


@pytest.mark.gpu
@pytest.mark.parametrize("n_samples,B", _mfcc_random_cases(12, 99))
def test_mfcc_random_sizes(n_samples, B):
    """Sizes nobody chose: frames are numbered through the batch in tiles of 16 and paired two to a transform, so
    B x frames mod 32, the ragged last frame and the pair that straddles two utterances all vary."""
    import xvector_amd as xa
    from conftest import assert_parity
    fe = xa.MfccFrontEnd()
    waves = np.stack([_speechlike(n_samples, 500 + i) * (0.1 + 0.05 * i) for i in range(B)])
    got = fe(torch.from_numpy(waves).to("cuda:0")).cpu().numpy()
    ref = np.stack([mo.mfcc(w, 16000, numcep=24, nfilt=26, nfft=512) for w in waves])
    assert_parity(got, ref, 1e-4, f"mfcc n={n_samples} B={B}", elem_tol=1e-3)


@pytest.mark.gpu
@pytest.mark.parametrize("kw", [dict(nfft=1024), dict(nfft=256, winlen=0.016), dict(nfft=2048, winlen=0.05, nfilt=40, numcep=13),
                                dict(nfft=512, nfilt=40, numcep=20), dict(nfft=512, nfilt=20, numcep=13, lowfreq=300, highfreq=3400),
                                dict(nfft=512, winlen=0.04), dict(nfft=512, appendEnergy=False, ceplifter=0, preemph=0.0),
                                dict(nfft=512, winlen=0.005, winstep=0.01), dict(nfft=1024, winlen=0.005, winstep=0.01)])
def test_mfcc_other_configurations(kw):
    """The package's other keyword arguments: nfft != 512 and nfilt > 32 take the general kernel, the rest the
    nfft = 512 kernel with other tables (a frame longer than nfft is truncated, sigproc.powspec)."""
    import xvector_amd as xa
    from conftest import assert_parity
    fe = xa.MfccFrontEnd(**kw)
    waves = np.stack([_speechlike(20000, 70 + i) * (0.3 + 0.3 * i) for i in range(3)])
    got = fe(torch.from_numpy(waves).to("cuda:0")).cpu().numpy()
    okw = {k: v for k, v in kw.items()}
    ref = np.stack([mo.mfcc(w, 16000, **{"numcep": 24, "nfilt": 26, "nfft": 512, **okw}) for w in waves])
    assert got.shape == ref.shape
    assert_parity(got, ref, 1e-4, f"mfcc {kw}", elem_tol=1e-3)


@pytest.mark.gpu
@pytest.mark.parametrize("nfft", [512, 1024])
def test_mfcc_digital_silence_inside_a_signal(nfft):
    """Frames of exact zeros between live frames: the package puts eps where a power is 0 (log = -36.04).  The kernels
    transform two frames as one complex signal; the silent one must not pick up the other's rounding noise."""
    import xvector_amd as xa
    from conftest import assert_parity
    fe = xa.MfccFrontEnd(nfft=nfft)
    waves = np.stack([_speechlike(16000, 90 + i) for i in range(2)])
    waves[0, 5000:9000] = 0.0
    waves[1, 12000:] = 0.0
    got = fe(torch.from_numpy(waves).to("cuda:0")).cpu().numpy()
    ref = np.stack([mo.mfcc(w, 16000, numcep=24, nfilt=26, nfft=nfft) for w in waves])
    silent = np.isclose(ref[..., 0], np.log(np.finfo(float).eps))
    assert silent.sum() > 20                      # the test does contain all-zero frames
    assert_parity(got, ref, 1e-4, f"mfcc with digital silence, nfft={nfft}", elem_tol=1e-3)


@pytest.mark.gpu
def test_mfcc_feeds_the_extractor(gpu_model):
    """waveforms -> MFCC -> x-vectors entirely on the GPU, shapes as in the reference (299 x 24)."""
    import xvector_amd as xa
    fe = xa.MfccFrontEnd()
    waves = torch.from_numpy(np.stack([_speechlike(48000, 40 + i) for i in range(4)])).to("cuda:0")
    feats = fe(waves)
    assert feats.shape == (4, 299, 24)
    xv = gpu_model.extract_x_vec(feats)
    assert xv.shape == (4, 512) and torch.isfinite(xv).all()


@pytest.mark.gpu
def test_mfcc_silence_and_errors():
    import xvector_amd as xa
    fe = xa.MfccFrontEnd()
    out = fe(torch.zeros(1, 1600, device="cuda:0")).cpu().numpy()
    ref = mo.mfcc(np.zeros(1600), 16000, numcep=24, nfilt=26, nfft=512)      # zeros -> eps -> finite logs
    assert np.isfinite(out).all() and np.allclose(out[0], ref, rtol=1e-4, atol=1e-3)
    with pytest.raises(RuntimeError):
        fe(torch.zeros(1, 1600))
    with pytest.raises(Exception):
        xa.MfccFrontEnd(nfft=500)


@pytest.mark.gpu
@pytest.mark.parametrize("precision,tol", [("bf16", 1e-2), ("fp32", 1e-4)])
def test_waveforms_to_x_vectors_at_the_bench_size(sd42, precision, tol):
    """The whole product pipeline of `bench.py --workload wave` against the two oracles chained, at ITS size: 256 x 48 000
    samples -> MfccFrontEnd (T = 299: 285 pooled frames, an odd row count for the large-batch kernels) -> extract_x_vec,
    eight sampled utterances against mfcc_oracle.mfcc -> float32 (the reference's samples.float(), main.py:137) ->
    xvector_oracle.extract_x_vec in float64.  VERDICT r05 item 4."""
    import xvector_amd as xa
    import xvector_oracle as xo
    from conftest import assert_parity, float_params
    base = [_speechlike(48000, 200 + i) for i in range(16)]
    waves = np.stack([base[i % 16] * (0.06 + 0.94 * (i // 16) / 15.0) for i in range(256)]).astype(np.float32)
    fe = xa.MfccFrontEnd()
    m = xa.XVectorModel(precision=precision)
    m.load_state_dict(sd42)
    m = m.to("cuda:0").eval()
    feats = fe(torch.from_numpy(waves).to("cuda:0"))
    assert feats.shape == (256, 299, 24)
    got = m.extract_x_vec(feats)
    if precision == "bf16":
        assert m.last_dispatch() == ["first", "pp", "pp", "pp", "pp"]
    idx = [0, 1, 17, 100, 128, 200, 254, 255]
    ref_feats = np.stack([mo.mfcc(waves[i], 16000, numcep=24, nfilt=26, nfft=512) for i in idx])
    assert_parity(feats[idx], ref_feats, 1e-4, "mfcc at the bench size", elem_tol=1e-3)
    p64 = xo.cast_params(float_params(sd42), torch.float64)
    ref = xo.extract_x_vec(torch.from_numpy(ref_feats).float().double(), p64)
    assert_parity(got[idx], ref, tol, f"waveforms -> x-vectors, {precision}")


@pytest.mark.gpu
@pytest.mark.parametrize("kw,n_samples,B,scale", [(dict(), 48000, 5, 1.0), (dict(), 16001, 3, 1.0 / 32768), (dict(), 401, 2, 0.37),
                                                   (dict(nfft=1024), 9000, 2, 1.0), (dict(nfft=512, nfilt=40, numcep=20), 9000, 2, 3.0e-5)])
def test_mfcc_int16_input_equals_float_input_bit_for_bit(kw, n_samples, B, scale):
    """16-bit PCM as scipy.io.wavfile.read yields it (reference dataset.py:125), converted inside the kernel as (float)s * scale:
    one fp32 rounding, exactly what the float path is given for `pcm.float() * scale` -- so the two results are EQUAL, on both
    kernels (nfft = 512 and the general one), for any scale; and both match the oracle on the converted signal."""
    import xvector_amd as xa
    from conftest import assert_parity
    fe = xa.MfccFrontEnd(**kw)
    pcm = np.stack([np.round(_speechlike(n_samples, 300 + i) * (3000.0 + 9000.0 * i)) for i in range(B)]).astype(np.int16)
    pcm[0, :3] = [-32768, 32767, -1]                           # the range's ends
    t = torch.from_numpy(pcm).to("cuda:0")
    got_i = fe(t, scale=scale)
    as_float = t.float() * torch.tensor(scale, dtype=torch.float32)
    got_f = fe(as_float)
    assert got_i.dtype == torch.float32 and torch.equal(got_i, got_f)
    okw = {"numcep": 24, "nfilt": 26, "nfft": 512, **kw}
    ref = np.stack([mo.mfcc(w, 16000, **okw) for w in as_float.cpu().numpy()])
    assert_parity(got_i.cpu().numpy(), ref, 1e-4, f"mfcc of int16 PCM {kw}", elem_tol=1e-3)


@pytest.mark.gpu
def test_first_frame_previous_samples():
    """Pre-emphasis at a signal's start, sample by sample: y[0] = x[0] (no previous sample), y[64 a] = x[64 a] - 0.97 x[64 a - 1].
    The nfft = 512 kernel addresses x[n - 1] of frame 0 as lane offset -4 (-2 for 16-bit PCM) plus the immediate 256 a, which is
    in range for a >= 1 and must read as zero for a = 0 (csrc/mfcc.hip, MF_LOAD_FRAME).  Spikes at exactly the samples 64 a - 1
    and 64 a make a wrong neighbour visible in frame 0's coefficients."""
    import xvector_amd as xa
    from conftest import assert_parity
    fe = xa.MfccFrontEnd()
    for variant in range(3):
        x = 0.01 * np.random.default_rng(variant).standard_normal((2, 4000)).astype(np.float32)
        for a in range(1, 7):
            x[0, 64 * a - 1] += 1.0 + 0.1 * a                  # the previous sample of lane 0's a-th value
            if variant == 1:
                x[0, 64 * a] -= 0.7
        if variant == 2:
            x[:, 0] = 2.5                                      # a large first sample: nothing may be subtracted from it
        got = fe(torch.from_numpy(x).to("cuda:0")).cpu().numpy()
        ref = np.stack([mo.mfcc(w, 16000, numcep=24, nfilt=26, nfft=512) for w in x])
        assert_parity(got[:, :3], ref[:, :3], 1e-4, f"first frames, variant {variant}", elem_tol=1e-3)
        pcm = torch.from_numpy(np.round(x * 8000).astype(np.int16)).to("cuda:0")
        got16 = fe(pcm, scale=1.0 / 8000).cpu().numpy()
        ref16 = np.stack([mo.mfcc(w, 16000, numcep=24, nfilt=26, nfft=512) for w in (pcm.float() / 8000).cpu().numpy()])
        assert_parity(got16[:, :3], ref16[:, :3], 1e-4, f"first frames, 16-bit PCM, variant {variant}", elem_tol=1e-3)


@pytest.mark.gpu
def test_kernel_form_of_a_plan():
    """xvec_mfcc_kernel_form: the reference's call (dataset.py:128) takes the nfft-512 kernel with the BANDED filterbank (2),
    nfft != 512 or more than 32 filters the general kernel (0); a silent fall back of the reference's configuration to a slower
    form would otherwise only show in the bench line."""
    import xvector_amd as xa
    assert xa.MfccFrontEnd().kernel_form() == 2
    assert xa.MfccFrontEnd(nfft=1024).kernel_form() == 0
    assert xa.MfccFrontEnd(nfft=512, nfilt=40, numcep=20).kernel_form() == 0
    assert xa.MfccFrontEnd(nfft=512, appendEnergy=False, ceplifter=0, preemph=0.0).kernel_form() == 2


@pytest.mark.gpu
@pytest.mark.parametrize("kw", [dict(), dict(nfft=512, nfilt=20, numcep=13, lowfreq=300, highfreq=3400), dict(nfft=512, winlen=0.04),
                                dict(nfft=512, nfilt=32, numcep=32)])
def test_dense_and_banded_filterbank_agree(kw, monkeypatch):
    """The nfft-512 kernel has the mel filterbank in two forms: banded 4 x 4 x 1 products (v_mfma_f32_4x4x1_16b_f32, bins in
    groups whose filters fit a window of four) and, for a filterbank the grouping cannot hold, dense 16 x 16 x 16 products.
    XVEC_MFCC_FILTERBANK=dense (read at create time) forces the second: both against the oracle, and against each other far
    inside the bar -- the same weights in another summation order."""
    import xvector_amd as xa
    from conftest import assert_parity
    waves = np.stack([_speechlike(20000, 170 + i) * (0.2 + 0.4 * i) for i in range(3)])
    w = torch.from_numpy(waves).to("cuda:0")
    fe_auto = xa.MfccFrontEnd(**kw)
    monkeypatch.setenv("XVEC_MFCC_FILTERBANK", "dense")
    fe_dense = xa.MfccFrontEnd(**kw)
    monkeypatch.delenv("XVEC_MFCC_FILTERBANK")
    assert fe_dense.kernel_form() == 1 and fe_auto.kernel_form() in (1, 2)
    if not kw:
        assert fe_auto.kernel_form() == 2
    ref = np.stack([mo.mfcc(x, 16000, **{"numcep": 24, "nfilt": 26, "nfft": 512, **kw}) for x in waves])
    got_a, got_d = fe_auto(w).cpu().numpy(), fe_dense(w).cpu().numpy()
    assert_parity(got_a, ref, 1e-4, f"mfcc {kw}, form {fe_auto.kernel_form()}", elem_tol=1e-3)
    assert_parity(got_d, ref, 1e-4, f"mfcc {kw}, dense filterbank", elem_tol=1e-3)
    assert_parity(got_a, got_d, 2e-6, f"mfcc {kw}: banded against dense", elem_tol=1e-4)
    # 16-bit PCM through the same form: bit for bit the float path (one instantiation per form)
    pcm = (torch.from_numpy(waves) * 20000).round().clamp(-32768, 32767).to(torch.int16).to("cuda:0")
    assert torch.equal(fe_auto(pcm, scale=1.0 / 20000), fe_auto(pcm.float() * (1.0 / 20000)))
    assert torch.equal(fe_dense(pcm, scale=1.0 / 20000), fe_dense(pcm.float() * (1.0 / 20000)))


@pytest.mark.gpu
@pytest.mark.parametrize("form", ["auto", "dense"])
def test_repeated_launches_are_bit_identical(form, monkeypatch):
    """The nfft-512 kernel keeps a tile's second pairs of power rows and its partial sums inside the waves' exchange regions and
    runs five blocks per CU: a missing barrier there would show as a result that depends on timing.  Forty launches per shape,
    float and 16-bit PCM input, every one equal to the first bit for bit (2 000 launches were run once by hand: 0 differences)."""
    import xvector_amd as xa
    if form == "dense":
        monkeypatch.setenv("XVEC_MFCC_FILTERBANK", "dense")
    fe = xa.MfccFrontEnd()
    monkeypatch.delenv("XVEC_MFCC_FILTERBANK", raising=False)
    assert fe.kernel_form() == (1 if form == "dense" else 2)
    gen = torch.Generator(device="cuda:0")
    gen.manual_seed(5)
    for B, n in ((256, 48000), (1000, 4000), (7, 401)):
        w = 0.1 * torch.randn(B, n, device="cuda:0", generator=gen)
        pcm = (w * 20000).round().clamp(-32768, 32767).to(torch.int16)
        ref, ref_i = fe(w).clone(), fe(pcm, scale=1 / 20000).clone()
        assert torch.isfinite(ref).all()
        for _ in range(40):
            assert torch.equal(fe(w), ref) and torch.equal(fe(pcm, scale=1 / 20000), ref_i), (form, B, n)

