"""CPU oracle for the MFCC front end (next-row N3).  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED: the reference computes its features with the third-party package
`python_speech_features` (`from python_speech_features import mfcc`, reference dataset.py:8;
call site dataset.py:128:  mfcc(sample, 16000, numcep=24, nfilt=26, nfft=512)).  That package
(latest release 0.6, not pinned in requirements.txt) is not installed in the build image and
cannot be fetched, and the reference holds no test or golden vector for it.  This file restates
the package's published algorithm (python_speech_features/base.py and sigproc.py, release 0.6)
in numpy float64, function by function; it has NOT been checked against the package itself.
What is checked (tests/test_mfcc.py): internal identities of the restatement (power spectrum
vs a direct DFT, filterbank shape and partition-of-unity properties, DCT orthonormality), the
frame count the reference relies on (3 s at 16 kHz -> 299 frames, main.py:113 / SURVEY §2), and
the HIP kernel against this file.
"""
from __future__ import annotations

import decimal
import math

import numpy as np


def _round_half_up(number):
    # sigproc.round_half_up
    return int(decimal.Decimal(number).quantize(decimal.Decimal("1"), rounding=decimal.ROUND_HALF_UP))


def preemphasis(signal, coeff=0.95):
    """sigproc.preemphasis: y[0] = x[0], y[n] = x[n] - coeff*x[n-1]."""
    return np.append(signal[0], signal[1:] - coeff * signal[:-1])


def framesig(sig, frame_len, frame_step, winfunc=lambda x: np.ones((x,))):
    """sigproc.framesig (non-strided path): zero-pad the tail so the last frame is whole."""
    slen = len(sig)
    frame_len = int(_round_half_up(frame_len))
    frame_step = int(_round_half_up(frame_step))
    if slen <= frame_len:
        numframes = 1
    else:
        numframes = 1 + int(math.ceil((1.0 * slen - frame_len) / frame_step))
    padlen = int((numframes - 1) * frame_step + frame_len)
    padsignal = np.concatenate((sig, np.zeros((padlen - slen,))))
    idx = np.tile(np.arange(0, frame_len), (numframes, 1)) + \
        np.tile(np.arange(0, numframes * frame_step, frame_step), (frame_len, 1)).T
    frames = padsignal[idx.astype(np.int32)]
    return frames * np.tile(winfunc(frame_len), (numframes, 1))


def powspec(frames, NFFT):
    """sigproc.powspec: 1/NFFT * |rfft(frame, NFFT)|^2 (frames longer than NFFT are truncated)."""
    return 1.0 / NFFT * np.square(np.absolute(np.fft.rfft(frames, NFFT)))


def hz2mel(hz):
    return 2595 * np.log10(1 + hz / 700.0)


def mel2hz(mel):
    return 700 * (10 ** (mel / 2595.0) - 1)


def get_filterbanks(nfilt=20, nfft=512, samplerate=16000, lowfreq=0, highfreq=None):
    """base.get_filterbanks: nfilt triangular filters on FFT bins, [nfilt, nfft//2+1]."""
    highfreq = highfreq or samplerate / 2
    melpoints = np.linspace(hz2mel(lowfreq), hz2mel(highfreq), nfilt + 2)
    bins = np.floor((nfft + 1) * mel2hz(melpoints) / samplerate)
    fbank = np.zeros([nfilt, nfft // 2 + 1])
    for j in range(0, nfilt):
        for i in range(int(bins[j]), int(bins[j + 1])):
            fbank[j, i] = (i - bins[j]) / (bins[j + 1] - bins[j])
        for i in range(int(bins[j + 1]), int(bins[j + 2])):
            fbank[j, i] = (bins[j + 2] - i) / (bins[j + 2] - bins[j + 1])
    return fbank


def dct2_ortho(x, ncoef):
    """scipy.fftpack.dct(x, type=2, axis=1, norm='ortho')[:, :ncoef] written out."""
    n = x.shape[1]
    k = np.arange(ncoef)[:, None]
    m = np.arange(n)[None, :]
    basis = np.cos(np.pi * k * (2 * m + 1) / (2.0 * n))
    scale = np.full((ncoef, 1), math.sqrt(2.0 / n))
    scale[0, 0] = math.sqrt(1.0 / n)
    return x @ (basis * scale).T


def lifter_coeffs(ncoef, L=22):
    n = np.arange(ncoef)
    return 1 + (L / 2.0) * np.sin(np.pi * n / L)


def fbank(signal, samplerate=16000, winlen=0.025, winstep=0.01, nfilt=26, nfft=512, lowfreq=0, highfreq=None,
          preemph=0.97):
    """base.fbank with the default rectangular window."""
    highfreq = highfreq or samplerate / 2
    signal = preemphasis(signal, preemph)
    frames = framesig(signal, winlen * samplerate, winstep * samplerate)
    pspec = powspec(frames, nfft)
    energy = np.sum(pspec, 1)
    energy = np.where(energy == 0, np.finfo(float).eps, energy)
    fb = get_filterbanks(nfilt, nfft, samplerate, lowfreq, highfreq)
    feat = np.dot(pspec, fb.T)
    feat = np.where(feat == 0, np.finfo(float).eps, feat)
    return feat, energy


def mfcc(signal, samplerate=16000, winlen=0.025, winstep=0.01, numcep=13, nfilt=26, nfft=512, lowfreq=0,
         highfreq=None, preemph=0.97, ceplifter=22, appendEnergy=True):
    """base.mfcc.  The reference calls mfcc(x, 16000, numcep=24, nfilt=26, nfft=512) -> [frames, 24]
    float64 (reference dataset.py:128)."""
    feat, energy = fbank(np.asarray(signal, dtype=np.float64), samplerate, winlen, winstep, nfilt, nfft, lowfreq,
                         highfreq, preemph)
    feat = np.log(feat)
    feat = dct2_ortho(feat, numcep)
    if ceplifter > 0:
        feat = feat * lifter_coeffs(numcep, ceplifter)
    if appendEnergy:
        feat[:, 0] = np.log(energy)
    return feat


def num_frames(n_samples, samplerate=16000, winlen=0.025, winstep=0.01):
    fl, fs = _round_half_up(winlen * samplerate), _round_half_up(winstep * samplerate)
    return 1 if n_samples <= fl else 1 + int(math.ceil((1.0 * n_samples - fl) / fs))
