"""MFCC front end on the GPU (next-row N3): the features the reference computes per utterance on
the CPU with `python_speech_features.mfcc(signal, 16000, numcep=24, nfilt=26, nfft=512)`
(reference dataset.py:128), for a batch of equal-length waveforms in one kernel launch.

    fe = MfccFrontEnd()                       # the reference's parameters
    feats = fe(waves)                         # waves: float tensor [B, n_samples] on the HIP device
    xvecs = model.extract_x_vec(feats)        # [B, 299, 24] for 3 s at 16 kHz

Parity with the package itself is unpinned (it is not installed in the build image); the kernel is
checked against oracle/mfcc_oracle.py, a restatement of the package's published algorithm.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import hip as _hip


class MfccFrontEnd:
    def __init__(self, samplerate=16000, winlen=0.025, winstep=0.01, numcep=24, nfilt=26, nfft=512, lowfreq=0,
                 highfreq=None, preemph=0.97, ceplifter=22, appendEnergy=True, device="cuda:0"):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("MfccFrontEnd runs on a HIP device only (no CPU path)")
        self.numcep = numcep
        cfg = _hip.MfccCfg(samplerate, winlen, winstep, numcep, nfilt, nfft, float(lowfreq), float(highfreq or 0),
                           preemph, int(ceplifter), int(bool(appendEnergy)),
                           self.device.index if self.device.index is not None else torch.cuda.current_device())
        plan = C.c_void_p()
        rc = _hip.lib.xvec_mfcc_create(C.byref(cfg), C.byref(plan))
        if rc != _hip.OK:
            raise _hip.XvecError(rc, _hip.lib.xvec_mfcc_last_error().decode())
        self._plan = plan

    def __del__(self):
        plan, self._plan = getattr(self, "_plan", None), None
        if plan:
            try:
                _hip.lib.xvec_mfcc_destroy(plan)
            except Exception:
                pass

    def kernel_form(self) -> int:
        """0: the general kernel; 1: the nfft-512 kernel, dense filterbank products; 2: the same, banded filterbank
        (xvec_mfcc_kernel_form; what tests and benchmarks name the measured kernel by)."""
        return int(_hip.lib.xvec_mfcc_kernel_form(self._plan))

    def num_frames(self, n_samples: int) -> int:
        return int(_hip.lib.xvec_mfcc_frames(self._plan, n_samples))

    def __call__(self, waves: torch.Tensor, scale: float = 1.0) -> torch.Tensor:
        """waves: float [B, n_samples], or torch.int16 PCM (what scipy.io.wavfile.read yields, reference dataset.py:125),
        converted inside the kernel as (float)s * scale -- bit for bit the result for `waves.float() * scale`, at half the
        bytes over PCIe.  `scale` applies to int16 input only (1.0 = the raw PCM range the reference's call sees)."""
        if not waves.is_cuda:
            raise RuntimeError(f"MfccFrontEnd: expected a tensor on a HIP device, got {waves.device}")
        if waves.dim() == 1:
            waves = waves[None]
        if waves.dim() != 2 or waves.shape[1] < 1:
            raise ValueError(f"MfccFrontEnd: expected waves[B, n_samples], got {tuple(waves.shape)}")
        i16 = waves.dtype == torch.int16
        w = waves.detach().contiguous() if i16 else waves.detach().float().contiguous()
        B, n = w.shape
        out = torch.empty((B, self.num_frames(n), self.numcep), dtype=torch.float32, device=w.device)
        with torch.cuda.device(w.device):
            stream = torch.cuda.current_stream(w.device).cuda_stream
            if i16:
                rc = _hip.lib.xvec_mfcc_i16(self._plan, w.data_ptr(), float(scale), B, n, out.data_ptr(), stream)
            else:
                rc = _hip.lib.xvec_mfcc(self._plan, w.data_ptr(), B, n, out.data_ptr(), stream)
        if rc != _hip.OK:
            raise _hip.XvecError(rc, _hip.lib.xvec_mfcc_last_error().decode())
        return out
