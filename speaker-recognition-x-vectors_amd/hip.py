"""ctypes binding of libxvec_hip.so (C ABI: include/xvec_hip.h).

The library is the product; there is no CPU or eager-PyTorch fallback.  If the shared
object has not been built (python -c "import __graft_entry__ as g; g.build()") or cannot
be loaded, importing this module raises immediately.
"""
from __future__ import annotations

import ctypes as C
import os

# torch first: it ships its own HIP runtime, and libxvec_hip.so must bind to THAT copy.  Loading the
# library before torch pulls in the system's libamdhip64 as a second runtime in the process, and the
# later one reports "no ROCm-capable device" (seen with build() followed by smoke() in one process).
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
# XVEC_LIB: development override (A/B runs of two builds on one GPU box; diagnostic builds under build/).  It must
# be an absolute path and is announced on stderr, so a stray library can never be picked up silently; the
# default -- and the only thing that ships -- is the in-tree build.
_override = os.environ.get("XVEC_LIB")
if _override:
    import sys
    if not os.path.isabs(_override):
        raise ImportError(f"XVEC_LIB={_override!r} must be an absolute path (unset it to load the in-tree library)")
    print(f"[xvector_amd] XVEC_LIB override: loading {_override}", file=sys.stderr)
LIB_PATH = _override or os.path.join(_HERE, "libxvec_hip.so")

OK, ERR_ARG, ERR_HIP, ERR_STATE, ERR_WORKSPACE, ERR_TOO_LARGE = 0, 1, 2, 3, 4, 5
F32, BF16, BF16X3 = 0, 1, 2
MODE_LOGITS, MODE_POOLED, MODE_XVEC6, MODE_XVEC7 = 0, 5, 6, 7
SEG6, SEG7, OUTPUT = 6, 7, 8
KERNEL_NAMES = {0: None, 1: "tile128", 2: "pp", 3: "first"}      # XVEC_KERNEL_*
TIMING_NAMES = ("tdnn1", "tdnn2", "tdnn3", "tdnn4", "tdnn5_pool", "pool_finalize",
                "segment6", "segment7", "output", "pack")


class XvecError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"xvec_hip error {code}: {msg}")
        self.code = code


class MfccCfg(C.Structure):
    _fields_ = [("samplerate", C.c_int32), ("winlen", C.c_float), ("winstep", C.c_float), ("numcep", C.c_int32),
                ("nfilt", C.c_int32), ("nfft", C.c_int32), ("lowfreq", C.c_float), ("highfreq", C.c_float),
                ("preemph", C.c_float), ("ceplifter", C.c_int32), ("append_energy", C.c_int32),
                ("device", C.c_int32)]


class WsLayout(C.Structure):
    _fields_ = [("act_a", C.c_size_t), ("act_b", C.c_size_t), ("part", C.c_size_t), ("part_cnt", C.c_size_t),
                ("pooled", C.c_size_t), ("bytes", C.c_size_t), ("rows_alloc", C.c_int64), ("part_slots", C.c_int64),
                ("pool_n_pad", C.c_int32), ("hidden_n_pad", C.c_int32), ("num_cu", C.c_int32)]


class Cfg(C.Structure):
    _fields_ = [("input_size", C.c_int32), ("hidden_size", C.c_int32), ("num_classes", C.c_int32),
                ("x_vector_size", C.c_int32), ("batch_norm", C.c_int32), ("device", C.c_int32)]


if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} is missing: build the HIP library first "
        "(python -c 'import __graft_entry__ as g; g.build()' or make -C "
        f"{os.path.join(_HERE, 'csrc')}). There is no CPU fallback.")

lib = C.CDLL(LIB_PATH)

_vp, _i32, _i64, _f32p = C.c_void_p, C.c_int32, C.c_int64, C.c_void_p
_SIGS = {
    "xvec_create": (C.c_int, [C.POINTER(Cfg), C.POINTER(_vp)]),
    "xvec_destroy": (None, [_vp]),
    "xvec_last_error": (C.c_char_p, []),
    "xvec_version": (C.c_char_p, []),
    "xvec_load_tdnn": (C.c_int, [_vp, C.c_int, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p, C.c_float, _vp]),
    "xvec_load_affine": (C.c_int, [_vp, C.c_int, _f32p, _f32p, _vp]),
    "xvec_workspace_bytes": (C.c_size_t, [_vp, _i64, _i32]),
    "xvec_workspace_layout": (C.c_int, [_vp, _i64, _i32, C.POINTER(WsLayout)]),
    "xvec_forward": (C.c_int, [_vp, _f32p, C.POINTER(_i32), _i32, _i32, C.c_int, C.c_int, _f32p, _vp,
                               C.c_size_t, _vp]),
    "xvec_forward_packed": (C.c_int, [_vp, _f32p, C.POINTER(_i64), _i32, C.c_int, C.c_int, _f32p, _vp,
                                      C.c_size_t, _vp]),
    "xvec_tdnn_layer": (C.c_int, [_vp, C.c_int, _f32p, _i32, _i32, C.c_int, _f32p, _vp, C.c_size_t, _vp]),
    "xvec_tdnn_pool_layer": (C.c_int, [_vp, _f32p, _i32, _i32, C.c_int, _f32p, _vp, C.c_size_t, _vp]),
    "xvec_stat_pool": (C.c_int, [_f32p, _vp, _i32, _i32, _i32, _f32p, _vp]),
    "xvec_affine": (C.c_int, [_vp, C.c_int, _f32p, _i32, C.c_int, _f32p, _vp]),
    "xvec_set_profiling": (C.c_int, [_vp, C.c_int]),
    "xvec_get_timings": (C.c_int, [_vp, C.POINTER(C.c_float), C.POINTER(C.c_int)]),
    "xvec_get_dispatch": (C.c_int, [_vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "xvec_mfcc_create": (C.c_int, [C.POINTER(MfccCfg), C.POINTER(_vp)]),
    "xvec_mfcc_destroy": (None, [_vp]),
    "xvec_mfcc_last_error": (C.c_char_p, []),
    "xvec_mfcc_frames": (C.c_int32, [_vp, _i64]),
    "xvec_mfcc_kernel_form": (C.c_int32, [_vp]),
    "xvec_mfcc": (C.c_int, [_vp, _f32p, _i32, _i64, _f32p, _vp]),
    "xvec_mfcc_i16": (C.c_int, [_vp, _vp, C.c_float, _i32, _i64, _f32p, _vp]),
    # include/xvec_score.h
    "xvec_score_last_error": (C.c_char_p, []),
    "xvec_gemm_nt_f64": (C.c_int, [_vp, _i64, _vp, _i64, _i64, _i64, _i32, _vp, _vp, C.c_double, C.c_double, _vp,
                                   _i64, _vp]),
    "xvec_score_workspace_bytes": (C.c_size_t, [_i64, _i64, _i32]),
    "xvec_plda_score": (C.c_int, [_vp, _i64, _vp, _i64, _i32, _vp, _vp, _vp, C.c_double, C.c_double, _vp, _vp,
                                  C.c_size_t, _vp]),
    "xvec_plda_score_lowrank": (C.c_int, [_vp, _i64, _vp, _i64, _i32, _i32, _vp, _vp, _vp, C.c_double, C.c_double, _vp,
                                          _vp, C.c_size_t, _vp]),
    "xvec_cosine_score": (C.c_int, [_vp, _i64, _vp, _i64, _i32, _vp, _vp, C.c_size_t, _vp]),
}
EXPORTS = tuple(_SIGS)
for _name, (_res, _args) in _SIGS.items():
    _fn = getattr(lib, _name)      # AttributeError here = header/library mismatch
    _fn.restype = _res
    _fn.argtypes = _args


def last_error() -> str:
    return lib.xvec_last_error().decode()


def check(rc: int):
    if rc != OK:
        raise XvecError(rc, last_error())


def version() -> str:
    return lib.xvec_version().decode()
