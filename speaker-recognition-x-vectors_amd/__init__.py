"""MI355X-native x-vector embedding extractor (TDNN x5 -> stats pooling -> affine).

Drop-in for the model surface of TorbenHellriegel/Speaker-Recognition-x-vectors
(`XVectorModel.forward / extract_x_vec / stat_pool`, reference main.py:23-94) backed by
hand-written gfx950 HIP kernels behind the C-ABI in include/xvec_hip.h.
"""
from . import synth  # noqa: F401  (numpy only)

__all__ = ["synth", "XVectorModel", "TdnnLayer", "get_time_context", "MfccFrontEnd", "PldaScorer", "hip", "extract",
           "frontend", "scoring"]


def __getattr__(name):
    # torch-dependent parts load lazily so that `synth` stays importable anywhere
    if name in ("XVectorModel", "TdnnLayer", "get_time_context"):
        from . import model
        return getattr(model, name)
    if name == "MfccFrontEnd":
        from . import frontend
        return frontend.MfccFrontEnd
    if name == "PldaScorer":
        from . import scoring
        return scoring.PldaScorer
    if name in ("hip", "model", "extract", "frontend", "scoring"):
        import importlib
        return importlib.import_module("." + name, __name__)
    raise AttributeError(name)
