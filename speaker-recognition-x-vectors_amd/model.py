"""Host-side mirror of the reference's model surface for the extraction path.

Same names, constructor arguments, state_dict keys and call signatures as
`tdnn_layer.TdnnLayer` / `get_time_context` (reference tdnn_layer.py:5-60) and
`main.XVectorModel` (reference main.py:23-94), so `extract_x_vectors` in the reference's
driver is a drop-in: `test_step` calls `self.extract_x_vec(samples.float())`
(main.py:137) and gets the same [B, x_vector_size] fp32 tensor back.

Every forward here runs on the MI355X through libxvec_hip.so (ctypes, raw device
pointers, torch's current HIP stream).  PyTorch is plumbing only: device memory,
streams, parameter containers.  There is NO eager / CPU fallback: a CPU tensor, a
missing library or training mode raise.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence

import torch
import torch.nn as nn

from . import hip as _hip

TOTAL_CONTEXT = 14
POOL_CHANNELS = 1500
_CONTEXTS = [[-2, -1, 0, 1, 2], [-2, 0, 2], [-3, 0, 3], [0], [0]]
_DTYPES = {"fp32": _hip.F32, "f32": _hip.F32, "bf16": _hip.BF16, "bf16x3": _hip.BF16X3}


def get_time_context(x: torch.Tensor, c: Sequence[int] = (0,)) -> List[torch.Tensor]:
    """Time-shifted views of x[B,T,C], one per context offset (reference
    tdnn_layer.py:43-60).  Pure view arithmetic -- kept for API compatibility; the HIP
    kernels never materialise the concatenation, they index rows p + offset directly."""
    last = len(c) - 1
    lo, hi = c[0], c[last]
    return [x[:, hi + cc: (lo + cc if cc != hi else None), :] for cc in c]


def _stream_ptr(device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def _require_gpu(x: torch.Tensor, what: str):
    if not x.is_cuda:
        raise RuntimeError(f"{what}: expected a tensor on a HIP device, got {x.device}; "
                           "this package has no CPU path (the CPU oracle lives in oracle/ for tests only)")


class _Engine:
    """One libxvec_hip handle per (model, device) plus its cached workspace."""

    def __init__(self, cfg: dict, device: torch.device):
        self.device = device
        self.cfg = _hip.Cfg(cfg["input_size"], cfg["hidden_size"], cfg["num_classes"], cfg["x_vector_size"],
                            int(cfg["batch_norm"]), device.index if device.index is not None else
                            torch.cuda.current_device())
        h = C.c_void_p()
        _hip.check(_hip.lib.xvec_create(C.byref(self.cfg), C.byref(h)))
        self.h = h
        self.workspace: Optional[torch.Tensor] = None
        self.signature = None

    def workspace_bytes(self, total_frames: int, n_utts: int) -> int:
        need = _hip.lib.xvec_workspace_bytes(self.h, total_frames, n_utts)
        if need == 0:
            raise _hip.XvecError(_hip.ERR_ARG, "xvec_workspace_bytes returned 0")
        return int(need)

    def __del__(self):
        h, self.h = getattr(self, "h", None), None
        if h:
            try:
                _hip.lib.xvec_destroy(h)
            except Exception:
                pass

    def ensure_workspace(self, total_frames: int, n_utts: int):
        need = self.workspace_bytes(total_frames, n_utts)
        if self.workspace is None or self.workspace.numel() < need:
            self.workspace = None          # release before growing
            self.workspace = torch.empty(int(need * 1.25) if total_frames > 4096 else need, dtype=torch.uint8,
                                         device=self.device)
        return self.workspace.data_ptr(), self.workspace.numel()


class TdnnLayer(nn.Module):
    """Parameter container + single-layer entry point with the reference's constructor
    (tdnn_layer.py:6-24).  forward(x[B,T,in]) -> [B, T-(c[-1]-c[0]), out] runs the fused
    HIP layer (context gather + Linear + ReLU + eval BatchNorm in one kernel)."""

    def __init__(self, input_size=24, output_size=512, context=[0], batch_norm=True, dropout_p=0.0):
        super().__init__()
        self.input_size = input_size
        self.output_size = output_size
        self.context = context
        self.batch_norm = batch_norm
        self.dropout_p = dropout_p
        self.linear = nn.Linear(input_size * len(context), output_size)
        self.relu = nn.ReLU()
        if self.batch_norm:
            self.norm = nn.BatchNorm1d(output_size)
        if self.dropout_p:
            self.drop = nn.Dropout(p=self.dropout_p)
        self._owner = None      # (weakref to XVectorModel, layer index), set by the parent
        self._index = None

    def forward(self, x):
        owner = self._owner() if self._owner is not None else None
        if owner is None:
            raise RuntimeError("TdnnLayer must belong to an XVectorModel (its HIP engine owns the packed weights)")
        return owner._tdnn_layer(self._index, x)


class XVectorModel(nn.Module):
    """reference main.XVectorModel (main.py:23-94) on MI355X.

    Constructor arguments are the reference's; `batch_size`, `learning_rate`,
    `augmentations_per_sample` and `data_folder_path` belong to its training harness and
    are accepted and stored only.  Extensions: every entry point takes an optional
    `lengths` (valid frames per utterance of a zero-padded batch, BASELINE config 3) and
    `precision` selects the frame-level arithmetic: "fp32" (default; exact fp32 MFMA, the reference's
    arithmetic), "bf16" (parity bar 1e-2) or "bf16x3" (fp32 values carried as two bf16 planes, three
    bf16 products per k-step: fp32-level results, parity bar 1e-4, at bf16 matrix rates)."""

    def __init__(self, input_size=24, hidden_size=512, num_classes=1211, x_vector_size=512,
                 x_vec_extract_layer=6, batch_size=512, learning_rate=0.001, batch_norm=True,
                 dropout_p=0.0, augmentations_per_sample=2, data_folder_path='data', precision="fp32"):
        super().__init__()
        kw = dict(batch_norm=batch_norm, dropout_p=dropout_p)
        self.time_context_layers = nn.Sequential(
            TdnnLayer(input_size=input_size, output_size=hidden_size, context=_CONTEXTS[0], **kw),
            TdnnLayer(input_size=hidden_size, output_size=hidden_size, context=_CONTEXTS[1], **kw),
            TdnnLayer(input_size=hidden_size, output_size=hidden_size, context=_CONTEXTS[2], **kw),
            TdnnLayer(input_size=hidden_size, output_size=hidden_size, **kw),
            TdnnLayer(input_size=hidden_size, output_size=POOL_CHANNELS, **kw),
        )
        self.segment_layer6 = nn.Linear(2 * POOL_CHANNELS, x_vector_size)
        self.segment_layer7 = nn.Linear(x_vector_size, x_vector_size)
        self.output = nn.Linear(x_vector_size, num_classes)

        self.x_vec_extract_layer = x_vec_extract_layer
        self.batch_size = batch_size
        self.learning_rate = learning_rate
        self.augmentations_per_sample = augmentations_per_sample
        self.data_folder_path = data_folder_path
        self.precision = precision
        self.hparams = dict(input_size=input_size, hidden_size=hidden_size, num_classes=num_classes,
                            x_vector_size=x_vector_size, batch_norm=bool(batch_norm), dropout_p=dropout_p)
        import weakref
        ref = weakref.ref(self)
        for i, layer in enumerate(self.time_context_layers):
            layer._owner, layer._index = ref, i
        self._engines = {}
        self.eval()   # extraction-only build: BatchNorm always uses running statistics

    # ------------------------------------------------------------------ checkpoint ingestion (N2)
    _CTOR_KEYS = ("input_size", "hidden_size", "num_classes", "x_vector_size", "x_vec_extract_layer",
                  "batch_size", "learning_rate", "batch_norm", "dropout_p", "augmentations_per_sample",
                  "data_folder_path")

    @classmethod
    def load_from_checkpoint(cls, checkpoint_path, map_location="cpu", **overrides):
        """Counterpart of Lightning's `XVectorModel.load_from_checkpoint(path)` (reference
        main.py:213) without Lightning installed: reads `ckpt['hyper_parameters']` (what
        `save_hyperparameters()` stored, main.py:56) and `ckpt['state_dict']` from a reference
        `.ckpt` file.  Classes the pickle refers to but that are not importable here (Lightning's
        AttributeDict, callbacks) are read as plain dicts.

        SECURITY: a .ckpt is a pickle and is loaded with weights_only=False, exactly as Lightning's own loader does --
        un-pickling runs code named by the file.  Only load checkpoints you trust."""
        import pickle

        class _Lenient(pickle.Unpickler):
            def find_class(self, module, name):
                try:
                    return super().find_class(module, name)
                except (ImportError, AttributeError):
                    return type(name, (dict,), {"__setstate__": lambda self, st: self.update(st or {})})

        class _PickleModule:
            Unpickler = _Lenient
            load = staticmethod(lambda f, **kw: _Lenient(f, **kw).load())
            __name__ = "lenient_pickle"

        ckpt = torch.load(checkpoint_path, map_location=map_location, weights_only=False,
                          pickle_module=_PickleModule)
        hp = dict(ckpt.get("hyper_parameters", {}))
        hp.update(overrides)
        model = cls(**{k: hp[k] for k in cls._CTOR_KEYS if k in hp})
        sd = {k: v for k, v in ckpt["state_dict"].items()
              if k.startswith(("time_context_layers.", "segment_layer6.", "segment_layer7.", "output."))}
        model.load_state_dict(sd)
        return model

    # ------------------------------------------------------------------ engine plumbing
    def _check_mode(self):
        if self.training:
            raise RuntimeError(
                "XVectorModel is in training mode: batch-statistics BatchNorm and autograd are outside this "
                "build's scope (extraction path only). Call model.eval(); train with the reference and load "
                "its checkpoint here.")

    def _engine(self, device: torch.device) -> _Engine:
        if device.index is None:
            device = torch.device(device.type, torch.cuda.current_device())
        key = (device.type, device.index)
        with torch.cuda.device(device):     # handle creation and the packing kernels run on `device`
            eng = self._engines.get(key)
            if eng is None:
                eng = self._engines[key] = _Engine(self.hparams, device)
            self._sync_weights(eng)
        return eng

    def _hot_tensors(self):
        for layer in self.time_context_layers:
            yield layer.linear.weight
            yield layer.linear.bias
            if layer.batch_norm:
                yield layer.norm.weight
                yield layer.norm.bias
                yield layer.norm.running_mean
                yield layer.norm.running_var
        for lin in (self.segment_layer6, self.segment_layer7, self.output):
            yield lin.weight
            yield lin.bias

    def _sync_weights(self, eng: _Engine):
        """(Re)pack parameters into the engine when any of them changed (load_state_dict,
        .to(), in-place edits bump tensor._version)."""
        sig = tuple((t.data_ptr(), t._version) for t in self._hot_tensors())
        if sig == eng.signature:
            return
        dev = eng.device
        s = _stream_ptr(dev)

        def ptr(t):
            if t.device != dev or t.dtype != torch.float32:
                raise RuntimeError(f"parameter on {t.device}/{t.dtype}; move the model to {dev} as float32 first")
            return t.detach().contiguous().data_ptr()

        keep = []   # contiguous() temporaries must outlive the enqueued pack kernels
        for i, layer in enumerate(self.time_context_layers):
            ts = [layer.linear.weight, layer.linear.bias]
            if layer.batch_norm:
                ts += [layer.norm.weight, layer.norm.bias, layer.norm.running_mean, layer.norm.running_var]
                eps = layer.norm.eps
            else:
                eps = 0.0
            cs = [t.detach().contiguous() for t in ts]
            keep += cs
            for t in cs:
                if t.device != dev or t.dtype != torch.float32:
                    raise RuntimeError(f"parameter on {t.device}/{t.dtype}; move the model to {dev} as float32")
            p = [t.data_ptr() for t in cs] + [None] * (6 - len(cs))
            _hip.check(_hip.lib.xvec_load_tdnn(eng.h, i, p[0], p[1], p[2], p[3], p[4], p[5], eps, s))
        for which, lin in ((_hip.SEG6, self.segment_layer6), (_hip.SEG7, self.segment_layer7),
                           (_hip.OUTPUT, self.output)):
            _hip.check(_hip.lib.xvec_load_affine(eng.h, which, ptr(lin.weight), ptr(lin.bias), s))
        torch.cuda.current_stream(dev).synchronize()   # packing done before temporaries die
        del keep
        eng.signature = sig

    def _prep_input(self, x: torch.Tensor, what: str) -> torch.Tensor:
        _require_gpu(x, what)
        self._check_mode()
        if x.dim() != 3:
            raise ValueError(f"{what}: expected x[B,T,C], got shape {tuple(x.shape)}")
        if x.shape[0] < 1:
            raise ValueError(f"{what}: empty batch")
        return x.detach().float().contiguous()

    @staticmethod
    def _lengths_arg(lengths, B, T):
        if lengths is None:
            return None, None
        if torch.is_tensor(lengths):
            lengths = lengths.detach().cpu().tolist()
        lengths = [int(v) for v in lengths]
        if len(lengths) != B:
            raise ValueError(f"lengths has {len(lengths)} entries for a batch of {B}")
        arr = (C.c_int32 * B)(*lengths)
        return arr, lengths

    #: utterances per library call (xvec_hip.h: the pinned offsets ring holds 65 535 + 1 entries)
    MAX_UTTS_PER_CALL = 65535

    def _max_frames_per_call(self) -> Optional[int]:
        """Frames (B*T, padding included) one library call may carry, or None.  Only bf16x3 has such a limit: its two
        bf16 planes are addressed with 30-bit offsets (xvec_api.hip: plane = rows_alloc * hidden_pad * 2 bytes)."""
        if self.precision != "bf16x3":
            return None
        hidden_pad = -(-max(self.hparams["hidden_size"], 1) // 128) * 128
        return (0x3FFFFFFF // (hidden_pad * 2)) - 1024            # rows_alloc = frames rounded up to 128, + 264

    @staticmethod
    def _call_ranges(B: int, T: int, max_frames: Optional[int], max_utts: int, pool_pad: int = 1536, partial_limit: bool = True):
        """[lo, hi) utterance ranges of a [B, T, .] batch such that no range exceeds `max_frames` padded frames or
        `max_utts` utterances (a service handing over more than one call can hold gets several calls, not an error).
        `partial_limit`: the pooling partials of the 128x128 kernels (one slot per 32 frames and utterance, 3 planes of
        `pool_pad` = layer 5's padded width floats) are addressed with 32 bits and must stay below 2 GiB per call; the
        large-batch kernel's partials have no such limit (whatever this estimate misses, the library answers with
        XVEC_ERR_TOO_LARGE and _run halves the batch)."""
        per = max_utts if max_frames is None else min(max_utts, max_frames // T)
        if per < 1:
            raise ValueError(f"one utterance of {T} frames exceeds the {max_frames} frames a call can hold in this precision")
        if partial_limit:
            # (a single utterance beyond that -- 3.7 M frames, ten hours -- is refused by the library)
            per = max(1, min(per, int((0x7FFFFFFF // (3 * pool_pad * 4) - 2) / (T / 32.0 + 1.0))))
        return [(lo, min(lo + per, B)) for lo in range(0, B, per)]

    def _run(self, x: torch.Tensor, mode: int, lengths=None, workspace: Optional[torch.Tensor] = None) -> torch.Tensor:
        x = self._prep_input(x, "XVectorModel")
        B, T, Cin = x.shape
        if workspace is None:
            # layer 5 of a bf16 / bf16x3 batch this large goes to the large-batch kernel, whose partials have no 2 GiB limit
            # (xvec_api.hip, pp_blocks_per_col: 256-channel columns and >= 1.8 units of 64 frames per CU; 1536 = 6 columns)
            big = self.precision != "fp32" and B * (T - TOTAL_CONTEXT) >= 64 * 1024
            ranges = self._call_ranges(B, T, self._max_frames_per_call(), self.MAX_UTTS_PER_CALL,
                                       pool_pad=-(-POOL_CHANNELS // 128) * 128, partial_limit=not big)
            if len(ranges) > 1:
                if lengths is not None and torch.is_tensor(lengths):
                    lengths = lengths.detach().cpu().tolist()
                return torch.cat([self._run(x[lo:hi], mode, None if lengths is None else list(lengths[lo:hi]))
                                  for lo, hi in ranges], 0)
        if Cin != self.hparams["input_size"]:
            raise ValueError(f"expected {self.hparams['input_size']} input channels, got {Cin}")
        if T <= TOTAL_CONTEXT:
            raise ValueError(f"T={T}: the TDNN stack consumes {TOTAL_CONTEXT} frames of context; need T >= 15 "
                             "(the reference silently returns empty/NaN tensors here)")
        arr, lens = self._lengths_arg(lengths, B, T)
        if lens is not None and (min(lens) <= TOTAL_CONTEXT or max(lens) > T):
            raise ValueError(f"lengths must lie in [15, T={T}]")
        eng = self._engine(x.device)
        total = sum(lens) if lens is not None else B * T
        if workspace is not None:           # caller-owned scratch (GraphedPath: its pointers are baked into a graph)
            ws, ws_bytes = workspace.data_ptr(), workspace.numel()
        else:
            ws, ws_bytes = eng.ensure_workspace(total, B)
        n_out = (self.hparams["num_classes"] if mode == _hip.MODE_LOGITS else
                 2 * POOL_CHANNELS if mode == _hip.MODE_POOLED else self.hparams["x_vector_size"])
        out = torch.empty((B, n_out), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            rc = _hip.lib.xvec_forward(eng.h, x.data_ptr(), arr, B, T, mode, _DTYPES[self.precision],
                                       out.data_ptr(), ws, ws_bytes, _stream_ptr(x.device))
        if rc == _hip.ERR_TOO_LARGE and B > 1 and workspace is None:
            # one of the library's per-call size limits (32-bit row offsets, pooling partials < 2 GiB, 30-bit bf16x3
            # planes: include/xvec_hip.h, XVEC_ERR_TOO_LARGE) that _call_ranges did not foresee: two calls of half the batch each
            half = B // 2
            return torch.cat([self._run(x[:half], mode, None if lens is None else lens[:half]),
                              self._run(x[half:], mode, None if lens is None else lens[half:])], 0)
        _hip.check(rc)
        return out

    # ------------------------------------------------------------------ reference surface
    def stat_pool(self, x, lengths=None):
        """main.py:59-63: mean ‖ unbiased std over time, [B,T,C] -> [B,2C]."""
        _require_gpu(x, "stat_pool")
        if x.dim() != 3:
            raise ValueError(f"stat_pool: expected x[B,T,C], got {tuple(x.shape)}")
        x = x.detach().float().contiguous()
        B, T, Cc = x.shape
        out = torch.empty((B, 2 * Cc), dtype=torch.float32, device=x.device)
        lp = None
        if lengths is not None:
            lt = torch.as_tensor(lengths, dtype=torch.int32, device=x.device).contiguous()
            if lt.numel() != B:
                raise ValueError("stat_pool: one length per utterance expected")
            lp = lt.data_ptr()
        with torch.cuda.device(x.device):
            _hip.check(_hip.lib.xvec_stat_pool(x.data_ptr(), lp, B, T, Cc, out.data_ptr(), _stream_ptr(x.device)))
        return out

    def forward(self, x, lengths=None):
        """main.py:66-75: logits [B, num_classes]."""
        return self._run(x, _hip.MODE_LOGITS, lengths)

    def extract_x_vec(self, x, lengths=None):
        """main.py:81-94: x-vectors [B, x_vector_size] taken before the ReLU of
        segment_layer6 (x_vec_extract_layer == 6 or any other value) or segment_layer7 (== 7)."""
        mode = _hip.MODE_XVEC7 if self.x_vec_extract_layer == 7 else _hip.MODE_XVEC6
        return self._run(x, mode, lengths)

    def pooled(self, x, lengths=None):
        """`self.stat_pool(self.time_context_layers(x))` as forward / extract_x_vec compute it (main.py:68-69,
        83-84): [B, 3000] = mean ‖ unbiased std of the layer-5 output, which never exists in memory here (the
        pooling sums are formed in layer 5's epilogue)."""
        return self._run(x, _hip.MODE_POOLED, lengths)

    def graphed(self, example: torch.Tensor, logits: bool = False) -> "GraphedPath":
        """The whole path for one fixed shape [B, T, C] as ONE hipGraph (fixed-length batches only).

        Small or bf16 batches are launch-bound (7-8 launches of 10-100 us); the library neither
        synchronises nor allocates, so the launch sequence can be captured once and replayed:
        `g = model.graphed(x0); y = g(x)` copies x into the graph's input buffer and replays.
        The result tensor is owned by the graph and overwritten by the next call."""
        return GraphedPath(self, example, logits)

    def extract_packed(self, x_packed: torch.Tensor, offsets: Sequence[int], logits: bool = False):
        """Ragged batch without padding: x_packed[sum(len), C], utterance i in rows
        [offsets[i], offsets[i+1])."""
        _require_gpu(x_packed, "extract_packed")
        self._check_mode()
        x = x_packed.detach().float().contiguous()
        offs = [int(v) for v in offsets]
        B = len(offs) - 1
        if B < 1 or offs[0] != 0 or offs[-1] != x.shape[0] or x.dim() != 2:
            raise ValueError("extract_packed: need x[sum(len),C] and offsets[0]=0, offsets[-1]=rows")
        eng = self._engine(x.device)
        ws, ws_bytes = eng.ensure_workspace(offs[-1], B)
        mode = _hip.MODE_LOGITS if logits else (_hip.MODE_XVEC7 if self.x_vec_extract_layer == 7 else _hip.MODE_XVEC6)
        n_out = self.hparams["num_classes"] if logits else self.hparams["x_vector_size"]
        out = torch.empty((B, n_out), dtype=torch.float32, device=x.device)
        arr = (C.c_int64 * (B + 1))(*offs)
        with torch.cuda.device(x.device):
            _hip.check(_hip.lib.xvec_forward_packed(eng.h, x.data_ptr(), arr, B, mode, _DTYPES[self.precision],
                                                    out.data_ptr(), ws, ws_bytes, _stream_ptr(x.device)))
        return out

    # ------------------------------------------------------------------ per-stage entry points
    def _tdnn_layer(self, index: int, x: torch.Tensor) -> torch.Tensor:
        x = self._prep_input(x, "TdnnLayer")
        B, T, _ = x.shape
        layer = self.time_context_layers[index]
        span = layer.context[-1] - layer.context[0]
        if x.shape[2] != layer.input_size:
            raise ValueError(f"TdnnLayer {index}: expected {layer.input_size} channels, got {x.shape[2]}")
        if T <= span:
            raise ValueError(f"TdnnLayer {index}: T={T} not longer than the context span {span}")
        eng = self._engine(x.device)
        ws, ws_bytes = eng.ensure_workspace(B * T, B)
        y = torch.empty((B, T - span, layer.output_size), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            _hip.check(_hip.lib.xvec_tdnn_layer(eng.h, index, x.data_ptr(), B, T, _DTYPES[self.precision],
                                                y.data_ptr(), ws, ws_bytes, _stream_ptr(x.device)))
        return y

    def pooled_last_layer(self, x: torch.Tensor) -> torch.Tensor:
        """`self.stat_pool(self.time_context_layers[4](x))` on a given x[B,T,hidden]: the fused layer-5 + pooling
        kernel alone (per-stage entry for the parity tests; main.py:44,59-63)."""
        x = self._prep_input(x, "pooled_last_layer")
        B, T, Cin = x.shape
        if Cin != self.hparams["hidden_size"]:
            raise ValueError(f"pooled_last_layer: expected {self.hparams['hidden_size']} channels, got {Cin}")
        eng = self._engine(x.device)
        ws, ws_bytes = eng.ensure_workspace(B * T, B)
        out = torch.empty((B, 2 * POOL_CHANNELS), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            _hip.check(_hip.lib.xvec_tdnn_pool_layer(eng.h, x.data_ptr(), B, T, _DTYPES[self.precision], out.data_ptr(),
                                                     ws, ws_bytes, _stream_ptr(x.device)))
        return out

    def affine(self, which: str, x: torch.Tensor, relu: bool = False) -> torch.Tensor:
        """segment_layer6 / segment_layer7 / output as a stand-alone HIP GEMM (+ReLU)."""
        ids = {"segment_layer6": _hip.SEG6, "segment_layer7": _hip.SEG7, "output": _hip.OUTPUT}
        _require_gpu(x, "affine")
        self._check_mode()
        x = x.detach().float().contiguous()
        lin = getattr(self, which)
        if x.dim() != 2 or x.shape[1] != lin.in_features:
            raise ValueError(f"{which}: expected [M,{lin.in_features}], got {tuple(x.shape)}")
        eng = self._engine(x.device)
        y = torch.empty((x.shape[0], lin.out_features), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            _hip.check(_hip.lib.xvec_affine(eng.h, ids[which], x.data_ptr(), x.shape[0], int(relu), y.data_ptr(),
                                            _stream_ptr(x.device)))
        return y

    # ------------------------------------------------------------------ measurement
    def set_profiling(self, on: bool, device=None):
        dev = torch.device(device) if device is not None else next(self.parameters()).device
        eng = self._engine(dev)
        _hip.check(_hip.lib.xvec_set_profiling(eng.h, int(on)))

    def timings_ms(self, device=None) -> dict:
        dev = torch.device(device) if device is not None else next(self.parameters()).device
        eng = self._engine(dev)
        buf = (C.c_float * 16)()
        n = C.c_int(0)
        _hip.check(_hip.lib.xvec_get_timings(eng.h, buf, C.byref(n)))
        return {name: float(buf[i]) for i, name in enumerate(_hip.TIMING_NAMES[:n.value])}

    def last_dispatch(self, device=None) -> list:
        """Kernel family the last launch of each frame-level layer went to: "tile128" | "pp" | "first" | None."""
        dev = torch.device(device) if device is not None else next(self.parameters()).device
        eng = self._engine(dev)
        buf = (C.c_int * 8)()
        n = C.c_int(0)
        _hip.check(_hip.lib.xvec_get_dispatch(eng.h, buf, C.byref(n)))
        return [_hip.KERNEL_NAMES.get(int(buf[i])) for i in range(n.value)]

    # ------------------------------------------------------------------ caller shims (main.py:135-146)
    def test_step(self, batch, batch_index=0):
        """Same I/O as the reference's Lightning hook: casts to fp32, returns
        [(x_vecs, labels, ids)]."""
        samples, labels, ids = batch
        return [(self.extract_x_vec(samples.float()), labels, ids)]


class GraphedPath:
    """Captured launch sequence of XVectorModel.extract_x_vec / forward for one input shape."""

    def __init__(self, model: XVectorModel, example: torch.Tensor, logits: bool = False):
        _require_gpu(example, "graphed")
        self._x = example.detach().float().contiguous().clone()
        mode = _hip.MODE_LOGITS if logits else (_hip.MODE_XVEC7 if model.x_vec_extract_layer == 7 else _hip.MODE_XVEC6)
        eng = model._engine(self._x.device)            # weights packed outside the capture
        B, T, _ = self._x.shape
        # The graph bakes raw pointers into the scratch buffer, so the graph OWNS its scratch: the
        # engine's shared workspace is reallocated whenever a later, larger batch needs more room.
        self._ws = torch.empty(eng.workspace_bytes(B * T, B), dtype=torch.uint8, device=self._x.device)
        model._run(self._x, mode, workspace=self._ws)
        torch.cuda.synchronize(self._x.device)
        self._graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self._graph):
            self._y = model._run(self._x, mode, workspace=self._ws)
        self._model = model                            # keeps the engine (packed weights) alive

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        if x.shape != self._x.shape:
            raise ValueError(f"graph captured for {tuple(self._x.shape)}, got {tuple(x.shape)}")
        self._x.copy_(x, non_blocking=True)
        self._graph.replay()
        return self._y

