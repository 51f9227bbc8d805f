"""Exact check of the matrix-pipe pooling sums of `pp16::tdnn_pp_kernel<true, false>` (csrc/tdnn_pp16.hip, SegMx), ADVICE r04:
the kernel-vs-kernel and oracle checks of the pooled statistics sit at 1e-3 / 5e-3 since the deviations are rounded to bf16, and
one dropped or doubly counted frame moves a mean by 3.5e-3 -- under that bar.  Here the rounding is taken OUT of the comparison:

  * the full plain-bf16 path runs to the pooled statistics (XVEC_MODE_POOLED), then the test reads, from the workspace
    (xvec_workspace_layout), layer 4's bf16 output -- exactly the rows layer 5 read -- and layer 5's partials: per (block of the
    column, utterance, frames half) the pivot C, S1 = sum d, S2 = sum d^2 and the frame count;
  * the host forms z = x . W^T + bias in fp64 from those same bf16 rows and bf16 weights (every product exact), then
    d = max(bf16(z + bias - C), -C) with the C the kernel stored, and sums d and d^2 over exactly the frames the kernel's tiling
    gives that (block, utterance, half) -- the block ranges and tile cuts are restated from the kernel;
  * counts must match exactly, the pivots must be bf16(relu) of the block's first frame, and S1 / S2 must agree to two and a half bf16 steps of the segment's
    LARGEST deviation: what is left is fp32 accumulation order flipping the bf16 rounding of a few deviations (2^-8 |d| each, one
    in a few thousand); a frame that is dropped, masked by mistake or counted twice moves S1 by its whole |d| -- a typical
    frame is 25 times the bound.

Layer 4's BatchNorm is the identity and layer 5's weights are bf16-exact in this model, so that the deferred-BatchNorm fold
(xvec_api.hip, refold) leaves W' = W and bias' = bias and the host needs no copy of the fold's arithmetic."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _model(sd42):
    import xvector_amd as xa
    sd = {k: v.clone() for k, v in sd42.items()}
    p = "time_context_layers.3.norm."
    sd[p + "weight"].fill_(1.0)
    sd[p + "bias"].zero_()
    sd[p + "running_mean"].zero_()
    sd[p + "running_var"].fill_(1.0 - 1e-5)
    w5 = "time_context_layers.4.linear.weight"
    sd[w5] = sd[w5].bfloat16().float()
    m = xa.XVectorModel(precision="bf16")
    m.load_state_dict(sd)
    return m.to(DEV).eval(), sd


def _tiles(n_units):
    """tile heights (units of 64 frames) of a block that owns n_units (tdnn_pp16.hip, tdnn_pp_kernel)"""
    nt = (n_units + 3) // 4
    base, extra = n_units // nt, n_units % nt
    if base < 2:
        base, extra, nt = 2, 0, (n_units + 1) // 2
    return [base + 1 if i < extra else base for i in range(nt)]


def _check(m, sd, x, lengths):
    from xvector_amd import hip
    dev = torch.device(DEV)
    B, T, _ = x.shape
    pooled = m.pooled(x, lengths=lengths)
    torch.cuda.synchronize()
    assert torch.isfinite(pooled).all()
    assert m.last_dispatch(dev)[4] == "pp", m.last_dispatch(dev)
    lens = [T] * B if lengths is None else [int(v) for v in lengths]
    total = sum(lens)
    eng = m._engine(dev)
    lay = hip.WsLayout()
    hip.check(hip.lib.xvec_workspace_layout(eng.h, total, B, C.byref(lay)))
    ws = eng.workspace
    n_pad, nh = lay.pool_n_pad, lay.hidden_n_pad
    off = np.concatenate([[0], np.cumsum([l - 14 for l in lens])]).astype(np.int64)       # compact rows of layers 4 / 5
    rows = int(off[-1])
    x4 = ws[lay.act_b: lay.act_b + rows * nh * 2].view(torch.bfloat16).reshape(rows, nh)[:, :512].double().cpu()
    part = ws[lay.part: lay.part + lay.part_slots * 3 * n_pad * 4].view(torch.float32).reshape(-1, 3, n_pad).double().cpu()
    cnt = ws[lay.part_cnt: lay.part_cnt + 2 * (lay.num_cu + B + 2) * 4].view(torch.int32).cpu()
    W = sd["time_context_layers.4.linear.weight"].double()
    bias = sd["time_context_layers.4.linear.bias"].double()
    zb = x4 @ W.t() + bias                                   # [rows, 1500] = z + bias, every product exact
    # what the kernel's fp32 accumulation of a 512-term sum may be off by, per value (a few ulp of the sum of magnitudes):
    # matters where z + bias cancels to nearly nothing
    ez = 4.0 * 2.0 ** -24 * (x4.abs() @ W.abs().t() + bias.abs())

    units = (rows + 63) // 64
    bpc = lay.num_cu // (n_pad // 256)
    utt_of = np.searchsorted(off, np.arange(rows), side="right") - 1
    n_seg = exact = n_ch = 0
    sensed = False
    worst1 = worst2 = 0.0
    for p in range(bpc):
        u_begin, u_end = units * p // bpc, units * (p + 1) // bpc
        if u_end <= u_begin:
            continue
        limit = min(u_end * 64, rows)
        heights = _tiles(u_end - u_begin)
        for g in range(2):
            rws, m0 = [], u_begin * 64
            for mr in heights:
                lo = m0 + g * 32 * mr
                rws.append(np.arange(lo, min(lo + 32 * mr, limit)) if lo < limit else np.arange(0))
                m0 += 64 * mr
            first_row = u_begin * 64 + g * 32 * heights[0]
            rws = np.concatenate(rws)
            utts = range(int(utt_of[u_begin * 64]), int(utt_of[limit - 1]) + 1)       # every utterance overlapping the block's rows
            for u in utts:
                slot = 2 * (p + u) + g
                sel = rws[utt_of[rws] == u] if len(rws) else rws
                assert int(cnt[slot]) == len(sel), f"block {p} half {g} utterance {u}: count {int(cnt[slot])}, tiling says {len(sel)}"
                Cp = part[slot, 0, :1500]
                if first_row < limit:        # the pivot: bf16(relu) of the half's first frame in the block (0 if it has none)
                    want = torch.clamp(zb[first_row], min=0).float().bfloat16().double()
                    okp = (Cp == want) | ((Cp - want).abs() <= 2.0 ** -7 * want.abs() + 1e-6)      # (one bf16 step: fp32 sum order)
                    assert okp.all(), f"block {p} half {g}: {int((~okp).sum())} pivots are not bf16(relu(first frame))"
                else:
                    assert (Cp == 0).all()
                if len(sel) == 0:
                    assert (part[slot, 1:, :1500] == 0).all(), f"block {p} half {g} utterance {u}: sums of an empty partial"
                    continue
                acc = (zb[sel] - Cp).float().bfloat16().double()
                d = torch.maximum(acc, -Cp)
                s1, s2, a1 = d.sum(0), (d * d).sum(0), d.abs().sum(0)
                # what may differ: the kernel's fp32 sums round a deviation to the other bf16 neighbour now and then (one in a
                # few thousand; 2^-8 |d| each) -- allow two and a half such steps of the LARGEST deviation, plus the fp32
                # accumulation of the sums themselves
                dmax = d.abs().max(0).values
                ezs = ez[sel].sum(0)
                t1 = 2.5 * 2.0 ** -8 * dmax + 2e-6 * a1 + ezs + 1e-12
                t2 = 2.5 * 2.0 ** -7 * dmax * dmax + 2e-6 * s2 + 2.0 * dmax * ezs + 1e-12
                r1 = ((part[slot, 1, :1500] - s1).abs() / t1).max().item()
                r2 = ((part[slot, 2, :1500] - s2).abs() / t2).max().item()
                worst1, worst2 = max(worst1, r1), max(worst2, r2)
                # (a dropped, masked or doubled frame moves S1 by its |d|: a typical frame is 20-30 % of the largest one,
                #  25 times this bound)
                if not (r1 <= 1.0 and r2 <= 1.0):
                    ch = int(((part[slot, 1, :1500] - s1).abs() / t1).argmax())
                    on = int((acc[:, ch] > -Cp[ch]).sum())
                    raise AssertionError(
                        f"block {p} half {g} utterance {u} ({len(sel)} frames): S1 off by {r1:.2f} x its bound, S2 by {r2:.2f} x "
                        f"(bound = 2.5 bf16 steps of the largest deviation); channel {ch}: S1 {float(part[slot, 1, ch])!r} vs {float(s1[ch])!r}, "
                        f"S2 {float(part[slot, 2, ch])!r} vs {float(s2[ch])!r}, C {float(Cp[ch])!r}, largest |d| {float(dmax[ch])!r}, "
                        f"sum |d| {float(a1[ch])!r}, {on} frames above -C; first rows {sel[:3].tolist()}; d[:6] {d[:6, ch].tolist()}")
                exact += int(((part[slot, 1, :1500] - s1).abs() <= 2e-6 * a1 + 1e-12).sum())
                if not sensed and len(sel) >= 32:
                    # the check's own power, once per run: the host sums WITHOUT one frame (what a kernel that dropped it would
                    # have left) must break the bound in most channels that are on in that frame
                    miss = (s1 - d[len(sel) // 2]).sub(part[slot, 1, :1500]).abs() / t1
                    live = d[len(sel) // 2] > -Cp
                    assert live.sum() > 100 and (miss[live] > 1.0).double().mean() > 0.9, "the bound would not notice a dropped frame"
                    sensed = True
                n_ch += 1500
                n_seg += 1
    print(f"[segmx] B={B} T={T} ragged={lengths is not None}: {n_seg} segment partials; worst S1 {worst1:.2f}, S2 {worst2:.2f} of the bound; "
          f"{exact / max(n_ch, 1):.4f} of the S1 sums agree to fp32 accumulation alone (no rounding flip)")
    assert n_seg >= bpc and sensed


# (160 x 300 and 256 x 300 -- BASELINE configs[4]'s own batch -- are the shapes whose blocks cut their ranges into four-unit tiles:
#  the host side holds two [rows, 1500] fp64 matrices, 1.8 GB at 256 x 300)
@pytest.mark.parametrize("B,T", [(20, 300), (52, 300), (63, 300), (100, 300), (128, 300), (37, 517), (160, 300), (256, 300)])
def test_segmx_sums_fixed(sd42, synth, B, T):
    m, sd = _model(sd42)
    _check(m, sd, torch.from_numpy(synth.make_mfcc(B, T, seed=B)).to(DEV), None)


def test_segmx_sums_ragged(sd42, synth):
    m, sd = _model(sd42)
    rng = np.random.default_rng(5)
    lens = rng.integers(15 + 14, 701, 48).tolist()
    lens[3], lens[17] = 29, 700
    _check(m, sd, torch.from_numpy(synth.make_mfcc(48, 700, seed=2)).to(DEV), lens)
