"""No shipped hot-path kernel may spill: scratch traffic goes through the vector-memory queue the LDS-DMA pipelines
count on (a `scratch_load` waits with vmcnt, i.e. for every DMA piece in flight), and round 1 shipped two variants
with 20 / 52 bytes per lane unnoticed.  hipcc reports per-kernel resources with -Rpass-analysis=kernel-resource-usage;
this test compiles the frame-level and segment-level sources for gfx950 (device code only) and checks every kernel."""
import concurrent.futures as cf
import os
import re
import shutil
import subprocess

import pytest

from conftest import ROOT

CSRC = os.path.join(ROOT, "speaker-recognition-x-vectors_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"
SOURCES = ["tdnn_layer.hip", "tdnn_pp16.hip", "tdnn_first.hip", "affine.hip", "pool.hip", "mfcc.hip", "score.hip"]
# bytes of scratch a kernel may use.  The 128 x 128 fp64 score GEMM parks a dozen tile-invariant 64-bit addresses in scratch
# OUTSIDE its K loop (stored once before the persistent loop, re-read in the epilogue and at a tile's first chunk; the loop
# itself has no scratch access: checked below on the assembly); everything else: none.
SCRATCH_OK = {"gemm_nt_f64_kernelILb1ELi4ELb0E": 128, "gemm_nt_f64_kernelILb0ELi4ELb0E": 128}


def _resources(src):
    cmd = [HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only", "-fno-slp-vectorize", "-Wno-unused-function",
           "-Wno-pass-failed", "-Wno-inline-asm", "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", os.devnull]
    out = subprocess.run(cmd, cwd=CSRC, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    kernels, name = {}, None
    for line in out.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = m.group(1)
            kernels[name] = {}
        for key, pat in (("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"), ("vgprs", r" VGPRs: (\d+)"),
                         ("spill", r"VGPRs Spill: (\d+)"), ("agprs", r" AGPRs: (\d+)"),
                         ("occupancy", r"Occupancy \[waves/SIMD\]: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)")):
            m = re.search(pat, line)
            if m and name:
                kernels[name][key] = int(m.group(1))
    return src, kernels


@pytest.mark.skipif(not os.path.exists(HIPCC) or shutil.which("make") is None, reason="needs hipcc")
def test_no_kernel_uses_scratch():
    with cf.ThreadPoolExecutor(max_workers=4) as pool:
        results = dict(pool.map(_resources, SOURCES))
    total = 0
    for src, kernels in results.items():
        assert kernels, f"{src}: no kernel reported"
        for name, r in kernels.items():
            total += 1
            allowed = max([v for k, v in SCRATCH_OK.items() if k in name] or [0])
            assert r.get("scratch", 0) <= allowed, f"{src}: {name} uses scratch: {r}"
            assert r.get("vgprs", 0) <= 256, f"{src}: {name}: {r}"
    assert len(results["tdnn_layer.hip"]) == 14 and len(results["tdnn_pp16.hip"]) == 4 and len(results["tdnn_first.hip"]) == 4
    assert len(results["mfcc.hip"]) == 6          # mfcc512_kernel<fp32 | 16-bit PCM samples, banded | dense filterbank>, mfcc_kernel<9 | 0>
    # the nfft-512 MFCC kernel runs FIVE blocks of four waves per CU (round 6: its SIMDs were half busy at four): that takes at
    # most 96 registers of the unified file (512 / 5, in eights) and 32 KiB of the CU's 160 KiB of LDS per block -- one register
    # or one row of padding more and the launch silently drops back to four
    for name, r in results["mfcc.hip"].items():
        if "mfcc512_kernel" in name:
            assert r["vgprs"] + r.get("agprs", 0) <= 96 and r["lds"] <= 32 * 1024 and r["occupancy"] == 5, f"{name}: {r}"
    assert len(results["score.hip"]) == 7         # gemm_nt_f64_kernel<VEC, WT, PRE>: 2 x (2 + 1) + normalize_rows_kernel
    assert total >= 41


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc")
def test_score_gemm_k_loop_is_free_of_scratch():
    """The one kernel with a scratch allowance (SCRATCH_OK): its spills must stay outside the K loop -- the innermost loop around
    the MFMAs (label ... backward branch) holds no scratch_ instruction."""
    cmd = [HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only", "-fno-slp-vectorize", "-Wno-unused-function",
           "-Wno-pass-failed", "-Wno-inline-asm", "-S", "score.hip", "-o", "-"]
    out = subprocess.run(cmd, cwd=CSRC, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    checked = 0
    for body in re.split(r"\n(?=_ZN\S*gemm_nt_f64_kernel\S*:)", out.stdout)[1:]:
        name = body.split(":", 1)[0]
        lines = body.split("s_endpgm")[0].splitlines()
        mf = [i for i, l in enumerate(lines) if "v_mfma_f64" in l]
        assert mf, name
        labels = {m.group(1): i for i, l in enumerate(lines) for m in [re.match(r"(\.LBB\d+_\d+):", l)] if m}
        # the first backward branch behind the last MFMA whose target lies in front of the first MFMA closes the K loop
        loop = None
        for i in range(mf[-1], len(lines)):
            m = re.search(r"s_cbranch_\w+ (\.LBB\d+_\d+)", lines[i])
            if m and labels.get(m.group(1), len(lines)) <= mf[0]:
                loop = (labels[m.group(1)], i)
                break
        assert loop, f"{name}: K loop not found"
        assert not [l for l in lines[loop[0]:loop[1]] if "scratch_" in l], f"{name}: scratch access inside the K loop"
        checked += 1
    assert checked == 6


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc")
def test_mfcc_sample_loads_are_waited_for_before_any_use():
    """fft512::mfcc512_kernel requests a pair of frames with 32 inline-asm loads and waits for them with an asm s_waitcnt that
    names their destinations (csrc/mfcc.hip, MF_REQUEST): hipcc's own wait-count pass sees neither.  On the assembly: between
    the first load of a burst and the s_waitcnt vmcnt(0) that ends it, no instruction reads or writes a register a load of the
    burst has been told to fill.  Both input types."""
    cmd = [HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only", "-fno-slp-vectorize", "-Wno-unused-function",
           "-Wno-pass-failed", "-Wno-inline-asm", "-S", "mfcc.hip", "-o", "-"]
    out = subprocess.run(cmd, cwd=CSRC, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    bursts = 0
    for body in re.split(r"\n(?=_ZN\S*mfcc512_kernel\S*:)", out.stdout)[1:]:
        name = body.split(":", 1)[0]
        pending = None                       # destination registers of the burst in flight
        in_asm = False                       # (loads hipcc emits itself -- table fetches -- are tracked by its own wait counts)
        for line in body.split("s_endpgm")[0].splitlines():
            if "#ASMSTART" in line or "#ASMEND" in line:
                in_asm = "#ASMSTART" in line
            ins = line.split(";")[0].strip()
            m = re.match(r"buffer_load_(?:dword|sshort) (v\d+), (v\d+),", ins) if in_asm else None
            if m:
                pending = set() if pending is None else pending
                assert m.group(2) not in pending, f"{name}: {ins!r} takes its address from a register a load in flight fills"
                pending.add(m.group(1))
                continue
            if pending is None:
                continue
            if re.match(r"s_waitcnt vmcnt\(0\)", ins):
                assert len(pending) == 32, f"{name}: a burst of {len(pending)} loads"
                pending, bursts = None, bursts + 1
                continue
            used = set(re.findall(r"\bv\d+\b", ins))
            for lo, hi in re.findall(r"v\[(\d+):(\d+)\]", ins):
                used |= {f"v{i}" for i in range(int(lo), int(hi) + 1)}
            assert not (used & pending), f"{name}: {ins!r} touches {sorted(used & pending)} before the loads have been waited for"
    assert bursts >= 4                       # one request site per kernel (fp32 samples | 16-bit PCM, banded | dense filterbank)
