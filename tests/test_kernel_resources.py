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
SOURCES = ["tdnn_layer.hip", "tdnn_pp16.hip", "tdnn_first.hip", "affine.hip", "pool.hip", "mfcc.hip"]


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
                         ("spill", r"VGPRs Spill: (\d+)")):
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
            assert r.get("scratch", 0) == 0 and r.get("spill", 0) == 0, f"{src}: {name} uses scratch: {r}"
            assert r.get("vgprs", 0) <= 256, f"{src}: {name}: {r}"
    assert len(results["tdnn_layer.hip"]) == 14 and len(results["tdnn_pp16.hip"]) == 4 and len(results["tdnn_first.hip"]) == 4
    assert len(results["mfcc.hip"]) == 3
    assert total >= 31
