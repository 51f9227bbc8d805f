"""Segment anatomy of pp16::tdnn_pp_kernel (diagnostic build, s_memtime stamps of waves 0 and 4 of every block).
usage: XVEC_LIB=$PWD/build/diag/libxvec_hip_diag.so python profiles/diag/pp_stamps.py [layer ...]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import xvector_amd as xa
from xvector_amd import hip
dev = torch.device("cuda:0")
sd = {k: torch.from_numpy(np.asarray(v)) for k, v in xa.synth.make_state_dict(seed=42).items()}
m = xa.XVectorModel(precision="bf16"); m.load_state_dict(sd); m = m.to(dev).eval()
rd = hip.lib.xvec_pp16_diag_read
rd.restype = C.c_int; rd.argtypes = [C.c_void_p, C.c_int, C.c_int]
B, T = 256, 300
names = ["L0 work+lgkm", "L0 vmcnt", "L0 barrier", "C0 mfma+lgkm", "C0 barrier", "rows+head1", "L1 issue", "L1 vmcnt", "L1 barrier",
         "C1 mfma+lgkm", "C1 barrier", "epilogue+head2", "head wait", "tail barrier", "tiles", "sum MR"]
PER_TILE = (5, 11, 12, 13)
layers = [int(v) for v in sys.argv[1:]] or [1, 2, 3, 4]
Tin = {1: 296, 2: 292, 3: 286}
x_full = torch.from_numpy(xa.synth.make_mfcc(B, T, seed=0)).to(dev)
for layer in layers:
    def run():
        if layer == 4:      # the pooling variant only runs inside the whole path; its stamps have their own array
            m.extract_x_vec(x_full)
        else:
            m.time_context_layers[layer](torch.randn(B, Tin[layer], 512, device=dev))
    run()
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * (2 * 512 * 32))()
    assert rd(buf, 2 * 512 * 32, 1) == 0
    run()
    torch.cuda.synchronize()
    assert rd(buf, 2 * 512 * 32, 1) == 0
    a = np.frombuffer(buf, dtype=np.uint64).reshape(2, 512, 2, 16).astype(np.float64)[1 if layer == 4 else 0]
    a = a[a[:, 0, 14] > 0]
    print(f"layer {layer}: {a.shape[0]} blocks, tiles/block {a[:, 0, 14].mean():.2f}, mean MR {a[:, 0, 15].sum() / a[:, 0, 14].sum():.2f}")
    nk = {1: 24, 2: 24, 3: 8, 4: 8}[layer]
    for g in (0, 1):
        tot = a[:, g, :14].sum(axis=1).mean()
        print(f"  group {g}: total cycles/block {tot:.0f}")
        for k in range(14):
            if names[k] == "-":
                continue
            v = a[:, g, k].mean()
            unit = v / a[:, g, 14].mean() if k in PER_TILE else v / (a[:, g, 14].mean() * nk)
            print(f"    {names[k]:14s} {v:10.0f} cycles/block  {100 * v / tot:5.1f} %   {unit:8.1f} per {'tile' if k in PER_TILE else 'K-tile'}")
        if os.environ.get("STAMP_EPI"):      # library built with DIAG=2: kinds 0-5 are the pooling epilogue's pieces
            for k, nm in ((13, "K loop etc."), (0, "restore"), (1, "row 0"), (2, "row 1"), (3, "row 2"), (4, "row 3"), (5, "finish+park")):
                print(f"      epilogue {nm:12s} {a[:, g, k].mean() / a[:, g, 14].mean():8.1f} per tile")
    try:        # core clock during the launch: block 0's s_memtime (core cycles) over s_memrealtime (100 MHz)
        ck = (C.c_ulonglong * 8)()
        hip.lib.xvec_pp16_clk_read.argtypes = [C.c_void_p]
        if hip.lib.xvec_pp16_clk_read(ck) == 0:
            o = 4 if layer == 4 else 0
            dt_core, dt_real = ck[o + 2] - ck[o + 0], ck[o + 3] - ck[o + 1]
            print(f"  block 0: {dt_core} core cycles in {dt_real / 100:.1f} us -> {dt_core / (dt_real / 100) / 1e3:.2f} GHz")
    except AttributeError:
        pass
