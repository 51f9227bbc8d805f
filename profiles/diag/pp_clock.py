"""Core clock inside the bf16 path under sustained load (diagnostic build): block 0 of the last layer-4 launch and of
the layer-5 launch record s_memtime (core cycles) and s_memrealtime (100 MHz) at entry and exit.
usage: XVEC_LIB=$PWD/build/diag/libxvec_hip_diag.so python profiles/diag/pp_clock.py [iterations]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import xvector_amd as xa
from xvector_amd import hip
dev = torch.device("cuda:0")
sd = {k: torch.from_numpy(np.asarray(v)) for k, v in xa.synth.make_state_dict(seed=42).items()}
m = xa.XVectorModel(precision="bf16"); m.load_state_dict(sd); m = m.to(dev).eval()
x = torch.from_numpy(xa.synth.make_mfcc(256, 300, seed=0)).to(dev)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
hip.lib.xvec_pp_clk_read.argtypes = [C.c_void_p]
for it in (10, n):
    for _ in range(it):
        m.extract_x_vec(x)
    torch.cuda.synchronize()
    ck = (C.c_ulonglong * 8)()
    assert hip.lib.xvec_pp_clk_read(ck) == 0
    for name, o in (("layer 4 (store, K=512)", 0), ("layer 5 (pool)", 4)):
        dc, dr = ck[o + 2] - ck[o], ck[o + 3] - ck[o + 1]
        print(f"after {it:4d} steps: {name}: block 0 lived {dc} core cycles = {dr / 100:.1f} us -> {dc / (dr / 100) / 1e3:.2f} GHz")
