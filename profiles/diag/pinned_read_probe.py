"""How fast does the CPU read page-locked memory right after the GPU wrote it?  (first-touch clone of a
[256,512] fp32 result buffer after a D2H copy; torch pin_memory vs hipHostRegister vs hipHostMalloc flags)"""
import ctypes as C, time, torch
dev = torch.device("cuda:0")
src = torch.randn(256, 512, device=dev)
rt = torch.cuda.cudart()
hip = C.CDLL("libamdhip64.so")
hip.hipHostMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
hip.hipHostMalloc.restype = C.c_int
def hostmalloc(nbytes, flags):
    p = C.c_void_p()
    rc = hip.hipHostMalloc(C.byref(p), nbytes, flags)
    assert rc == 0, rc
    buf = (C.c_float * (nbytes // 4)).from_address(p.value)
    return torch.frombuffer(buf, dtype=torch.float32).view(256, 512), buf
cands = {"torch pin_memory": torch.empty(256, 512).pin_memory()}
reg = torch.empty(256, 512)
assert int(rt.cudaHostRegister(reg.data_ptr(), reg.numel() * 4, 0)) == 0
cands["hipHostRegister"] = reg
keep = []
for name, flags in (("hipHostMalloc default(0)", 0x0), ("hipHostMalloc portable(1)", 0x1), ("hipHostMalloc coherent(0x40000000)", 0x40000000),
                    ("hipHostMalloc noncoherent(0x80000000)", 0x80000000), ("hipHostMalloc numa_user(0x20000000)", 0x20000000)):
    try:
        t, b = hostmalloc(256 * 512 * 4, flags); keep.append(b); cands[name] = t
    except AssertionError as e:
        print(name, "failed", e)
cands["pageable (blocking copy)"] = torch.empty(256, 512)
for name, buf in cands.items():
    ts = []
    for it in range(5):
        src.add_(1.0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        buf.copy_(src, non_blocking=True)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        c = buf.clone()
        t2 = time.perf_counter()
        assert torch.equal(c, src.cpu())
        ts.append(((t1 - t0) * 1e3, (t2 - t1) * 1e3))
    print(f"{name:40s} pinned={buf.is_pinned()}  copy+sync {min(t[0] for t in ts):.3f} ms   first clone {min(t[1] for t in ts):.3f} ms (max {max(t[1] for t in ts):.3f})", flush=True)
