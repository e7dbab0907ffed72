#!/usr/bin/env python3
"""svt_hip_malloc_spread (the C ABI's allocator, hipMalloc / hipFree directly) against three consecutive svt_hip_malloc calls:
the headline kernel on raw pointers."""
import ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as ge
pkg = ge.load_package(); dsp = pkg.SvtHipDsp(0); dev = torch.device("cuda:0"); L = dsp.lib
n = 1 << 20
qrow = {k: v[100].copy() for k, v in pkg.tables.quant_tables(8).items()}
iscan = torch.from_numpy(pkg.tables.scan_tables(pkg.TX_32X32, pkg.DCT_DCT)[1]).to(dev)
g = torch.Generator(device=dev); g.manual_seed(13596)
src = torch.randint(0, 256, (n, 32, 32), dtype=torch.uint8, device=dev, generator=g)
pred = torch.randint(0, 256, (n, 32, 32), dtype=torch.uint8, device=dev, generator=g)
eob = torch.zeros(n, dtype=torch.int16, device=dev); sad = torch.zeros(n, dtype=torch.int32, device=dev)
import numpy as np
tabs = [np.ascontiguousarray(qrow[k], dtype=np.int16) for k in ("zbin", "round", "quant", "quant_shift", "dequant")]
L.svt_hip_malloc_spread.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
def kernel(p):
    rc = L.svt_hip_fwd_quant_sad_batch(src.data_ptr(), pred.data_ptr(), n, pkg.TX_32X32, pkg.DCT_DCT, tabs[0].ctypes.data, tabs[1].ctypes.data,
                                       tabs[2].ctypes.data, tabs[3].ctypes.data, tabs[4].ctypes.data, iscan.data_ptr(), ctypes.c_void_p(p[0]),
                                       ctypes.c_void_p(p[1]), ctypes.c_void_p(p[2]), eob.data_ptr(), sad.data_ptr(), dsp._stream())
    assert rc == 0, L.svt_hip_last_error()
def timeit(fn, iters=8):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
B = n * 4096
for rnd in range(2):
    p = [int(L.svt_hip_malloc(B) or 0) for _ in range(3)]
    assert all(p)
    ms = timeit(lambda: kernel(p))
    print(json.dumps({"alloc": "three svt_hip_malloc calls", "Mblocks_per_s": round(n / ms / 1e3, 1), "ptr_GiB": [round(q / 2 ** 30, 2) for q in p]}), flush=True)
    for q in p: L.svt_hip_free(ctypes.c_void_p(q))
    nb = (ctypes.c_size_t * 3)(B, B, B); ptrs = (ctypes.c_void_p * 3)()
    assert L.svt_hip_malloc_spread(nb, 3, 0, ptrs) == 0
    p = [int(v) for v in ptrs]
    ms = timeit(lambda: kernel(p))
    print(json.dumps({"alloc": "svt_hip_malloc_spread (32 GiB spacers)", "Mblocks_per_s": round(n / ms / 1e3, 1), "ptr_GiB": [round(q / 2 ** 30, 2) for q in p]}), flush=True)
    for q in p: L.svt_hip_free(ctypes.c_void_p(q))
