#!/usr/bin/env python3
"""The three output arrays as separate allocations: back to back, or with 32 GiB spacers between them that are freed again."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as ge
pkg = ge.load_package(); dsp = pkg.SvtHipDsp(0); dev = torch.device("cuda:0")
n = 1 << 20
qrow = {k: v[100].copy() for k, v in pkg.tables.quant_tables(8).items()}
iscan = torch.from_numpy(pkg.tables.scan_tables(pkg.TX_32X32, pkg.DCT_DCT)[1]).to(dev)
g = torch.Generator(device=dev); g.manual_seed(13596)
src = torch.randint(0, 256, (n, 32, 32), dtype=torch.uint8, device=dev, generator=g)
pred = torch.randint(0, 256, (n, 32, 32), dtype=torch.uint8, device=dev, generator=g)
eob = torch.zeros(n, dtype=torch.int16, device=dev); sad = torch.zeros(n, dtype=torch.int32, device=dev)
def timeit(fn, iters=8):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
def run(arrs, tag):
    outs = tuple(arrs) + (eob, sad)
    ms = timeit(lambda: dsp.fwd_quant_sad(src, pred, pkg.TX_32X32, pkg.DCT_DCT, qrow, iscan, outs=outs))
    print(json.dumps({"case": tag, "Mblocks_per_s": round(n / ms / 1e3, 1), "ptr_GiB": [round(a.data_ptr() / 2 ** 30, 2) for a in arrs]}), flush=True)
mk = lambda: torch.empty((n, 1024), dtype=torch.int32, device=dev)
a = [mk(), mk(), mk()]
run(a, "three separate 4 GiB allocations, one after the other")
del a; torch.cuda.empty_cache()
for gap in (8, 16, 32, 64):
    arrs, spacers = [], []
    for k in range(3):
        arrs.append(mk())
        if k < 2: spacers.append(torch.empty(gap << 30, dtype=torch.uint8, device=dev))
    run(arrs, f"{gap} GiB spacers between them (still held)")
    del spacers; torch.cuda.empty_cache()
    run(arrs, f"{gap} GiB spacers freed again")
    del arrs; torch.cuda.empty_cache()
big = torch.empty(3 * n * 1024, dtype=torch.int32, device=dev)
run([big[:n * 1024].view(n, 1024), big[n * 1024:2 * n * 1024].view(n, 1024), big[2 * n * 1024:].view(n, 1024)], "one 12 GiB allocation (bench.py until now)")
