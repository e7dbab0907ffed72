#!/usr/bin/env python3
"""Box calibration: device copy / fill bandwidth next to the fused kernel, same process."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import __graft_entry__ as ge
pkg = ge.load_package(); dsp = pkg.SvtHipDsp(0)
dev = torch.device("cuda:0")
def timeit(fn, iters=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
n = 1 << 20
a = torch.empty(n * 1024, dtype=torch.int32, device=dev); b = torch.empty_like(a)
big = torch.empty(3 * n * 1024, dtype=torch.int32, device=dev)
src = torch.randint(0, 256, (n, 32, 32), dtype=torch.uint8, device=dev); pred = torch.randint(0, 256, (n, 32, 32), dtype=torch.uint8, device=dev)
qt = pkg.tables.quant_tables(8); qrow = {k: v[100].copy() for k, v in qt.items()}
_, isc = pkg.tables.scan_tables(3, 0); iscan = torch.from_numpy(isc).to(dev)
outs = (big[:n*1024].view(n, 1024), big[n*1024:2*n*1024].view(n, 1024), big[2*n*1024:].view(n, 1024), torch.zeros(n, dtype=torch.int16, device=dev), torch.zeros(n, dtype=torch.int32, device=dev))
for rnd in range(3):
    t_copy = timeit(lambda: b.copy_(a)); t_fill = timeit(lambda: big.fill_(7)); t_k = timeit(lambda: dsp.fwd_quant_sad(src, pred, 3, 0, qrow, iscan, outs=outs))
    print(json.dumps({"copy_GBps": 2 * a.numel() * 4 / t_copy / 1e6, "fill_GBps": big.numel() * 4 / t_fill / 1e6,
                      "kernel_ms": t_k, "kernel_GBps": 14342 * n / t_k / 1e6}), flush=True)
